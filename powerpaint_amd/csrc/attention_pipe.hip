// Software-pipelined flash attention for the hot self-attention shapes (head dim 40, N = 4096 / 1024 keys) on gfx950.
//
// Why a second kernel: in attn_fwd_kernel (attention.hip) every wave runs  QK^T MFMAs -> softmax VALU -> PV MFMAs  as
// three serial phases, and the waves that share a SIMD run the same phases in lock step.  PMC counters on MI355X
// (profiles/r01_attention_pmc.txt): VALU busy 52 % + MFMA busy 32 % + 16 % stalls = 100 %, i.e. the matrix pipe and the
// vector ALUs never overlap.  tools/micro/valu_rate.hip shows that ONE wave does overlap its own in-flight MFMA with
// independent VALU instructions that follow it in the instruction stream.  So the loop is restructured such that the
// MFMAs and the VALU work inside one iteration are independent:
//
//     stage t :   S(t+1) = K(t+1) Q^T          6 MFMA      \   interleaved in ONE basic block with
//                 O     += V(t-1)^T P(t-1)^T   8 MFMA      /   softmax(S(t)) -> P(t)      ~90 VALU (32 v_exp)
//
//   * S and P are double buffered in registers (the stage body is instantiated twice with the buffers swapped -> no
//     register copies); the O rescale of the (rare) running-max update is deferred to the top of the next stage, after
//     the PV MFMAs that still belong to the old max;
//   * K / V^T tiles arrive by LDS-DMA (buffer_load ... lds, no staging registers, no ds_write): 3-deep K ring, 5-deep
//     V^T ring, loads issued three tiles ahead with counted vmcnt + one s_barrier per stage;
//   * LDS rows keep an odd number of 16-byte slots (K: 7, V^T: 9) -> conflict-free fragment reads; the DMA is lane-linear
//     so the padding slot is just a lane whose source offset is out of range (the buffer descriptor returns 0);
//   * V^T is stored with natural key order, each PV A-fragment is two ds_read_b64 (keys 4h..4h+3 and 8+4h..8+4h+3 of the
//     16-key block = exactly the keys of the S^T accumulator registers this lane packed into its B-fragment);
//   * the softmax denominator comes out of the PV MFMAs: V^T row 63 is all ones.
//   * round 2: a wave owns 32 * QB queries.  QB = 2 on 32-key tiles does the same 14 MFMAs per stage with 7 fragment
//     reads instead of 14 (the LDS is 45 % index-active + 18 % conflict cycles at QB = 1) and half the tile DMAs per query.
// Restrictions (the host falls back to attn_fwd_kernel otherwise): d == 40, nk % 64 == 0.
#include <cstdlib>
#include <type_traits>

#include "pp_common.h"

namespace {

// queries per wave = 32 * QB (template parameter): QB = 2 lets every K / V^T fragment read from LDS feed two MFMAs

// SUBS (round 3): an LDS tile = SUBS stages of KB keys.  SUBS = 2 halves the barriers, counted waits, DMA issues and ring
// updates per key (a stage keeps the registers of KB keys: S / P double buffers for 64 keys x 64 queries do not fit).
template <int D, int KB, int NW, int SUBS = 1>     // NW = waves per block (4: two blocks per CU; 8: one, half the tile DMAs per query)
struct PCfg {
  static constexpr int KT = KB * SUBS;                    // keys per LDS tile
  static constexpr int JB = KB / 32;                      // 32-key S^T blocks per stage
  static constexpr int DP = (D + 15) / 16 * 16;
  static constexpr int DS = DP / 16;
  static constexpr int VROWS = (D + 1 + 31) / 32 * 32;   // head-dim rows + the all-ones row, whole 32-row MFMA tiles
  static constexpr int DT = VROWS / 32;
  static constexpr int KSL = DP / 8 + 1;                 // 16-B slots per K row (odd)
  static constexpr int KS = KSL * 16;
  static constexpr int KTILE = (KT * KS + 1023) / 1024 * 1024;   // whole DMA instructions (tail slots unused)
  static constexpr int KI = KTILE / 1024;                // wave-wide DMA instructions per K tile
  static constexpr int VSL = KT / 8 + 1;                 // 9 slots per V^T row (64 keys)
  static constexpr int VS = VSL * 16;
  static constexpr int VTILE = VROWS * VS;
  static constexpr int VI = (D * VS + 1023) / 1024;      // DMA instructions for the streamed rows [0, D)
  static constexpr int LPW = (KI + NW - 1) / NW + (VI + NW - 1) / NW;   // DMA instructions per wave per tile
  static constexpr int NKB = SUBS == 1 ? 3 : 4, NVB = 5; // ring depths (SUBS = 2: tiles are issued three PAIRS of stages ahead)
  static constexpr int VBASE = NKB * KTILE;
  static constexpr int DUMMY = VBASE + NVB * VTILE;
  static constexpr int LDS = DUMMY + 1024;
  static_assert(VI * 1024 <= (VROWS - 1) * VS, "DMA spill of the V^T tile must stay below the ones row");
  static_assert(KSL % 2 == 1 && VSL % 2 == 1, "odd slot counts");
};

constexpr float RESCALE_THR = 8.0f;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// OPT (round 3; bit mask, PP_ATTN_OPT_DEFAULT ships):
//   1  the cross-half row maximum by v_permlane32_swap (VALU) instead of __shfl_xor(.., 32) = ds_bpermute_b32, whose
//      `s_waitcnt lgkmcnt(0)` also drained the prefetched K / V^T fragment reads at the head of every stage;
//   2  the exp argument as two scalar v_fma_f32 instead of one v_pk_fma_f32 (packed fp32 VALU does not overlap an
//      in-flight MFMA on gfx950: MI355X_MICROARCH.md, "price of one filler beside MFMAs");
//   4  fragment reads issued at the HEAD of a slot, two fragments ahead, fenced from the slot's VALU block (the
//      compiler otherwise sinks the ds_read behind the slot's exps and the next MFMA waits the whole LDS latency).
//  16  64-key LDS tiles consumed as two 32-key stages (PCfg SUBS = 2): half the barriers / counted waits / DMA issues /
//      ring updates per key.
//   8  fewer issue slots per stage (the wave is instruction-issue bound: ~236 instructions per 32-key stage at ~5 cycles):
//      the v_cvt_pk of an exp step is issued one step LATER (behind the next step's v_exp pair: no `s_nop` for the
//      transcendental-result hazard, 13 per stage before), and the deferred-rescale factor exp2(m_old - m_new) is
//      evaluated inside the (rare) rescale branch instead of on every stage.
//  64  (round 5) LOG2 form: q holds Q * scale * log2(e) (the producer's epilogue pre-multiplies: pp_tfront q_scale), and the
//      running reference -m is the INITIAL ACCUMULATOR of the first QK^T MFMA of a block instead of zero, so a score leaves
//      the matrix pipe as the exp2 argument itself: no v_fma per score (32 of ~120 VALU instructions per 32-key stage; the
//      kernel is VALU-issue bound, profiles/r03_attention_pmc.txt).  For that the stage runs its PV MFMAs FIRST and the
//      QK^T ones behind the running-max phase (the reference S(t+1) is baked with is the one P(t) uses), and the rare
//      reference update (a lane's tile maximum more than RESCALE_THR above its reference; always at the first tile) fixes
//      up the 16 scores per lane and block it arrived too late for, in a wave-uniform branch.  Needs bit 8.
template <int D, int KB, int dbg, int EDT, int NW, int QB, int OPT>
__global__ void __launch_bounds__(64 * NW, (KB == 32 && QB == 1 ? 4 : 2))
attn_pipe_kernel(const uint16_t* __restrict__ q, int ldq, const uint16_t* __restrict__ k, int ldk,
                 const uint16_t* __restrict__ vt, int ldvt, uint16_t* __restrict__ o, int ldo, int heads, int nq, int nk,
                 float scale_log2e) {
  // dbg (PP_ATTN_DBG, timing experiments only; results are garbage): 1 no MFMA, 2 no exp, 4 no tile DMA / barrier,
  // 8 no LDS fragment reads, 16 no output store
  constexpr int SUBS = (OPT & 16) ? 2 : 1;
  constexpr int KT = KB * SUBS;
  constexpr bool BAKE = (OPT & 64) != 0;
  static_assert(!BAKE || (OPT & 8), "the LOG2 form keeps the rescale exponent in alpha[] (OPT bit 8)");
  using C = PCfg<D, KB, NW, SUBS>;
  using E = E16<EDT>;
  typedef typename E::v8 v8_t;
  constexpr int JB = C::JB;
  constexpr int T = 64 * NW;
  constexpr int QW = 32 * QB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (QW * NW) + wave * QW;
  const int qi = lane & 31, half = lane >> 5;
  const int ntiles = nk / KB;

  // ---- LDS init: V^T ring all zero (ring slot NVB-1 is read by stage 0 with P = 0; 0 x garbage could be NaN), ones row
  for (int i = tid; i < C::NVB * C::VTILE / 16; i += T)
    *reinterpret_cast<u32x4_t*>(smem + C::VBASE + i * 16) = u32x4_t{0u, 0u, 0u, 0u};
  __syncthreads();
  for (int i = tid; i < C::NVB * (KT / 2); i += T) {
    const int bufi = i / (KT / 2), w = i - bufi * (KT / 2);
    *reinterpret_cast<uint32_t*>(smem + C::VBASE + bufi * C::VTILE + (C::VROWS - 1) * C::VS + w * 4) = E::pack2(1.0f, 1.0f);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- Q fragments (B operand): lane = query qi, k-slot = 8*half + jj  ->  Q[q0+qi][16 s + 8 half + jj]
  u32x4_t qraw[QB][C::DS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = q0 + 32 * qb + qi;
    const uint16_t* qp = q + ((size_t)b * nq + (qrow < nq ? qrow : 0)) * ldq + h * D;
#pragma unroll
    for (int s = 0; s < C::DS; ++s) {
      const int dc = 16 * s + 8 * half;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (qrow < nq && dc < D) v = *reinterpret_cast<const u32x4_t*>(qp + dc);
      qraw[qb][s] = v;
    }
  }

  // ---- DMA plan.  Per tile every wave issues LPW wave-wide 16-byte DMA instructions: i < LPK move K slots, the rest V^T
  // slots; which slots = 64 * (i' * NW + wave).  Instructions past the end of a tile (and the padding slots inside it)
  // carry an out-of-range source offset -> they write zeros, into a dummy KB of LDS when the whole instruction is dead.
  // Everything that changes from tile to tile is SCALAR state advanced with a handful of SALU ops per stage (descriptor
  // base / size, ring offsets) -- the per-stage bookkeeping must not eat the issue slots of the 2 waves per SIMD.
  const uint16_t* k_bh = k + (size_t)b * nk * ldk + h * D;
  const uint16_t* vt_bh = vt + ((size_t)b * heads + h) * D * (size_t)ldvt;
  constexpr int LPK = (C::KI + NW - 1) / NW, LPV = (C::VI + NW - 1) / NW;
  static_assert(LPK + LPV == C::LPW || LPK + LPV <= C::LPW, "DMA split");
  int vo[LPK + LPV];
  int rel[LPK + LPV];       // LDS offset inside the ring slot, or -1: dead instruction -> dummy area   (wave-uniform)
#pragma unroll
  for (int i = 0; i < LPK + LPV; ++i) {
    int off = (int)PP_OOB;
    if (i < LPK) {
      const int g = i * NW + wave;
      const int L = 64 * g + lane, r = L / C::KSL, p = L - r * C::KSL;
      if (g < C::KI && r < KT && p < D / 8) off = (r * ldk + p * 8) * 2;
      rel[i] = g < C::KI ? g * 1024 : -1;
    } else {
      const int g = (i - LPK) * NW + wave;
      const int L = 64 * g + lane, r = L / C::VSL, p = L - r * C::VSL;
      if (g < C::VI && r < D && p < KT / 8) off = (r * ldvt + p * 8) * 2;
      rel[i] = g < C::VI ? g * 1024 : -1;
    }
    vo[i] = off;
  }
  int kread = SUBS == 1 ? 1 : 0, vread = C::NVB - 1;   // ring slots stage 0 reads: K(1), V(-1)  (SUBS = 2: K(1) = 2nd half of tile 0)
  int kring = 0, vring = 0;                 // ring slots of the NEXT tile to issue
  int t_issue = 0;                          // its index
  // The descriptors never change: the tile advance is the scalar offset operand of the DMA (soffset is not part of the
  // range check, so dead lanes -- voffset out of range -- still read zeros), and the two look-ahead issues past the last
  // tile get a zero-sized descriptor.  ~20 SALU instructions per stage instead of ~45: the SIMD is issue-bound.
  const uint32_t kbytes = (uint32_t)((nk - 1) * ldk + D) * 2u, vbytes = (uint32_t)D * (uint32_t)ldvt * 2u;
  const int kstep = KT * ldk * 2, vstep = KT * 2;
  int ksoff = 0, vsoff = 0;
  auto issue = [&]() {
    const bool live = t_issue * SUBS < ntiles;      // (t_issue counts LDS tiles)
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(k_bh, live ? kbytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vt_bh, live ? vbytes : 0u);
    char* kdst = smem + kring * C::KTILE;
    char* vdst = smem + C::VBASE + vring * C::VTILE;
#pragma unroll
    for (int i = 0; i < LPK + LPV; ++i) {
      const int voff = vo[i];
      char* dst = rel[i] < 0 ? smem + C::DUMMY : (i < LPK ? kdst : vdst) + rel[i];
      if (i < LPK) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_ptr_t)dst, 16, voff, ksoff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)dst, 16, voff, vsoff, 0, 0);
    }
    ++t_issue;
    ksoff += kstep;
    vsoff += vstep;
    kring = kring + 1 == C::NKB ? 0 : kring + 1;
    vring = vring + 1 == C::NVB ? 0 : vring + 1;
  };

  f32x16_t oacc[QB][C::DT];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][t][r] = 0.f;
  float m_run[QB], alpha[QB];   // running max (log2 domain) / deferred O rescale factor of the previous stage, per q-block
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -1.0e30f; alpha[qb] = (OPT & 8) ? 0.0f : 1.0f; }
  bool pend = false;            // wave-uniform: some alpha != 1 somewhere
  // (LOG2 form) -reference of every query of the lane, as the accumulator the first QK^T MFMA of a block starts from;
  // the reference starts at 0 and the first tile always replaces it by that tile's row maximum
  f32x16_t cinit[BAKE ? QB : 1];
#pragma unroll
  for (int qb = 0; qb < (BAKE ? QB : 1); ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[qb][r] = 0.f;
  bool first = true;            // wave-uniform

  // plain (non-interleaved) pieces: prologue S(0) and the drain PV
  auto qk = [&](const char* ks, f32x16_t (&sn)[QB][JB]) {
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::DS; ++s) {
        const v8_t kf = *reinterpret_cast<const v8_t*>(ks + (32 * j + qi) * C::KS + (2 * s + half) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
          sn[qb][j] = E::mfma32(kf, __builtin_bit_cast(v8_t, qraw[qb][s]), s == 0 ? zero : sn[qb][j], 0, 0, 0);
      }
    }
  };
  auto vfrag = [&](const char* vs, int j, int u, int dt) {
    const char* src = vs + (dt * 32 + qi) * C::VS + (32 * j + 16 * u + 4 * half) * 2;
    const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(src);
    const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(src + 16);
    const u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(v8_t, w);
  };
  auto pv = [&](const char* vs, const v8_t (&pp)[QB][JB][2]) {
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          const v8_t vf = vfrag(vs, j, u, dt);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][dt] = E::mfma32(vf, pp[qb][j][u], oacc[qb][dt], 0, 0, 0);
        }
  };

  // stage t: consumes S(t) (sc) and P(t-1) (pp); produces S(t+1) (sn) and P(t) (pc).
  // The body is choreographed by hand: NM = QB * JB * (DS + 2 DT) MFMAs, one per slot; every slot also carries its share of
  // the softmax VALU work, and every QB-th slot the LDS read of the fragment FDF fragments ahead (a fragment feeds the QB
  // consecutive MFMAs of the wave's q-blocks).  sched_barrier(0) between slots keeps the compiler from regrouping (left
  // alone it emits all MFMAs back to back, then the VALU block: zero overlap).
  auto stage = [&](auto sub_tag, f32x16_t (&sc)[QB][JB], f32x16_t (&sn)[QB][JB], v8_t (&pc)[QB][JB][2],
                   const v8_t (&pp)[QB][JB][2]) {
    constexpr int sub = decltype(sub_tag)::value;       // t & 1
    // SUBS = 2: `sub` = t & 1 is a compile-time constant of the two stage instantiations (the loop is unrolled by two).
    // Stage t reads K(t+1) and V(t-1): with 64-key tiles the even stage t = 2p reads the SECOND halves of K tile p and of
    // V^T tile p-1, the odd stage the FIRST halves of K tile p+1 and V^T tile p.  Barrier, counted wait and the issue of
    // tile p+3 happen once per pair, at the even stage.
    if (!(dbg & 4) && (SUBS == 1 || sub == 0)) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::LPW) : "memory");   // tile t+1 (issued two stages ago) has landed
    asm volatile("s_barrier" ::: "memory");                         // ... for every wave; stage t-1 reads are done
    issue();
    }
    if (pend) {                                                     // deferred rescale: after PV(t-2), before PV(t-1)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float al = (OPT & 8) ? __builtin_amdgcn_exp2f(alpha[qb]) : alpha[qb];   // (OPT 8: alpha[] holds m_old - m_new)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qb][dt][r] *= al;
      }
    }
    const char* ks = smem + kread * C::KTILE;            // K(t+1)
    const char* vs = smem + C::VBASE + vread * C::VTILE; // V(t-1)
    if constexpr (SUBS == 1) {
      kread = kread + 1 == C::NKB ? 0 : kread + 1;
      vread = vread + 1 == C::NVB ? 0 : vread + 1;
    } else {
      if (sub == 0) {        // even stage: second halves; both rings move on to the tiles the odd stage reads
        ks += KB * C::KS;
        vs += KB * 2;
        kread = kread + 1 == C::NKB ? 0 : kread + 1;
        vread = vread + 1 == C::NVB ? 0 : vread + 1;
      }
    }
    constexpr int NKF = JB * C::DS, NF = NKF + JB * 2 * C::DT;   // K fragments / all fragments of a stage
    constexpr int NQK = NKF * QB, NM = NF * QB;                  // QK^T slots / all slots
    constexpr int FDF = (dbg >> 4) ? (dbg >> 4) : ((OPT & 4) ? 2 : (QB == 1 ? 2 : 1));   // fragment prefetch distance, in fragments
    //                                       (= two slots either way; QB = 2 with FDF 2 measured the same)
    constexpr int NVF = NF - NKF;
    // slot order: QK^T fragments first, then PV; LOG2 form: PV first (S(t+1) must see the reference this stage settles)
    auto fk = [](int kk) { return BAKE ? (kk < NVF ? NKF + kk : kk - NVF) : kk; };
    constexpr int SB = QB * JB;                  // 32 x 32 score blocks per stage
    constexpr int MAXSLOTS = 2 * SB;             // slots carrying the running-max phase (8 scores each)
    constexpr int NES = 8 * SB;                  // exp steps (2 scores each)
    constexpr int ESLOTS = NM - MAXSLOTS;        // slots carrying them
    v8_t frag[NF];
    auto fetch = [&](int k) {
      if (dbg & 8) { frag[k] = __builtin_bit_cast(v8_t, qraw[0][0]); return; }
      if (k < NKF) {
        const int j = k / C::DS, sidx = k % C::DS;
        frag[k] = *reinterpret_cast<const v8_t*>(ks + (32 * j + qi) * C::KS + (2 * sidx + half) * 16);
      } else {
        const int g = k - NKF, ju = g / C::DT, dt = g % C::DT;
        frag[k] = vfrag(vs, ju >> 1, ju & 1, dt);
      }
    };
#pragma unroll
    for (int k = 0; k < FDF; ++k) fetch(fk(k));
    float tmax[QB];
    f32x2_t c2 = {scale_log2e, scale_log2e}, nm2[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { tmax[qb] = -1.0e30f; nm2[qb] = f32x2_t{0.f, 0.f}; }
    u32x4_t w[QB][JB][2];
    float pend0 = 0.f, pend1 = 0.f;              // (OPT 8) the exps whose pack is still to come
    bool any = false;
    int es = 0;                                  // exp steps done (compile-time after unrolling)
#pragma unroll
    for (int f = 0; f < NM; ++f) {
      const int kk = f / QB, qb = f % QB;        // fragment of this slot (in slot order), q-block it multiplies
      const int k = fk(kk);
      if (qb == 0 && kk + FDF < NF) {
        fetch(fk(kk + FDF));
        if (OPT & 4) __builtin_amdgcn_sched_barrier(0);   // the read leaves at the head of the slot
      }
      if (dbg & 1) {
        asm volatile("" ::"v"(frag[k]));
      } else if (k < NKF) {
        const int j = k / C::DS, sidx = k % C::DS;
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sn[qb][j] = E::mfma32(frag[k], __builtin_bit_cast(v8_t, qraw[qb][sidx]),
                              sidx == 0 ? (BAKE ? cinit[BAKE ? qb : 0] : zero) : sn[qb][j], 0, 0, 0);
      } else {
        const int g = k - NKF, ju = g / C::DT, dt = g % C::DT;
        oacc[qb][dt] = E::mfma32(frag[k], pp[qb][ju >> 1][ju & 1], oacc[qb][dt], 0, 0, 0);
      }
      if (f < MAXSLOTS) {                        // running max over a lane's 16 scores of one block, 8 per slot
        const int sb = f >> 1, mq = sb / JB, mj = sb % JB, r0 = (f & 1) * 8;
#pragma unroll
        for (int r = r0; r < r0 + 8; r += 2)
          tmax[mq] = __builtin_fmaxf(__builtin_fmaxf(tmax[mq], sc[mq][mj][r]), sc[mq][mj][r + 1]);
        if (mj == JB - 1 && (f & 1)) {           // last block of q-block mq: its row maximum is complete
          float tm;
          if (OPT & 1) {      // lanes i and i + 32 exchange through the VALU swap: no LDS round trip, no lgkmcnt drain.
            // swap(x, x): the first result holds the low half's value in both halves, the second the high half's
            // Inline asm on purpose: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) returned the FIRST result
            // in both elements here (a v_mov over the second register right after the swap: the low half's maximum only,
            // visible as 1-ulp output differences because softmax is invariant to the reference value).  s_nop 1 = the two
            // wait states between a VALU write of an operand and the swap (cdna_hip_programming.md, T21).
            float r0 = tmax[mq], r1 = tmax[mq];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(r0), "+v"(r1));
            tm = fmaxf(r0, r1);
          } else {
            tm = fmaxf(tmax[mq], __shfl_xor(tmax[mq], 32, 64));
          }
          if constexpr (BAKE) {
            // tm = this tile's row maximum MINUS the lane's reference (the scores carry it already)
            const bool need = first || !__all(tm <= RESCALE_THR);
            alpha[mq] = 0.f;
            if (need) {
              const float dlt = first ? tm : fmaxf(tm, 0.f);      // the reference moves up by dlt (first tile: to the maximum)
#pragma unroll
              for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[mq][jj][r] -= dlt;
#pragma unroll
              for (int r = 0; r < 16; ++r) cinit[mq][r] -= dlt;
              alpha[mq] = first ? 0.f : -dlt;                     // (O is still zero at the first tile)
              any = true;
            }
          } else {
          const float ts = tm * scale_log2e;
          const bool need = !__all(ts - m_run[mq] <= RESCALE_THR);
          const float m_new = need ? fmaxf(m_run[mq], ts) : m_run[mq];
          alpha[mq] = (OPT & 8) ? (m_run[mq] - m_new) : __builtin_amdgcn_exp2f(m_run[mq] - m_new);
          any = any || need;
          m_run[mq] = m_new;
          nm2[mq] = f32x2_t{-m_new, -m_new};
          }
        }
        if (f == MAXSLOTS - 1) { pend = any; first = false; }
      } else {                                   // exp steps: 2 scores each (pk_fma, 2 x exp2, cvt_pk)
        const int kk = f - MAXSLOTS;
        const int upto = (NES * (kk + 1) + ESLOTS - 1) / ESLOTS;
#pragma unroll
        for (; es < upto; ++es) {
          const int sb = es >> 3, eq = sb / JB, ej = sb % JB, u = (es >> 2) & 1, e = es & 3;
          const f32x2_t s2 = {sc[eq][ej][8 * u + 2 * e], sc[eq][ej][8 * u + 2 * e + 1]};
          f32x2_t e2;
          if (BAKE || (OPT & 32)) {                         // LOG2 form: the score IS the exponent  (32: lab timing probe of
            e2 = s2;                                        //  the same without the reference, results garbage)
          } else if (OPT & 2) {
            float e0 = __builtin_fmaf(s2[0], scale_log2e, nm2[eq][0]), e1 = __builtin_fmaf(s2[1], scale_log2e, nm2[eq][1]);
            asm volatile("" : "+v"(e0), "+v"(e1));        // (keeps LLVM's SLP vectoriser from re-packing the pair)
            e2 = f32x2_t{e0, e1};
          } else {
            e2 = __builtin_elementwise_fma(s2, c2, nm2[eq]);
          }
          if (OPT & 8) {
            // this step's exps now, the PREVIOUS step's pack behind them (its exps are a step old: no hazard wait)
            float x0 = __builtin_amdgcn_exp2f(e2[0]), x1 = __builtin_amdgcn_exp2f(e2[1]);
            asm volatile("" : "+v"(x0), "+v"(x1));   // pinned here: P(t) is only consumed next stage
            if (es > 0) {
              const int ps = es - 1, psb = ps >> 3, pq = psb / JB, pj = psb % JB, pu = (ps >> 2) & 1, pe = ps & 3;
              w[pq][pj][pu][pe] = E::pack2(pend0, pend1);
              asm volatile("" ::"v"(w[pq][pj][pu][pe]));
            }
            pend0 = x0;
            pend1 = x1;
          } else {
            w[eq][ej][u][e] = (dbg & 2) ? E::pack2(e2[0], e2[1])
                                        : E::pack2(__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1]));
            asm volatile("" ::"v"(w[eq][ej][u][e]));   // P(t) is only consumed next stage: keep LLVM from sinking the exps there
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (OPT & 8) {                               // the last step's pack
      constexpr int ps = NES - 1, psb = ps >> 3, pq = psb / JB, pj = psb % JB, pu = (ps >> 2) & 1, pe = ps & 3;
      w[pq][pj][pu][pe] = E::pack2(pend0, pend1);
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
#pragma unroll
        for (int u = 0; u < 2; ++u) pc[qb][jj][u] = __builtin_bit_cast(v8_t, w[qb][jj][u]);
  };

  f32x16_t sA[QB][JB], sB[QB][JB];
  v8_t pA[QB][JB][2], pB[QB][JB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int jj = 0; jj < JB; ++jj)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pA[qb][jj][u] = __builtin_bit_cast(v8_t, u32x4_t{0u, 0u, 0u, 0u});
        pB[qb][jj][u] = pA[qb][jj][u];
      }

  issue();
  issue();
  issue();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::LPW) : "memory");
  asm volatile("s_barrier" ::: "memory");
  qk(smem, sA);                                                    // S(0)

  for (int t = 0; t < ntiles; t += 2) {
    stage(std::integral_constant<int, 0>{}, sA, sB, pA, pB);
    if (t + 1 < ntiles) stage(std::integral_constant<int, 1>{}, sB, sA, pB, pA);
  }
  // ---- drain: PV of the last tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (pend) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float al = (OPT & 8) ? __builtin_amdgcn_exp2f(alpha[qb]) : alpha[qb];
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][dt][r] *= al;
    }
  }
  {
    const char* vs = smem + C::VBASE + (((ntiles - 1) / SUBS) % C::NVB) * C::VTILE + (SUBS == 2 ? KB * 2 : 0);   // V(ntiles-1)
    if (ntiles & 1) pv(vs, pA);
    else pv(vs, pB);
  }

  // ---- epilogue: denominator = O^T row VROWS-1 (tile DT-1, r = 15, lane half 1)
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = __shfl(oacc[qb][C::DT - 1][15], qi + 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + 32 * qb + qi;
    if (qrow < nq) {
      uint16_t* op = o + ((size_t)b * nq + qrow) * ldo + h * D;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dc = dt * 32 + 8 * g + 4 * half;
          if (dc < D) {
            u32x2_t w;
            w[0] = E::pack2(oacc[qb][dt][4 * g + 0] * inv, oacc[qb][dt][4 * g + 1] * inv);
            w[1] = E::pack2(oacc[qb][dt][4 * g + 2] * inv, oacc[qb][dt][4 * g + 3] * inv);
            if constexpr (!(dbg & 16)) *reinterpret_cast<u32x2_t*>(op + dc) = w;
            else if (w[0] == 0x12345u) op[0] = 1;
          }
        }
    }
  }
}

}  // namespace

// Returns PP_ERR_UNSUPPORTED for shapes this kernel does not cover (the caller then uses attn_fwd_kernel).
// Default: 64-key tiles, two workgroups (8 waves) per CU; carries the PP_ATTN_DBG ablation variants.  PP_ATTN_KB=32:
// 32-key tiles, <= 128 VGPRs, four workgroups per CU -- measured identical (347 vs 346 us at N = 4096): doubling the
// occupancy hides nothing, the SIMD is issue-bound (~230 instructions per wave-tile at ~4 cycles + 14 MFMA at 32).
// shipping: 29 = VALU-swap maximum + early fragment reads + late v_cvt / lazy rescale factor + 64-key LDS tiles
// (profiles/r03_attention_opt_ab.txt: 287 -> 254 us at N = 4096 / batch 8 on one box, 1981 -> 1869 us at N = 16384 / batch 4,
// bit-identical output).  The 64-key tiles (bit 16) belong to the 32-key-stage kernels (QB = 2); the 64-key-stage kernel
// (QB = 1) would need 128-key tiles and 143 KB of LDS per workgroup, so it keeps single-stage tiles.
#ifndef PP_ATTN_OPT_DEFAULT
#define PP_ATTN_OPT_DEFAULT 29
#endif
template <int KB, int DBG, int EDT, int NW = 4, int QB = 1, int OPT = (KB == 32 ? PP_ATTN_OPT_DEFAULT : (PP_ATTN_OPT_DEFAULT & ~16)), int D = 40>
static int launch_pipe(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                       int batch, int heads, int nq, int nk, float sl2, hipStream_t st) {
  using C = PCfg<D, KB, NW, (OPT & 16) ? 2 : 1>;
  if (pp_func_lds(reinterpret_cast<const void*>(attn_pipe_kernel<D, KB, DBG, EDT, NW, QB, OPT>), C::LDS,
                  "hipFuncSetAttribute(attention pipe)") != PP_OK)
    return PP_ERR_LAUNCH;
  const dim3 grid((nq + 32 * QB * NW - 1) / (32 * QB * NW), heads, batch), block(64 * NW);
  hipLaunchKernelGGL((attn_pipe_kernel<D, KB, DBG, EDT, NW, QB, OPT>), grid, block, C::LDS, st, (const uint16_t*)q, ldq,
                     (const uint16_t*)k, ldk, (const uint16_t*)vt, ldvt, (uint16_t*)o, ldo, heads, nq, nk, sl2);
  PP_CHECK_LAUNCH("attn_pipe_kernel");
  return PP_OK;
}

// variant (PP_ATTN_* of pp_hip.h): AUTO picks by shape; PIPE_Q32 / PIPE_Q64 force the 32- / 64-queries-per-wave kernel
// (PP_ERR_UNSUPPORTED when the shape is outside it) -- the parity tests address each shipping kernel by name.
// key counts the software-pipelined kernels take (64-key LDS tiles, ring depth 4): the ONE predicate behind the launcher
// and pp_attention_log2_ok()
bool pp_attention_pipe_keys_ok(int nk) { return nk % 64 == 0 && nk >= 4 * 64; }

int pp_attention_pipe_launch(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                             int batch, int heads, int nq, int nk, int d, float sl2, int dtype, int variant,
                             hipStream_t st) {
#define PP_ARGS q, ldq, k, ldk, vt, ldvt, o, ldo, batch, heads, nq, nk, sl2, st
  // both tile sizes need whole tiles and at least four of them (three look-ahead issues + the prologue)
  if (!pp_attention_pipe_keys_ok(nk)) return PP_ERR_UNSUPPORTED;
#ifdef PP_LAB
  // d = 80 (the 32x32 level) on the pipelined loop, 32 queries per wave: measured and NOT shipped -- 38.6 against 36.5 us
  // per launch for the three-phase kernel inside the step (9.515 against 9.481 ms per step; the 96-row V^T tiles leave room
  // for one 4-wave workgroup per CU only).  PP_ATTN_PIPE80=1 in a lab build runs it.
  if (d == 80 && pp_lab_env("PP_ATTN_PIPE80", 0)) {
    constexpr int OPT80 = PP_ATTN_OPT_DEFAULT & ~16;
    if (dtype == PP_DT_F16) return launch_pipe<64, 0, PP_DT_F16, 4, 1, OPT80, 80>(PP_ARGS);
    return launch_pipe<64, 0, PP_DT_BF16, 4, 1, OPT80, 80>(PP_ARGS);
  }
#endif
  if (d != 40) return PP_ERR_UNSUPPORTED;
#ifdef PP_LAB
  // timing experiments (tools/attn_ablate.py, tools/attn_pmc.sh): PP_ATTN_KB=32 32-key tiles at 4 workgroups per CU,
  // PP_ATTN_NW=8 one 8-wave workgroup per CU, PP_ATTN_DBG ablation masks (results are garbage), PP_ATTN_QB=1|2
  const int kb = pp_lab_env("PP_ATTN_KB", 64), dbg = pp_lab_env("PP_ATTN_DBG", 0), nw = pp_lab_env("PP_ATTN_NW", 4);
  const int qbk = pp_lab_env("PP_ATTN_QB", 0);
  if (variant == PP_ATTN_AUTO && qbk) variant = qbk == 2 ? PP_ATTN_PIPE_Q64 : PP_ATTN_PIPE_Q32;
  if (dtype == PP_DT_BF16 && (kb != 64 || dbg != 0 || nw != 4)) {
    if (nw == 8 && kb == 64 && dbg == 0) return launch_pipe<64, 0, PP_DT_BF16, 8>(PP_ARGS);
    if (kb == 32) return launch_pipe<32, 0, PP_DT_BF16>(PP_ARGS);
    if (kb != 64) return PP_ERR_BAD_ARG;
    switch (dbg) {
      case 1: return launch_pipe<64, 1, PP_DT_BF16>(PP_ARGS);
      case 2: return launch_pipe<64, 2, PP_DT_BF16>(PP_ARGS);
      case 4: return launch_pipe<64, 4, PP_DT_BF16>(PP_ARGS);
      case 8: return launch_pipe<64, 8, PP_DT_BF16>(PP_ARGS);
      case 12: return launch_pipe<64, 12, PP_DT_BF16>(PP_ARGS);
      case 13: return launch_pipe<64, 13, PP_DT_BF16>(PP_ARGS);
      case 15: return launch_pipe<64, 15, PP_DT_BF16>(PP_ARGS);
      default: return PP_ERR_BAD_ARG;
    }
  }
#endif
  // 64 queries per wave on 32-key tiles (QB = 2: every K / V^T fragment read from LDS feeds two MFMAs, half the tile DMAs
  // per query): 319 us against 338 us at N = 4096 hot, -0.5 % on the UNet step.  Needs two 256-query workgroups per CU
  // to be worth it.
  const long long wg2 = (long long)batch * heads * ((nq + 255) / 256);
  const bool log2q = variant == PP_ATTN_PIPE_LOG2;      // q = Q * scale * log2(e): same kernel choice as AUTO, OPT bit 64
  const bool q64 = variant == PP_ATTN_PIPE_Q64 || ((variant == PP_ATTN_AUTO || log2q) && wg2 >= 512);
  if (log2q) {
    constexpr int O64 = PP_ATTN_OPT_DEFAULT | 64, O32 = (PP_ATTN_OPT_DEFAULT & ~16) | 64;
#ifdef PP_LAB
    if (q64 && dtype == PP_DT_BF16 && pp_lab_env("PP_ATTN_NOSTORE", 0)) return launch_pipe<32, 16, PP_DT_BF16, 4, 2, O64>(PP_ARGS);
#endif
    if (q64) {
      if (dtype == PP_DT_F16) return launch_pipe<32, 0, PP_DT_F16, 4, 2, O64>(PP_ARGS);
      return launch_pipe<32, 0, PP_DT_BF16, 4, 2, O64>(PP_ARGS);
    }
    if (dtype == PP_DT_F16) return launch_pipe<64, 0, PP_DT_F16, 4, 1, O32>(PP_ARGS);
    return launch_pipe<64, 0, PP_DT_BF16, 4, 1, O32>(PP_ARGS);
  }
#ifdef PP_LAB
  if (q64 && dtype == PP_DT_BF16 && pp_lab_env("PP_ATTN_OPT", -1) >= 0) {      // A/B of the round-3 loop changes
    switch (pp_lab_env("PP_ATTN_OPT", 0)) {
      case 0: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 0>(PP_ARGS);
      case 1: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 1>(PP_ARGS);
      case 2: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 2>(PP_ARGS);
      case 3: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 3>(PP_ARGS);
      case 4: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 4>(PP_ARGS);
      case 5: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 5>(PP_ARGS);
      case 6: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 6>(PP_ARGS);
      case 7: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 7>(PP_ARGS);
      case 13: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 13>(PP_ARGS);
      case 21: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 21>(PP_ARGS);
      case 29: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 29>(PP_ARGS);
      case 61: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 61>(PP_ARGS);
      case 15: return launch_pipe<32, 0, PP_DT_BF16, 4, 2, 15>(PP_ARGS);
      default: return PP_ERR_BAD_ARG;
    }
  }
#endif
  if (q64) {
    if (dtype == PP_DT_F16) return launch_pipe<32, 0, PP_DT_F16, 4, 2>(PP_ARGS);
    return launch_pipe<32, 0, PP_DT_BF16, 4, 2>(PP_ARGS);
  }
  if (dtype == PP_DT_F16) return launch_pipe<64, 0, PP_DT_F16>(PP_ARGS);
  return launch_pipe<64, 0, PP_DT_BF16>(PP_ARGS);
#undef PP_ARGS
}
