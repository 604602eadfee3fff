// Coalesced global I/O for the "row-local" kernels (tfront.hip, xattn_fused.hip): a wave owns 16 rows x all columns and
// holds them in the MFMA accumulator layout -- lane (r16 = row, g) owns four consecutive columns 16 nb + 4 g .. of n-block nb,
// i.e. 8 bytes of ONE row.  Stored (or loaded) straight from that layout a wave instruction moves 64 pieces of 8 bytes
// scattered over 16 rows: measured on MI355X (round 6, lab PP_TF_DBG / PP_XA_DBG) the hs / q / k stores were 21 of
// pp_tfront's 49 us and the epilogue of pp_xattn_block 19 of its 56 us -- for 21 MB each.  Through a wave-PRIVATE LDS tile of
// 16 rows x 64 columns (144-byte rows) the same data crosses the memory pipeline as whole 128-byte lines, 16 bytes per
// lane.  Private tile + in-order LDS queue of a wave: no barrier and no s_waitcnt between the wave's writes and its reads;
// the empty asm statements only pin the compiler's order.  (A register-direct form -- v_permlane16_swap re-pairing, 64-byte
// pieces -- measured 43.5 against 41.6 us on pp_tfront: profiles/r06_rowtile_io.txt.)
#pragma once
#include "pp_common.h"

namespace {

constexpr int RT_LD = 144;                  // bytes per staged row: 128 + 16 (16-byte aligned pieces, banks spread)
constexpr int RT_TILE = 16 * RT_LD;         // bytes of one wave's tile

// lane's (o0, o1) = columns 16 j + 4 g .. + 3 of row r16 (n-block j of the current 64-column chunk) -> tile
PP_DEVINL void rt_put(char* stg, int r16, int g, int j, uint32_t o0, uint32_t o1) {
  *reinterpret_cast<u32x2_t*>(stg + r16 * RT_LD + j * 32 + g * 8) = u32x2_t{o0, o1};
}
// tile -> lane's (o0, o1) of n-block j
PP_DEVINL u32x2_t rt_get(const char* stg, int r16, int g, int j) {
  return *reinterpret_cast<const u32x2_t*>(stg + r16 * RT_LD + j * 32 + g * 8);
}
// the tile's 16 rows x 128 bytes as two 16-byte pieces per lane: piece q = lane + 64 jj -> row q >> 3, bytes 16 (q & 7) ..
PP_DEVINL void rt_flush(const char* stg, int lane, uint16_t* dst_row0, int ld, int col0) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int q = lane + 64 * jj, row = q >> 3, pc = q & 7;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stg + row * RT_LD + pc * 16);
    *reinterpret_cast<u32x4_t*>(dst_row0 + (size_t)row * ld + col0 + pc * 8) = v;
  }
  asm volatile("" ::: "memory");
}
// two row-major 16-byte pieces per lane (as loaded by rt_load_rows) -> tile
PP_DEVINL void rt_fill(char* stg, int lane, u32x4_t p0, u32x4_t p1) {
  asm volatile("" ::: "memory");
  *reinterpret_cast<u32x4_t*>(stg + (lane >> 3) * RT_LD + (lane & 7) * 16) = p0;
  *reinterpret_cast<u32x4_t*>(stg + ((lane + 64) >> 3) * RT_LD + (lane & 7) * 16) = p1;
  asm volatile("" ::: "memory");
}

}  // namespace
