"""ds_read_b128 bank-conflict degree of an LDS layout under the lane groups gfx950 actually uses
(MI355X_MICROARCH.md, LDS: four non-contiguous 16-lane groups; bank of byte address a = (a / 4) mod 64; identical addresses
broadcast).  Round 5 derived its layouts for CONTIGUOUS 16-lane groups; under the real grouping the 64-byte-row weight
tiles of ff_fused.hip were 2-way conflicted (the "unexplained 45 % of LDS-active cycles" of profiles/r05_ff_fused_pmc.txt).
  python tools/lds_conflicts.py        -> table of the layouts of csrc/*.hip"""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
        list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def degree_b128(addr):
    """addr(lane) -> byte address of the lane's 16 bytes; returns LDS cycles per wave instruction (4 = conflict free)."""
    cyc = 0
    for grp in G128:
        per_bank = {}
        for l in grp:
            a = addr(l)
            assert a % 16 == 0
            for w in range(4):
                per_bank.setdefault((a // 4 + w) % 64, set()).add(a)
        cyc += max(len(v) for v in per_bank.values())
    return cyc


def ff_old(lane):
    r16, g = lane & 15, lane >> 4
    return r16 * 64 + ((g ^ ((r16 >> 2) & 3)) << 4)


def ff_new(lane):
    r16, g = lane & 15, lane >> 4
    return r16 * 64 + ((g ^ (((r16 >> 3) & 1) * 3)) << 4)


def gemm(ks):
    def f(lane):
        r16, fk = lane & 15, lane >> 4
        return r16 * 128 + (((ks * 4 + fk) ^ (r16 & 7)) << 4)
    return f


def conv_halo(c, ks):
    """conv_gn.hip fragment read of tap offset c (halo pixel = tile row + c): 128-byte pixels, slot g ^ (pixel & 7)"""
    def f(lane):
        r16, g = lane & 15, lane >> 4
        hp = r16 + c
        return (hp << 7) + ((g ^ (hp & 7)) << 4) ^ (64 * ks)
    return f


if __name__ == "__main__":
    print("layout                                                       LDS cycles per ds_read_b128 (4 = conflict free)")
    print(f"ff_fused weight / activation tiles, round 5: slot ^ (row >> 2) & 3   {degree_b128(ff_old)}")
    print(f"ff_fused weight / activation tiles, round 6: slot ^ 3 (row >> 3)     {degree_b128(ff_new)}")
    for ks in (0, 1):
        print(f"gemm / tfront / xattn 128-byte rows, slot ^ (row & 7), ks = {ks}        {degree_b128(gemm(ks))}")
    for c in range(0, 9):
        print(f"conv_gn halo fragment, pixel offset {c}: ks 0 / 1                    {degree_b128(conv_halo(c, 0))} / {degree_b128(conv_halo(c, 1))}")
