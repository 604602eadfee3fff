#!/usr/bin/env python
"""Per-launch HBM traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports both counters in KiB derived from the L2's
memory-side request counters (TCC_EA0_RDREQ x 64 B); on gfx950 wide coalesced reads (16 B / lane, global_load and
buffer_load ... lds alike -- every load in these kernels) are 128-byte requests tallied at 64 B, so FETCH_SIZE is
DOUBLED.  WRITE_SIZE is uncalibrated in the guide and taken as reported.  Infinity-Cache hits are counted, so this is
L2 <-> fabric traffic, an upper bound of DRAM traffic.
"""
import glob
import json
import sqlite3
import sys


def per_kernel(dbdir, counter):
    db = glob.glob(dbdir + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ix = {n: i for i, n in enumerate(cols)}
    agg = {}
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        kn = r[ix["kernel_name"]]
        a = agg.setdefault(kn, [0.0, 0])
        a[0] += r[ix["value"]]
        a[1] += 1
    return agg


def family(kn):
    if "pp_conv_gn_kernel" in kn:       # round 4: the halo-tile conv with GroupNorm + SiLU in its loader (csrc/conv_gn.hip)
        return "conv3x3 implicit GEMM"
    if "xattn_" in kn:
        return "fused cross-attention block"
    if "pp_gemm_kernel" in kn:
        args = kn.split("<")[1].split(">")[0].replace(" ", "").split(",")
        conv = args[4] == "1"
        return "conv3x3 implicit GEMM" if conv else "linear / 1x1 GEMM"
    for key, fam in (("splitk_reduce", "split-K combine"), ("attn_", "attention"), ("gn_stats", "groupnorm stats"),
                     ("gn_apply", "groupnorm apply")):
        if key in kn:
            return fam
    return None


def main(root, out):
    f = per_kernel(root + "/FETCH_SIZE", "FETCH_SIZE")
    w = per_kernel(root + "/WRITE_SIZE", "WRITE_SIZE")
    fams = {}
    for kn in set(f) | set(w):
        fam = family(kn)
        if fam is None:
            continue
        d = fams.setdefault(fam, dict(launches=0, fetch_kib=0.0, write_kib=0.0))
        d["launches"] += f.get(kn, [0, 0])[1]
        d["fetch_kib"] += f.get(kn, [0, 0])[0]
        d["write_kib"] += w.get(kn, [0, 0])[0]
    res = {"command": "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph (rocprofv3 --pmc crashes on hipGraph replays)",
           "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB; FETCH doubled per the gfx950 calibration in "
                   "MI355X_MICROARCH.md; counts L2<->fabric requests (Infinity-Cache hits included)",
           "families": {}}
    for fam, d in sorted(fams.items()):
        n = max(d["launches"], 1)
        res["families"][fam] = {"launches": d["launches"],
                                "read_bytes_per_launch": 2.0 * d["fetch_kib"] * 1024 / n,
                                "write_bytes_per_launch": d["write_kib"] * 1024 / n,
                                "bytes_per_launch": (2.0 * d["fetch_kib"] + d["write_kib"]) * 1024 / n}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
