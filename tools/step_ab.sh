#!/bin/bash
# Same-box interleaved A/B of the headline step (the boxes of the pool differ by up to 8 %: only interleaved runs on ONE box
# compare).  usage: [BENCH_ARGS="--config v2"] bash tools/step_ab.sh OUT ROUNDS "label|VAR=val VAR=val" "label2|..." ...
#   every label runs `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $BENCH_ARGS` ROUNDS times, round-robin;
#   an empty environment ("ship|") is the shipping configuration.  Lab switches need PP_LAB=1 (Python-side switches work with
#   the shipping library; kernel-side ones need PP_LIB=powerpaint_amd/libpp_hip_lab.so).  -> gpurun_out/OUT/ab.txt
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/$1; R=$2; shift 2
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-}"
for i in $(seq 1 $R); do
  for spec in "$@"; do
    label=${spec%%|*}; envs=${spec#*|}
    env $envs timeout 400 $B > $O/bench_${label}_$i.json 2>> $O/bench.err
  done
done
python - "$O" "$R" "$@" <<'PY' | tee $O/ab.txt
import json, sys
O, R, specs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
base = None
for spec in specs:
    label = spec.split("|")[0]
    ms, ln = [], None
    for i in range(1, R + 1):
        try:
            d = json.loads(open(f"{O}/bench_{label}_{i}.json").read().strip().splitlines()[-1])
            ms.append(d["ms_per_denoise_step"]); ln = d.get("launches_per_denoise_step")
        except Exception as e:
            ms.append(float("nan"))
    ok = [m for m in ms if m == m]
    mean = sum(ok) / len(ok) if ok else float("nan")
    if base is None:
        base = mean
    print(f"{label:16s} " + " ".join(f"{m:7.3f}" for m in ms) + f"  mean {mean:7.3f} ms/step  {4000.0 / (mean * 50):6.3f} img/s  "
          f"{(mean / base - 1) * 100:+5.2f} % vs {specs[0].split('|')[0]}  launches {ln}   [{spec.split('|', 1)[1]}]")
PY
tail -2 $O/bench.err
exit 0
