#!/bin/bash
# Round-4 full GPU suite + smoke (what the driver runs at round end).  -> gpurun_out/r04full/
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04full
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f gpurun_out/parity_r04.txt gpurun_out/parity_achieved.txt
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in ${SECONDS}s"; tail -4 $O/pytest_gpu.log | cut -c1-250; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
cp gpurun_out/parity_r04.txt $O/ 2>/dev/null; cp gpurun_out/parity_achieved.txt $O/ 2>/dev/null
grep "headline parity" gpurun_out/parity_r04.txt | grep -v "step [0-9]" | cut -c1-250
exit 0
