#!/bin/bash
# round 4, call aa: device tests; group-by shapes with the hot key accumulated in registers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-aa}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-420 | tee $O/r04_${T}_$3.json || exit 1; }
wl clickbench_zipf_1000000 "" cbz
wl clickbench_uniform_1000000 "" cbu
wl groupby_int64_unclustered_20000000 "" gb20
wl groupby_int64_unclustered_1000000 "" gb1
wl groupby_decimal_3key "" gb3
