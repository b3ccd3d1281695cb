#!/bin/bash
# round 4, call r: device tests; sort with / without the fused finish; the three group-by shapes; Q3 at SF100 and SF12.5 (default and with the packed one-sweep sort from 64 K rows)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-r}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-520 | tee $O/r04_${T}_$3.json || exit 1; }
wl sort "" sort_default
wl groupby_int64_unclustered_20000000 "" gb20
wl clickbench_uniform_1000000 "" cbu
wl clickbench_zipf_1000000 "" cbz
for sf in 100; do timeout -k 10 300 python3 bench.py --sf $sf --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --no-shuffled --detail $O/r04_${T}_q3_sf$sf.json 2> /dev/null | tail -1 | cut -c1-300 || exit 1; done
