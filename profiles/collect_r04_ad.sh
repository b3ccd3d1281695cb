#!/bin/bash
# round 4, call ad: device tests; the group-by shapes with the hot keys taken out before the partition passes / left in
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-ad}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-520 | tee $O/r04_${T}_$3.json || exit 1; }
wl clickbench_zipf_1000000 "" cbz_hot_keys_out
wl clickbench_zipf_1000000 "--option agg_hot_keys=0" cbz_hot_keys_in
wl clickbench_uniform_1000000 "" cbu
wl groupby_int64_unclustered_20000000 "" gb20
wl groupby_int64_unclustered_1000000 "" gb1
