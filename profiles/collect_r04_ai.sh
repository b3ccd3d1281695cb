#!/bin/bash
# round 4, call ai: device tests; the group-by shapes with the wave combine (DPP reductions, up to three candidate keys) from 16 / 8 / 4 lanes on
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-ai}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-300 | tee $O/r04_${T}_$3.json || exit 1; }
for m in 16 8 4; do wl clickbench_zipf_1000000 "--option agg_combine_min_lanes=$m" cbz_$m; done
for m in 16 4; do wl clickbench_uniform_1000000 "--option agg_combine_min_lanes=$m" cbu_$m; done
for m in 16 4; do wl groupby_int64_unclustered_20000000 "--option agg_combine_min_lanes=$m" gb20_$m; done
