#!/bin/bash
# round 4, call t: device tests; ClickBench shapes with the one-workgroup small sort and without the selection count; timeline of the uniform shape
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-t}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-620 | tee $O/r04_${T}_$3.json || exit 1; }
wl clickbench_uniform_1000000 "" cbu
wl clickbench_zipf_1000000 "" cbz
wl clickbench_uniform_1000000 "--option sort_one_block_max_rows=0" cbu_two_launch_passes
wl q1_float64 "" q1f
wl q5 "" q5
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r04_${T}_trace -- python3 $GRAFT_REPO_ROOT/bench_workloads.py --only clickbench_uniform_1000000 --sf 100 --steps 6 --warmup 2 > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; python3 profiles/step_timeline.py $(ls $O/r04_${T}_trace/*/*kernel_trace.csv | head -1) k_dict_predicate > $O/r04_${T}_timeline_cbu.txt; head -1 $O/r04_${T}_timeline_cbu.txt; rm -rf $O/r04_${T}_trace
