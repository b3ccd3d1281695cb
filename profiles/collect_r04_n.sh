#!/bin/bash
# round 4, call n: device tests, then kernel-trace timelines of one steady step of three workloads (ClickBench uniform, 20 M groups, sort)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-n}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
tl() {   # name, workload, first kernel of a step
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1 -- python3 $GRAFT_REPO_ROOT/bench_workloads.py --only $2 --sf 100 --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$O/r04_${T}_$1.json 2> /dev/null || return 1
  python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $(ls $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1/*/*kernel_trace.csv | head -1) $3 > $GRAFT_REPO_ROOT/$O/r04_${T}_timeline_$1.txt
  head -1 $GRAFT_REPO_ROOT/$O/r04_${T}_timeline_$1.txt; rm -rf $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1
}
tl cbu clickbench_uniform_1000000 k_dict_predicate && tl gb20 groupby_int64_unclustered_20000000 k_pa_sample && tl sort sort k_pk_minmax_fold
