#!/bin/bash
# round 4, call o: device tests; the sort workload with the one-sweep passes off / 8 / 16 rows per lane; timelines of sort, 20 M groups, ClickBench uniform
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-o}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
for r in 0 8 16; do timeout -k 10 200 python3 bench_workloads.py --only sort --sf 100 --steps 5 --warmup 2 --option sort_onesweep_rows=$r 2> /dev/null | tail -1 | cut -c1-400 | tee $O/r04_${T}_sort_onesweep_$r.json || exit 1; done
cd /tmp && export TMPDIR=/tmp
tl() {   # name, workload, first kernel of a step
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1 -- python3 $GRAFT_REPO_ROOT/bench_workloads.py --only $2 --sf 100 --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$O/r04_${T}_$1.json 2> /dev/null || return 1
  python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $(ls $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1/*/*kernel_trace.csv | head -1) $3 > $GRAFT_REPO_ROOT/$O/r04_${T}_timeline_$1.txt
  head -1 $GRAFT_REPO_ROOT/$O/r04_${T}_timeline_$1.txt; rm -rf $GRAFT_REPO_ROOT/$O/r04_${T}_trace_$1
}
tl sort sort k_pk_minmax_fold && tl gb20 groupby_int64_unclustered_20000000 k_pa_sample && tl cbu clickbench_uniform_1000000 k_dict_predicate
cd "$GRAFT_REPO_ROOT"; for w in sort gb20 cbu; do python3 -c "
import json,sys
for l in open('$O/r04_${T}_$w.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d.get('workload'), d.get('ms_per_step'), d.get('host_syncs_per_step'), d.get('result_check',{}).get('ok'))
"; done
