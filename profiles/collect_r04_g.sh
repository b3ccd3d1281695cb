#!/bin/bash
# round 4, call g: device tests; the gather cache-policy experiment; Q18 / hash join / group-by workloads after the run-accumulate and consumer-hint changes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-g}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
timeout -k 10 120 ./profiles/experiments/gather_policy_microbench.bin > $O/r04_${T}_gather_policy.txt 2>&1; echo "microbench rc $?"; cat $O/r04_${T}_gather_policy.txt
timeout -k 10 600 python3 bench_workloads.py --sf 100 --only q18,hash_join_dense_unsorted,groupby_int64,clickbench_uniform_1000000 > $O/r04_${T}_workloads.jsonl 2> $O/r04_${T}_workloads.err; echo "workloads rc $?"
python3 - <<'PY'
import json
for l in open('gpurun_out/r04_g_workloads.jsonl'):
    d = json.loads(l); print(d['workload'], d['ms_per_step'], d.get('host_syncs_per_step'), dict(list(d['kernel_ms_per_step'].items())[:6]), d.get('result_check', {}).get('ok'))
PY
