#!/bin/bash
# round 4, call b: device tests after the mailbox read-back / order statistics, then the SF12.5 and SF100 Q3 step with the mailbox on and off
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_b_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_b_pytest.log
tail -3 $O/r04_b_pytest.log
for sf in 12.5 100; do for mb in 1 0; do
  timeout -k 10 300 python3 bench.py --sf $sf --steps 40 --warmup 5 --no-workloads --no-shuffled --no-cpu-baseline --option mailbox_readback=$mb --detail $O/r04_b_detail_sf${sf}_mb${mb}.json 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sf $sf mailbox $mb', d['ms_per_step'], 'ms/step; syncs', d.get('host_syncs_per_step'), '; general', d['ms_per_step_other'])" || exit 1
done; done
