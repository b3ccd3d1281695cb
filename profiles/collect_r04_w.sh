#!/bin/bash
# round 4, call w: device tests; the sort workload with the payload column gathered by the last pass / by a take() afterwards; a two-rank rehearsal of bench.py on one device (gloo)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-w}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-520 | tee $O/r04_${T}_$3.json || exit 1; }
wl sort "" sort_payload_in_last_pass
wl sort "--option sort_payload_in_last_pass=0" sort_payload_by_take
wl q18 "" q18
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --one-device --sf 10 --steps 3 --warmup 1 --no-cpu-baseline > $O/r04_${T}_two_ranks_stdout.txt 2> $O/r04_${T}_two_ranks_stderr.txt; echo "two ranks rc $?"; tail -1 $O/r04_${T}_two_ranks_stdout.txt | cut -c1-400
