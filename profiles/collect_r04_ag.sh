#!/bin/bash
# round 4, call ag (final code): device tests, the driver's command
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-ag}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_${T}_stdout.txt 2> $O/r04_${T}_stderr.txt; rc=$?
cp bench_detail.json $O/r04_${T}_bench_detail.json; tail -c 2200 $O/r04_${T}_stdout.txt
exit $rc
