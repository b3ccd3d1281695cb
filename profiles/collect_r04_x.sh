#!/bin/bash
# round 4, call x: device tests; the workloads the last changes touch; the driver's command once more
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-x}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-420 | tee $O/r04_${T}_$3.json || exit 1; }
wl sort "" sort
wl clickbench_uniform_1000000 "" cbu
wl clickbench_zipf_1000000 "" cbz
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_${T}_stdout.txt 2> $O/r04_${T}_stderr.txt; rc=$?
cp bench_detail.json $O/r04_${T}_bench_detail.json; tail -c 2200 $O/r04_${T}_stdout.txt
exit $rc
