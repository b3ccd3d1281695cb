#!/bin/bash
# round 4, call e: device tests, then the driver's command (default run with every nested workload; host read-backs by cause in bench_detail.json)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-e}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_${T}_stdout.txt 2> $O/r04_${T}_stderr.txt; rc=$?
cp bench_detail.json $O/r04_${T}_bench_detail.json; tail -c 2600 $O/r04_${T}_stdout.txt; tail -5 $O/r04_${T}_stderr.txt | cut -c1-600; exit $rc
