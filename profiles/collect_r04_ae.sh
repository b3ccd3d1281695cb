#!/bin/bash
# round 4, call ae: device tests (partitioned join with up to 4096 partitions); the hash-join workloads (unchanged paths)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-ae}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench_workloads.py --only hash_join --sf 100 --steps 5 --warmup 2 2> /dev/null | cut -c1-330 | tee $O/r04_${T}_hash_join.json
