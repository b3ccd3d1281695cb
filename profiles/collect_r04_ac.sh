#!/bin/bash
# round 4, call ac: device tests; the Q3 shares (clustered / general / shuffled) on the final code
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-ac}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
bash profiles/collect_r04.sh $T shares
