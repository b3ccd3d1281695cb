#!/bin/bash
# round 4, call i: device tests, then the evidence passes of collect_r04.sh (tag and passes as arguments)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=$1; shift
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
bash profiles/collect_r04.sh $T "$@"
