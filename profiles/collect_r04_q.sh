#!/bin/bash
# round 4, call q: the sort tests; the sort workload with the status rows walked 4 bytes per lane by 256 lanes / 16 bytes per lane by one wave, 4096- and 8192-row tiles
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-q}
timeout -k 10 600 python3 -m pytest tests/test_gpu_sort.py -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
wl() { timeout -k 10 200 python3 bench_workloads.py --only $1 --sf 100 --steps 5 --warmup 2 $2 2> /dev/null | tail -1 | cut -c1-420 | tee $O/r04_${T}_$3.json || exit 1; }
wl sort "--option sort_onesweep_wide_status=1" sort_wide16
wl sort "--option sort_onesweep_wide_status=1 --option sort_onesweep_rows=8" sort_wide8
wl sort "" sort_default
