#!/bin/bash
# round 4, call m: device tests, the shares pass, the driver's command, a kernel trace of the SF12.5 step
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-m}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
bash profiles/collect_r04.sh $T shares
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_${T}_stdout.txt 2> $O/r04_${T}_stderr.txt; rc=$?
cp bench_detail.json $O/r04_${T}_bench_detail.json; tail -c 2600 $O/r04_${T}_stdout.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r04_${T}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --sf 12.5 --steps 20 --warmup 5 --no-workloads --no-shuffled --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r04_${T}_sf12_stdout.txt 2> /dev/null
cd "$GRAFT_REPO_ROOT"; python3 profiles/step_timeline.py $(ls $O/r04_${T}_trace/*/*kernel_trace.csv | head -1) > $O/r04_${T}_timeline.txt; head -3 $O/r04_${T}_timeline.txt; rm -rf $O/r04_${T}_trace
exit $rc
