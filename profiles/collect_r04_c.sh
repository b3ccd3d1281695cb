#!/bin/bash
# round 4, call c: device tests, Q3 at SF12.5 / SF100 with the selection-carrying join output on and off, kernel trace of the SF12.5 step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=${1:-c}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_${T}_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_${T}_pytest.log
tail -3 $O/r04_${T}_pytest.log
for sf in 12.5 100; do for so in 1 0; do
  timeout -k 10 300 python3 bench.py --sf $sf --steps 40 --warmup 5 --no-workloads --no-shuffled --no-cpu-baseline --option join_selection_output=$so --detail $O/r04_${T}_detail_sf${sf}_so${so}.json 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sf $sf selection_output $so', d['ms_per_step'], 'ms/step; syncs', d.get('host_syncs_per_step'), '; general', d['ms_per_step_other'])" || exit 1
done; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r04_${T}_trace -- python3 bench.py --sf 12.5 --steps 20 --warmup 5 --no-workloads --no-shuffled --no-cpu-baseline > $O/r04_${T}_sf12_stdout.txt 2> $O/r04_${T}_sf12_stderr.txt &&
python3 profiles/step_timeline.py $(ls $O/r04_${T}_trace/*/*kernel_trace.csv | head -1) > $O/r04_${T}_timeline.txt && head -5 $O/r04_${T}_timeline.txt
