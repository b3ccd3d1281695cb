#!/bin/bash
# round 4, first GPU call: device tests, the default bench run (the driver's command), and a kernel trace of the SF12.5 Q3 step (one rank's share at 8 GPUs)
# for profiles/gap_analysis.py.  Run from the repo root on the GPU box:  bash profiles/collect_r04_a.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r04_a_pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/r04_a_pytest.log
tail -3 $O/r04_a_pytest.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_a_stdout.txt 2> $O/r04_a_stderr.txt && cp bench_detail.json $O/r04_a_bench_detail.json && tail -c 2500 $O/r04_a_stdout.txt &&
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r04_a_trace -- python3 bench.py --sf 12.5 --steps 20 --warmup 5 --no-workloads --no-shuffled --no-cpu-baseline > $O/r04_a_sf12_stdout.txt 2> $O/r04_a_sf12_stderr.txt &&
tail -c 1200 $O/r04_a_sf12_stdout.txt && python3 profiles/gap_analysis.py $(ls $O/r04_a_trace/*/*kernel_trace.csv | head -1) --skip-first-frac 0.4 > $O/r04_a_gaps.txt && cat $O/r04_a_gaps.txt
