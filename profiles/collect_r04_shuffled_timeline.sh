set -o pipefail
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_shuf_trace -- python3 $GRAFT_REPO_ROOT/profiles/q3_shuffled_run.py --sf 12.5 --steps 12 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r04_shuf_sf12.5.json 2> /dev/null
cd $GRAFT_REPO_ROOT; python3 profiles/step_timeline.py $(ls gpurun_out/r04_shuf_trace/*/*kernel_trace.csv | head -1) k_dict_predicate > gpurun_out/r04_shuf_timeline_sf12.5.txt; head -1 gpurun_out/r04_shuf_timeline_sf12.5.txt; rm -rf gpurun_out/r04_shuf_trace; tail -1 gpurun_out/r04_shuf_sf12.5.json | cut -c1-300
