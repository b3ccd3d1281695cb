#!/bin/bash
# The rocprofv3 passes behind profiles/r04_<tag>_*: kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, nothing else traced with them) for
#   q3      bench.py's Q3 step at SF100 (clustered tables)               q3s     the same query over row-wise permuted tables (profiles/q3_shuffled_run.py)
#   gb      bench_workloads.py groupby_int64 (20 M groups) and the two clickbench shapes, one process each         hj      bench_workloads.py hash_join (sparse keys)
#   sort    bench_workloads.py sort (two keys, three columns out)
# Run on the GPU box from the repo root:  bash profiles/collect_r04.sh <tag> [passes...]   (default: all).  Writes gpurun_out/prof_r04_<tag>/, reduced to what travels back.
set -o pipefail
R=$GRAFT_REPO_ROOT; T=${1:-x}; shift; PASSES=${@:-q3 q3s gb hj shares}
O=$R/gpurun_out/prof_r04_$T; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run3() {   # name, program and its arguments: kernel stats, FETCH_SIZE, WRITE_SIZE
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/${name}_stats -o $name --output-format csv -- python3 "$@" > $O/${name}_stats.log 2>&1; echo "$name stats rc=$?"
  rocprofv3 --pmc FETCH_SIZE -d $O/${name}_fetch -o $name --output-format csv -- python3 "$@" > $O/${name}_fetch.log 2>&1; echo "$name fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE -d $O/${name}_write -o $name --output-format csv -- python3 "$@" > $O/${name}_write.log 2>&1; echo "$name write rc=$?"
  for d in ${name}_fetch ${name}_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then head -1 $f > $O/$d.csv; grep "dfgpu::" $f >> $O/$d.csv || true; fi; done
  f=$(find $O/${name}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_stats.csv
  rm -rf $O/${name}_fetch $O/${name}_write $O/${name}_stats
}
for p in $PASSES; do case $p in
  q3)  run3 q3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 10 --warmup 3 --detail $O/q3_detail.json ;;
  q3s) run3 q3s $R/profiles/q3_shuffled_run.py --sf 100 --steps 5 --warmup 2 ;;
  gb)  run3 gb20 $R/bench_workloads.py --only groupby_int64_unclustered_20000000 --sf 100        # one workload per process: kernel names repeat across them
       run3 cbu $R/bench_workloads.py --only clickbench_uniform_1000000 --sf 100
       run3 cbz $R/bench_workloads.py --only clickbench_zipf_1000000 --sf 100 ;;
  hj)  run3 hj $R/bench_workloads.py --only hash_join_plain --sf 100 ;;
  sort) run3 sort $R/bench_workloads.py --only sort --sf 100 ;;
  shares)   # one rank's share of the Q3 step at 8 / 4 / 2 GPUs when nothing has to move: clustered, general paths, shuffled
    for sf in 12.5 25 50; do python3 $R/bench.py --sf $sf --steps 20 --warmup 5 --no-workloads --no-cpu-baseline --detail $O/share_sf$sf.json 2> /dev/null | tail -1 > $O/share_sf$sf.line.json; echo "share sf $sf rc=$?"; done ;;
esac; done
cd $O; ls -la; tail -q -n 2 *_stats.log 2>/dev/null | cut -c1-400; (cat share_sf*.line.json 2>/dev/null | cut -c1-700) || true
