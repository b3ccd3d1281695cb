# rocprofv3 passes behind profiles/r03_*: kernel stats (Q3, hash join workload) and the two PMC passes each (FETCH_SIZE / WRITE_SIZE, separate runs).
# Run on the GPU box from the repo root: bash profiles/collect_r03.sh   (writes gpurun_out/prof_r03/)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/q3_stats -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 10 --warmup 3 > $O/q3_stats.log 2>&1; echo "q3 stats rc=$?"
rocprofv3 --kernel-trace --stats -d $O/hj_stats -o hj --output-format csv -- python3 $R/bench_workloads.py --only hash_join_plain --sf 100 > $O/hj_stats.log 2>&1; echo "hj stats rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $O/hj_fetch -o hj --output-format csv -- python3 $R/bench_workloads.py --only hash_join_plain --sf 100 > $O/hj_fetch.log 2>&1; echo "hj fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $O/hj_write -o hj --output-format csv -- python3 $R/bench_workloads.py --only hash_join_plain --sf 100 > $O/hj_write.log 2>&1; echo "hj write rc=$?"
rocprofv3 --kernel-trace --stats -d $O/gd_stats -o gd --output-format csv -- python3 $R/bench_workloads.py --only groupby_decimal_3key,groupby_int64 --sf 100 > $O/gd_stats.log 2>&1; echo "groupby stats rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $O/q3_fetch -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 3 --warmup 2 > $O/q3_fetch.log 2>&1; echo "q3 fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $O/q3_write -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 3 --warmup 2 > $O/q3_write.log 2>&1; echo "q3 write rc=$?"
cd $R; python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd $O; find . -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
# keep what travels back small: the stats csv as is, the counter csvs reduced to the dfgpu kernels
for d in hj_fetch hj_write q3_fetch q3_write; do f=$(find $d -name "*counter_collection.csv" | head -1); head -1 $f > $d.csv; grep "dfgpu::" $f >> $d.csv || true; done
for d in hj_stats q3_stats gd_stats; do cp $(find $d -name "*kernel_stats.csv" | head -1) $d.csv; done
rm -rf hj_fetch hj_write q3_fetch q3_write hj_stats q3_stats gd_stats
ls -la
