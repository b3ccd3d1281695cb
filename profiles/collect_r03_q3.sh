# the three Q3 passes of collect_r03.sh alone (kernel stats + FETCH_SIZE / WRITE_SIZE), into gpurun_out/prof_r03q/ -- for when only they need repeating
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03q; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/q3_stats -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 10 --warmup 3 > $O/q3_stats.log 2>&1; echo "q3 stats rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $O/q3_fetch -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 3 --warmup 2 > $O/q3_fetch.log 2>&1; echo "q3 fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $O/q3_write -o q3 --output-format csv -- python3 $R/bench.py --no-workloads --no-cpu-baseline --no-shuffled --steps 3 --warmup 2 > $O/q3_write.log 2>&1; echo "q3 write rc=$?"
cd $O
for d in q3_fetch q3_write; do f=$(find $d -name "*counter_collection.csv" | head -1); head -1 $f > $d.csv; grep "dfgpu::" $f >> $d.csv || true; done
cp $(find q3_stats -name "*kernel_stats.csv" | head -1) q3_stats.csv
rm -rf q3_fetch q3_write q3_stats
ls -la
