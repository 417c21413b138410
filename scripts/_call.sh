cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r03/final_tests.txt 2>&1
tail -2 gpurun_out/r03/final_tests.txt
timeout 1500 bash scripts/collect_profiles.sh r03_c4 > gpurun_out/r03/collect_c4.log 2>&1
tail -12 gpurun_out/r03/collect_c4.log | cut -c1-600
timeout 1200 bash scripts/collect_profiles.sh r03_c5 --workload c5 --batch 4096 > gpurun_out/r03/collect_c5.log 2>&1
tail -12 gpurun_out/r03/collect_c5.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_warm_pmc_$c -o run -- python $GRAFT_REPO_ROOT/scripts/warm_ab.py --columns 4096 --variants row:1:0:1 > $GRAFT_REPO_ROOT/gpurun_out/r03/warm_pmc_$c.txt 2>&1
done
cd $GRAFT_REPO_ROOT
grep -h "cd_tile" gpurun_out/r03_warm_pmc_*/run_counter_collection.csv | cut -c1-300
grep "^{" gpurun_out/r03/warm_pmc_FETCH_SIZE.txt | cut -c1-250
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1500 python bench.py --steps 2 --warmup 1 > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
tail -1 gpurun_out/r03/bench_default.json | cut -c1-3000
