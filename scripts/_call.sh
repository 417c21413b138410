cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1800 python bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r03/bench_10steps.json 2> gpurun_out/r03/bench_10steps.err
tail -1 gpurun_out/r03/bench_10steps.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_physical'], d['cpu_baseline']['value'], d['cpu_baseline']['parity'])"
export SLIM_GPU_TRACE=1
timeout 1200 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --scaling strong --batch 0 > gpurun_out/r03/c4_whole_matrix.json 2> gpurun_out/r03/c4_whole_matrix.err
grep "trace\] tiles" gpurun_out/r03/c4_whole_matrix.err | cut -c1-330; cut -c1-400 gpurun_out/r03/c4_whole_matrix.json
unset SLIM_GPU_TRACE
STEPS=1 timeout 1200 bash scripts/collect_profiles.sh r03_c401 --workload c4-0.1pct --scaling strong --batch 0 > gpurun_out/r03/collect_c401.log 2>&1
tail -14 gpurun_out/r03/collect_c401.log | cut -c1-500
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
