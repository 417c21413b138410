set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export SLIM_GPU_TRACE=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -m gpu > gpurun_out/r03/call1_tests.txt 2>&1
tail -5 gpurun_out/r03/call1_tests.txt
timeout 900 python scripts/warm_ab.py --workload c5 > gpurun_out/r03/call1_warm_ab.txt 2>&1
grep -E "^\{|trace\] tiles" gpurun_out/r03/call1_warm_ab.txt | cut -c1-330
for lib in variants/libslim_r02.so slim_amd/libslim.so; do
  echo "## c4 default $lib"
  SLIM_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep -E "trace\] tiles|^\{" | cut -c1-400
  echo "## c5 4096 $lib"
  SLIM_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c5 --batch 4096 2>&1 | grep -E "trace\] tiles|^\{" | cut -c1-400
done > gpurun_out/r03/call1_ab.txt 2>&1
cat gpurun_out/r03/call1_ab.txt
