set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r03/call2_tests.txt 2>&1
tail -5 gpurun_out/r03/call2_tests.txt
export SLIM_GPU_TRACE=1
for lib in slim_amd/libslim.so; do
  echo "## c4 default $lib"
  SLIM_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep -E "trace\] tiles|^\{" | cut -c1-400
  echo "## c5 4096 $lib"
  SLIM_AMD_LIB=$PWD/$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c5 --batch 4096 2>&1 | grep -E "trace\] tiles|^\{" | cut -c1-400
done > gpurun_out/r03/call2_ab.txt 2>&1
cat gpurun_out/r03/call2_ab.txt
timeout 900 python scripts/warm_ab.py --workload c5 --variants row:1 > gpurun_out/r03/call2_warm_ab.txt 2>&1
grep -E "^\{|trace\] tiles" gpurun_out/r03/call2_warm_ab.txt | cut -c1-330
