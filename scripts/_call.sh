cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
export SLIM_GPU_TRACE=1
for rep in 1 2; do for v in nopf cur; do
  lib=$PWD/variants/libslim_$v.so; [ $v = cur ] && lib=$PWD/slim_amd/libslim.so
  echo "## 0.1pct default $v"; SLIM_AMD_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c4-0.1pct 2>&1 | grep -E "trace\] tiles" | cut -c1-330
  echo "## 0.1pct 32768 $v"; SLIM_AMD_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c4-0.1pct --batch 32768 2>&1 | grep -E "trace\] tiles" | cut -c1-330
  echo "## c4 default $v"; SLIM_AMD_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep -E "trace\] tiles" | cut -c1-330
done; done > gpurun_out/r03/call12_pf.txt 2>&1
cat gpurun_out/r03/call12_pf.txt
