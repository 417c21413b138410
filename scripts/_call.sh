cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export SLIM_GPU_TRACE=1
for v in r02 dpp pipe prec cur r02; do
  lib=$PWD/variants/libslim_$v.so; [ $v = cur ] && lib=$PWD/slim_amd/libslim.so
  echo "## c4 default $v"; SLIM_AMD_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep -E "trace\] tiles" | cut -c1-330
  echo "## c5 4096 $v"; SLIM_AMD_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --workload c5 --batch 4096 2>&1 | grep -E "trace\] tiles" | cut -c1-330
done > gpurun_out/r03/call9_variants.txt 2>&1
cat gpurun_out/r03/call9_variants.txt
