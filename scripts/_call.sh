cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -x -q -m gpu > gpurun_out/r03/call5_tests.txt 2>&1
tail -3 gpurun_out/r03/call5_tests.txt
export SLIM_GPU_TRACE=1
timeout 1200 python scripts/warm_ab.py --workload c5 --variants row:1:1:1,row:1:0:1,row:1:1:0 > gpurun_out/r03/call5_warm_ab.txt 2>&1
grep -E "^\{|trace\] tiles" gpurun_out/r03/call5_warm_ab.txt | cut -c1-330
