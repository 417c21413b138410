cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python scripts/debug_hi.py 2>&1 | grep -v "^\[trace" | head -8
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03/call4_tests.txt 2>&1
tail -3 gpurun_out/r03/call4_tests.txt
