cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
timeout 900 python tools/placement_offsets.py 1024 5 > gpurun_out/r05m/placement.txt 2>&1
cat gpurun_out/r05m/placement.txt | grep -v amdgpu
timeout 1200 python -m pytest tests/test_hip_tb2.py tests/test_hip_parity.py -q -x 2>&1 | tail -3
