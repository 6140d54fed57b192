cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05h
export PFFDTD_VERBOSE=1
timeout 3000 python -m pytest tests/test_hip_autotune.py tests/test_hip_tb2.py tests/test_hip_slabs.py tests/test_hip_parity.py -q -k "full_device or three_steps or stepper_grids or refusals" > gpurun_out/r05h/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05h/t.log
grep -n "^E  \|^FAILED\|passed\|failed\|bytes of engine" gpurun_out/r05h/t.log | head -40
