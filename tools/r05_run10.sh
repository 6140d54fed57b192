cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
timeout 3000 python -m pytest tests/test_hip_multi.py tests/test_hip_slabs.py tests/test_hip_multiproc.py tests/test_bench_cmd.py tests/test_hip_refbinding.py -q > gpurun_out/r05j/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05j/t.log
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r05j/t.log | head -40
