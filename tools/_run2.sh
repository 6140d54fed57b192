python -m pytest tests/test_hip_multi.py tests/test_hip_slabs.py tests/test_hip_tb2.py -x -q --tb=short 2>&1 | tail -6
