cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 1800 python -m pytest tests/test_hip_tb2.py -x -q -k "three_steps" > gpurun_out/r05e/t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05e/t3.log
tail -30 gpurun_out/r05e/t3.log
export PFFDTD_VERBOSE=1
timeout 900 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-rigid-run --no-pmc > gpurun_out/r05e/b3.json 2> gpurun_out/r05e/b3.err
timeout 900 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-rigid-run --no-pmc --debug 0x20000 > gpurun_out/r05e/b2.json 2> gpurun_out/r05e/b2.err
tail -c 2500 gpurun_out/r05e/b3.json; echo; grep -v amdgpu.ids gpurun_out/r05e/b3.err | tail -15
