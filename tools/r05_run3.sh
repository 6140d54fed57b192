cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests/test_hip_tb2.py tests/test_hip_multi.py tests/test_hip_parity.py -x -q > gpurun_out/r05c/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05c/t.log
tail -5 gpurun_out/r05c/t.log
for cfg in "" "--debug 0x4000000" "--numerics 2"; do
  timeout 600 python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run $cfg >> gpurun_out/r05c/fcc.jsonl 2>> gpurun_out/r05c/fcc.err
done
export PFFDTD_VERBOSE=1
timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 2>> gpurun_out/r05c/rooms.err | grep '^{' >> gpurun_out/r05c/rooms.jsonl
timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 --debug 0x4000000 2>> gpurun_out/r05c/rooms.err | grep '^{' >> gpurun_out/r05c/rooms.jsonl
timeout 600 python tools/run_config.py ctk_cart_gpu --steps 200 --variant 40 2>> gpurun_out/r05c/rooms.err | grep '^{' >> gpurun_out/r05c/rooms.jsonl
timeout 600 python bench.py --gpus 4 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --transport host >> gpurun_out/r05c/chain.jsonl 2>> gpurun_out/r05c/chain.err
