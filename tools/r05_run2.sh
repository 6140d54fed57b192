cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05p
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05p/fcc -o fcc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fcc --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-rigid-run --no-selfcheck > $GRAFT_REPO_ROOT/gpurun_out/r05p/fcc_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05p/fcc_bench.err )
export PFFDTD_VERBOSE=1
for v in 0 40; do
  timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 --variant $v >> gpurun_out/r05p/rooms.jsonl 2>> gpurun_out/r05p/rooms.err
  timeout 600 python tools/run_config.py ctk_cart_gpu --steps 200 --variant $v >> gpurun_out/r05p/rooms.jsonl 2>> gpurun_out/r05p/rooms.err
done
find gpurun_out/r05p -name "*kernel_stats*" | head
