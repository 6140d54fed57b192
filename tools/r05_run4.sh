cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 600 python -m pytest tests/test_hip_multi.py -x -q -k "hung or fall or host" > gpurun_out/r05d/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05d/t.log
tail -3 gpurun_out/r05d/t.log
timeout 900 python tools/tb3_probe.py 1024 > gpurun_out/r05d/tb3.log 2>&1
tail -25 gpurun_out/r05d/tb3.log
for dbg in 0 0x80000 0x4000000 0x4080000; do
  timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 --variant 40 --debug $dbg 2>> gpurun_out/r05d/rooms.err | grep '^{' >> gpurun_out/r05d/rooms.jsonl
done
timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 --debug 0x4000 2>> gpurun_out/r05d/rooms.err | grep '^{' >> gpurun_out/r05d/rooms.jsonl
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05d/mv -o mv --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_config.py mv_fcc_gpu --steps 60 --variant 40 --debug 0x4000000 > /dev/null 2>&1 )
