cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_tb2.py -x -q -k "fcc or exchanged" > gpurun_out/r05_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t1.log
for cfg in "" "--debug 0x40000" "--numerics 2"; do
  timeout 600 python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run $cfg >> gpurun_out/r05_fcc_ab.jsonl 2>> gpurun_out/r05_fcc_ab.err
done
timeout 900 python bench.py --fcc --size 1536 --precision double --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-rigid-run >> gpurun_out/r05_fcc_ab.jsonl 2>> gpurun_out/r05_fcc_ab.err
tail -3 gpurun_out/r05_t1.log
