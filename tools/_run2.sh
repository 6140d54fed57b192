set -x
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_fcc -o s --output-format csv -- python $R/bench.py --fcc --steps 30 --warmup 6 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep '"metric"' > $R/gpurun_out/fcc.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_fcc64 -o s --output-format csv -- python $R/bench.py --fcc --precision double --size 1536 --steps 12 --warmup 4 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep '"metric"' > $R/gpurun_out/fcc64.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_fccs -o s --output-format csv -- python $R/bench.py --fcc --steps 30 --warmup 6 --repeats 3 --no-cpu-baseline --no-rigid-run --debug 0x4000 2>&1 | grep '"metric"' > $R/gpurun_out/fccs.json
cat $R/gpurun_out/fcc.json $R/gpurun_out/fcc64.json $R/gpurun_out/fccs.json | cut -c1-180
