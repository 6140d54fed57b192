cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
timeout 1200 python -m pytest tests/test_hip_tb2.py tests/test_hip_multi.py -q -k "fcc" 2>&1 | tail -4
for dbg in 0 0x80000; do
  timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 --variant 40 --debug $dbg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MV dbg $dbg', d['gvox_per_s'], d['ms_per_step'], d['air_ms_per_step'], d['blocked_cell_fraction'])"
done
timeout 900 python tools/run_config.py mv_fcc_gpu --steps 100 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MV auto', d['gvox_per_s'], d['ms_per_step'], d['tune_ms'], d['blocked_cell_fraction'])"
timeout 600 python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-pmc 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fcc1024', d['value'], d['ms_per_step'], d['selfcheck']['family_agreement'])"
