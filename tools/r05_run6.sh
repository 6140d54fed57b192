cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 2400 python -m pytest tests/test_hip_tb2.py tests/test_hip_parity.py tests/test_hip_edge_cases.py tests/test_hip_autotune.py -x -q > gpurun_out/r05f/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05f/t.log
tail -30 gpurun_out/r05f/t.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/r05f/b3.json 2> gpurun_out/r05f/b3.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-pmc --numerics 2 > gpurun_out/r05f/b3sg.json 2> gpurun_out/r05f/b3sg.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-pmc --precision double > gpurun_out/r05f/b3d.json 2> gpurun_out/r05f/b3d.err
python - <<'PY'
import json
for f in ('b3','b3sg','b3d'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r05f/{f}.json') if l.startswith('{')][-1])
        r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r['kernel_instantiation'], r['kernel_ms_per_launch'], r.get('autotune_ms_per_step'), d.get('selfcheck',{}).get('family_agreement'), d.get('rigid_walls',{}).get('value'))
    except Exception as e: print(f,'ERR',e)
PY
