set -x
python -m pytest tests/test_hip_tb2.py -x -q --tb=short -k fcc 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/p3 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fcc --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-rigid-run --variant 40 2>&1 | grep '"metric"' | cut -c1-200
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/p3/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pf::" in r["Name"]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
python bench.py --fcc --precision double --size 768 --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-rigid-run --variant 40 2>&1 | grep '"metric"' | cut -c1-250
python bench.py --fcc --precision double --size 768 --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-rigid-run --debug 0x4000 2>&1 | grep '"metric"' | cut -c1-250
