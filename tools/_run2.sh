set -x
python -m pytest tests/test_sim_setup.py -x -q --tb=short 2>&1 | grep -v "^--" | tail -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/p2 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/p2/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pf::" in r["Name"]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
