python -m pytest tests/test_hip_tb2.py tests/test_hip_parity.py tests/test_hip_fullsize.py -x -q --tb=short -k "not fcc and not reference_configurations and not 1536" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 0x40000000; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_z$d -o s --output-format csv -- python $R/bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-rigid-run --debug $d 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug $d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/p_z$d/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("zstrip", "k_boundary", "k_fd_sel")): print("   ", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
