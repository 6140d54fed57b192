cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 3 0; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_s$r -o s --output-format csv -- python $R/bench.py --emulate-slab $r/8 --emulate-transport rccl --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep -E "emulated|metric" | cut -c1-260
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/p_s$r/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:10]: print("  ", r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
done
cd $R; python - <<'PY'
import time, sys
sys.path.insert(0, '.')
import bench
from pffdtd_amd import engine
sd = bench.build_scene(1024, 400, "single", False, True, 11)
t0 = time.perf_counter(); el = engine.run_sim(sd); t1 = time.perf_counter()
print(f"pf_run_sim 1024^3 x 400 steps from host buffers: loop {el:.3f} s ({sd.Npts*400/el/1e9:.1f} Gvox/s), whole call {t1-t0:.3f} s ({sd.Npts*400/(t1-t0)/1e9:.1f} Gvox/s incl. upload, list sorting, creation-time measurement, teardown)")
PY
