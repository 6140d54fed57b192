#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05ab; mkdir -p $O; rm -f $O/ab.txt
timeout 1500 python -m pytest tests/test_hip_tb2.py -x -q 2>&1 | tail -4
P='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("   %.1f Gvox/s  %.4f ms/step (min %.4f max %.4f)  kernel %.3f ms; rigid %s; selfcheck %s" % (d["value"], d["ms_per_step"], d.get("ms_per_step_min",0), d.get("ms_per_step_max",0), d["roofline"]["kernel_ms_per_launch"], (d.get("rigid_walls") or {}).get("value"), (d.get("selfcheck") or {}).get("family_agreement")))'
for rep in 1 2; do
  echo "7-pt K=48" >> $O/ab.txt
  timeout 300 python bench.py --steps 48 --warmup 6 --repeats 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "$P" >> $O/ab.txt
  echo "7-pt K=20" >> $O/ab.txt
  timeout 300 python bench.py --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c "$P" >> $O/ab.txt
done
echo "fp64" >> $O/ab.txt
timeout 300 python bench.py --precision double --steps 24 --warmup 6 --repeats 3 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c "$P" >> $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/st -o s --output-format csv -- python $R/bench.py --steps 12 --warmup 6 --repeats 1 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc > /dev/null 2>&1
grep k_wall2 $R/$O/st/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
rm -rf $R/$O/st
