#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05ab; mkdir -p $O; rm -f $O/ab.txt
timeout 1500 python -m pytest tests/test_hip_tb2.py -x -q -k "wall or three or triple" 2>&1 | tail -3
P='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("   %.1f Gvox/s  %.4f ms/step (min %.4f max %.4f)  kernel %.3f ms; rigid %s; selfcheck %s" % (d["value"], d["ms_per_step"], d.get("ms_per_step_min",0), d.get("ms_per_step_max",0), d["roofline"]["kernel_ms_per_launch"], (d.get("rigid_walls") or {}).get("value"), (d.get("selfcheck") or {}).get("family_agreement")))'
for rep in 1 2 3; do
  echo "7-pt K=48" >> $O/ab.txt
  timeout 300 python bench.py --steps 48 --warmup 6 --repeats 5 --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c "$P" >> $O/ab.txt
done
cat $O/ab.txt
bash tools/r05_tl.sh; sed -n 2,14p gpurun_out/r05tl/n1.txt | cut -c1-125
