for i in 1 2 3; do PFFDTD_VERBOSE=1 python bench.py --no-cpu-baseline 2>gpurun_out/pl_err_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['grid_placement'], d['rigid_walls']['value'], d['rigid_walls']['autotune_ms_per_step'])"; grep "grid placement" gpurun_out/pl_err_$i.log | cut -c1-250; done
