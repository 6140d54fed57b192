for wy in 4 8 4 8; do
PFFDTD_TB2_WY=$wy python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wy $wy', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
done
PFFDTD_TB2_WY=8 python -m pytest tests/test_hip_tb2.py -x -q -k "not fcc" 2>&1 | tail -2
