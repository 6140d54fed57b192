for c in 16 12 20 24 32 16; do
PFFDTD_TB2_CHUNK=$c python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
done
