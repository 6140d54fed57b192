for spec in 0/2 1/2 0/4 1/4 0/8 3/8; do
python bench.py --emulate-slab $spec --emulate-transport rccl --steps 60 --warmup 10 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep -E "emulated" | cut -c1-200
done
