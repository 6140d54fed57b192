set -x
python -m pytest tests/test_sim_setup.py -x -q --tb=short 2>&1 | grep -v "^--" | tail -8
python bench.py 2>&1 | tail -1
PFFDTD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 4 --repeats 2 --size 512 2>&1 | tail -1
