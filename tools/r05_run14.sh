cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m; export PFFDTD_VERBOSE=1
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-pmc --no-selfcheck 2> gpurun_out/r05m/b20_$i.err | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=20', d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['roofline']['kernel_ms_per_launch'])"
grep "fifth grid" gpurun_out/r05m/b20_$i.err | head -5
done
timeout 600 python bench.py --steps 48 --warmup 6 --no-cpu-baseline --no-rigid-run --no-pmc --no-selfcheck 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=48', d['value'], d['ms_per_step'])"
