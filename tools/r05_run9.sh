cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
timeout 1800 python -m pytest tests/test_hip_tb2.py tests/test_hip_fullsize.py -q -k "geometry_inside or fullsize or test_hip_fullsize" > gpurun_out/r05i/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05i/t.log
tail -5 gpurun_out/r05i/t.log
# experiment: wall launches beside k_tb3
for dbg in 0 0x10000; do
  timeout 600 python bench.py --steps 21 --warmup 6 --no-cpu-baseline --no-rigid-run --no-pmc --no-selfcheck --debug $dbg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $dbg', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])"
done
COOL_S=20 bash tools/collect_n1_profile.sh r05_bench_n1 > gpurun_out/r05i/collect.log 2>&1
tail -3 gpurun_out/r05i/collect.log | cut -c1-1500
