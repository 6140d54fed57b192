cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
timeout 3000 python -m pytest tests/test_hip_slabs.py tests/test_hip_multiproc.py -q -k "blocked_pairs or box_pairs" > gpurun_out/r05k/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05k/t.log
grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/r05k/t.log | head -20
for spec in 0/2 0/4 1/4 0/8 3/8; do
  timeout 600 python bench.py --emulate-slab $spec --emulate-transport rccl --steps 42 --warmup 6 --repeats 5 --no-pmc 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['emulated_slab']; print('N=%d rank %d planes %d pairs=%s: %.4f ms/step -> %.0f Gvox/s; kernel %s %.3f ms' % (e['of'], e['rank'], e['planes'][1]-e['planes'][0], e['pairs'], e['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms_per_launch']))"
done
for n in 2 8; do
  timeout 600 python bench.py --gpus $n --steps 30 --warmup 6 --repeats 3 --transport rccl --no-pmc 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('virtual', d['n_gpus'], d['ms_per_step'], d['value'], d['exchange_verified'], [s['pairs'] for s in d['slabs']])"
done
