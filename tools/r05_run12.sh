cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05l
export TMPDIR=/tmp
for spec in 0/8 3/8 1/4 0/2; do
  timeout 600 python bench.py --emulate-slab $spec --emulate-transport rccl --steps 42 --warmup 6 --repeats 5 --no-pmc 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['emulated_slab']; print('N=%d rank %d planes %d pairs=%s: %.4f ms/step -> %.0f Gvox/s; kernel %s %.3f ms' % (e['of'], e['rank'], e['planes'][1]-e['planes'][0], e['pairs'], e['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms_per_launch']))"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05l/r38 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --emulate-slab 3/8 --emulate-transport copy --steps 42 --warmup 6 --repeats 2 --no-pmc > /dev/null 2>&1 )
