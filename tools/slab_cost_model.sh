#!/bin/bash
# usage (GPU box): tools/slab_cost_model.sh > gpurun_out/slab_cost_model.txt
# Per-rank cost model of the z-slab chain on ONE device, through the C chain object: bench.py --emulate-slab r/N instantiates slab
# r of an N-rank cut alone (pf_opts.only_slab) and lets it exchange its own edge planes with itself by peer copies or by a real
# RCCL send / receive through a 1-rank communicator.  Then the whole chains as virtual slabs on the one device.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
P='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); e=d.get("emulated_slab")
        if e: print("   N=%d rank %d (%d planes, pairs=%s): %.4f ms/step -> %.0f Gvox/s if every rank took that long; dominant kernel %s %.3f ms/launch; exchange %s" % (e["of"], e["rank"], e["planes"][1]-e["planes"][0], e["pairs"], e["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms_per_launch"], d["exchange"]["backend"]))
        else: print("   N=%d virtual slabs on one device: %.4f ms/step = %.1f Gvox/s; exchange %s verified=%s; pairs %s" % (d["n_gpus"], d["ms_per_step"], d["value"], d["exchange"]["backend"], d["exchange_verified"], [s["pairs"] for s in d["slabs"]]))'
echo "## single domain (N=1), same build"
python bench.py --steps 42 --warmup 6 --repeats 5 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c 'import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("   N=1: %.4f ms/step = %.1f Gvox/s (%s)" % (d["ms_per_step"], d["value"], d["roofline"].get("shell")))'
for tr in rccl copy; do
  echo "## rank cost model, exchange by $tr"
  for spec in 0/2 1/2 0/4 1/4 2/4 0/8 3/8 4/8 7/8; do  # (2/4, 4/8: the ranks that hold the source)
    python bench.py --emulate-slab $spec --emulate-transport $tr --steps 42 --warmup 6 --repeats 5 --no-pmc 2>/dev/null | python -c "$P"
  done
done
echo "## whole chains as virtual slabs on ONE device (every slab's work on the one GPU: control flow + fixed costs, not scaling)"
for n in 2 4 8; do
  python bench.py --gpus $n --steps 30 --warmup 6 --repeats 3 --transport rccl --no-pmc 2>/dev/null | python -c "$P"
done
echo "## BASELINE configs[4]: 1536^3 13-point folded FCC fp64 -- single domain, then ranks of an 8-rank chain (exchange by copy)"
python bench.py --fcc --size 1536 --precision double --steps 12 --warmup 4 --repeats 3 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc 2>/dev/null | python -c 'import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); print("   N=1: %.4f ms/step = %.1f Gvox/s (%s %.2f ms/launch)" % (d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms_per_launch"]))'
for spec in 0/8 3/8 7/8; do
  python bench.py --fcc --size 1536 --precision double --emulate-slab $spec --emulate-transport copy --steps 12 --warmup 4 --repeats 3 --no-pmc 2>/dev/null | python -c "$P"
done
echo "## one process per GPU (pffdtd_amd/dist.py under torch.distributed.run): a rank of 8 alone, exchanging with itself -- device copies / torch.distributed p2p (batch_isend_irecv) / the library's own ncclSend + ncclRecv on the edge stream (pf_rccl_*, what dist.py uses under an RCCL process group)"
for tr in copy rccl native; do
  python bench.py --emulate-slab 3/8 --emulate-via torch --emulate-transport $tr --steps 42 --warmup 6 --repeats 5 --no-pmc 2>&1 >/dev/null | grep emulated | sed 's/^/   /'
done
