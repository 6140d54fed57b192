cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
for c in mv_fcc_gpu ctk_cart_gpu; do
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$c.fetch -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_config.py $c --steps 30 --warmup 5 --debug 0x8000 > $O/$c.fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$c.write -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_config.py $c --steps 30 --warmup 5 --debug 0x8000 > $O/$c.write.log 2>&1
python - $O/$c <<'PY'
import csv, glob, sys, collections
def med(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "pf::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return {k: (sorted(v)[len(v)//2], len(v)) for k, v in agg.items()}
f, w = med(sys.argv[1] + ".fetch", "FETCH_SIZE"), med(sys.argv[1] + ".write", "WRITE_SIZE")
for k in sorted(set(f) | set(w)):
    print(f"{k:60s} read {f.get(k,(0,0))[0]*2048/1e6:9.1f} MB  write {w.get(k,(0,0))[0]*1024/1e6:9.1f} MB  n={f.get(k,(0,0))[1]}")
PY
done
