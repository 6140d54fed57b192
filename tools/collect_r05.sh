#!/bin/bash
# One GPU-box call: the round-5 measurement set.  Outputs under gpurun_out/r05/, summaries copied to gpurun_out/profiles_new/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p $O $R/gpurun_out/profiles_new
cd $R
export PYTHONUNBUFFERED=1
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = n1 ]; then
  COOL_S=20 bash tools/collect_n1_profile.sh r05_bench_n1 > $O/n1.log 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = chain ]; then
  PFFDTD_VERBOSE=1 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2_virtual_peer.json 2> $O/bench_n2_virtual_peer.err
  PFFDTD_VERBOSE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --transport rccl > $O/bench_n2_virtual_rccl.json 2> $O/bench_n2_virtual_rccl.err
  PFFDTD_VERBOSE=1 python bench.py --gpus 4 --steps 20 --warmup 5 --transport rccl > $O/bench_n4_virtual_rccl.json 2> $O/bench_n4_virtual_rccl.err
  cat $O/bench_n2_virtual_peer.json $O/bench_n2_virtual_rccl.json $O/bench_n4_virtual_rccl.json > $R/gpurun_out/profiles_new/r05_chain_virtual_slabs.jsonl
fi
if [ "$WHAT" = all ] || [ "$WHAT" = configs ]; then
  : > $O/reference_configs.jsonl
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/ctk -o s --output-format csv -- python $R/tools/run_config.py ctk_cart_gpu --steps 400 > $O/ctk.log 2>&1)
  grep '^{"config"' $O/ctk.log >> $O/reference_configs.jsonl
  (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/mv -o s --output-format csv -- python $R/tools/run_config.py mv_fcc_gpu --steps 200 > $O/mv.log 2>&1)
  grep '^{"config"' $O/mv.log >> $O/reference_configs.jsonl
  timeout 600 python tools/run_config.py ctk_cart_viz --precision double --energy 2> $O/viz.err | grep '^{"config"' >> $O/reference_configs.jsonl
  cp $O/reference_configs.jsonl $R/gpurun_out/profiles_new/r05_reference_configs.jsonl
  for c in ctk mv; do python - "$O/$c" "$R/gpurun_out/profiles_new/r05_${c}_kernel_stats.csv" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*kernel_stats.csv"):
    rows += [r for r in csv.DictReader(open(f)) if "pf::" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
if rows:
    w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = fcc ]; then
  python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fcc.json 2> $O/bench_fcc.err
  python bench.py --fcc --precision double --size 1536 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-rigid-run > $O/bench_fcc64_1536.json 2> $O/bench_fcc64_1536.err
  python bench.py --fcc --numerics 2 --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run > $O/bench_fcc_sg.json 2> $O/bench_fcc_sg.err
  cat $O/bench_fcc.json $O/bench_fcc_sg.json $O/bench_fcc64_1536.json > $R/gpurun_out/profiles_new/r05_fcc_n1.jsonl
  # per-kernel times and PMC bytes of the 13-point pair path
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/fcc_stats -o s --output-format csv -- python $R/bench.py --fcc --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc > $O/fcc_stats.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fcc_fetch -o p --output-format csv -- python $R/bench.py --fcc --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc --debug 0x8000 > $O/fcc_fetch.log 2>&1)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/fcc_write -o p --output-format csv -- python $R/bench.py --fcc --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc --debug 0x8000 > $O/fcc_write.log 2>&1)
  python tools/make_profile_summary.py r05_fcc_n1 $O/fcc_stats $O/fcc_fetch $O/fcc_write $O/bench_fcc.json > /dev/null && cp profiles/r05_fcc_n1_* gpurun_out/profiles_new/
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sg ]; then
  # safeguarded numerics (round 4: also in the 7-point pair kernels) against the CPU-exact arithmetic, both on the default paths
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc > $O/bench_exact_single.json 2> $O/bench_exact_single.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --numerics 2 --no-pmc > $O/bench_safeguarded.json 2> $O/bench_safeguarded.err
  python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --debug 0x4000 --no-pmc > $O/bench_fcc_exact_single.json 2> $O/bench_fcc_exact_single.err
  python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --numerics 2 --no-pmc > $O/bench_fcc_safeguarded.json 2> $O/bench_fcc_safeguarded.err
  cat $O/bench_exact_single.json $O/bench_safeguarded.json $O/bench_fcc_exact_single.json $O/bench_fcc_safeguarded.json > $R/gpurun_out/profiles_new/r05_safeguarded_vs_exact.jsonl
fi
ls -la $R/gpurun_out/profiles_new/
