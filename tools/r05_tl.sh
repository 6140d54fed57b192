#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/n1 -o t --output-format csv -- python $R/bench.py --steps 12 --warmup 6 --repeats 1 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc > $O/n1.log 2>&1
cd $R
for t in n1; do
  n=$(python - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/$t/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
print(len(rows))
PY
)
  echo "== $t: $n kernels" > $O/$t.txt
  python tools/timeline.py $O/$t $((n-40)) 36 >> $O/$t.txt
  rm -rf $O/$t
done
