#!/bin/bash
# time and fetched bytes of the pair-kernel variants of tools/tb2_bytes_sweep.py; usage: tb2_bytes_sweep.sh [tyes] [chunks]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TYES=${1:-11304,11308,11208,1304}; CH=${2:-8,16,32,64}
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/tb2_sweep_fetch
python $R/tools/tb2_bytes_sweep.py 1024 $TYES $CH 5 2>&1 | tee $R/gpurun_out/tb2_sweep_time.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tb2_sweep_fetch -o p -- python $R/tools/tb2_bytes_sweep.py 1024 $TYES $CH 2 > /dev/null 2>&1
cd $R; python - <<PY | tee $R/gpurun_out/tb2_sweep_bytes.txt
import csv,glob,collections,statistics
f=glob.glob("gpurun_out/tb2_sweep_fetch/**/*counter_collection.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_tb2_reg" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
        d[(r["Kernel_Name"].split("(")[0][-34:],r["Grid_Size"])].append(float(r["Counter_Value"]))
for k,v in d.items(): print(k, len(v), "read GB %.3f" % (statistics.median(v)*2048/1e9))
PY
rm -rf $R/gpurun_out/tb2_sweep_fetch
