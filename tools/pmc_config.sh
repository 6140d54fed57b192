#!/bin/bash
# usage: tools/pmc_config.sh <tag> <run_config args...>  -> HBM bytes and issue counters per kernel for a real configuration
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
   "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
   i=$((i+1))
   timeout 600 rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pmc$i -o p --output-format csv -- python $R/tools/run_config.py "$@" --steps 6 --warmup 2 > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT
