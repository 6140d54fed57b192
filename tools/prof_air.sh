#!/bin/bash
# usage: tools/prof_air.sh <tag> <tune_air args...>   -> gpurun_out/prof_<tag>/{stats,pmc*}/...
# Collects rocprofv3 kernel stats and PMC passes (each in its own run, --kernel-trace only) for tools/tune_air.py.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python $R/tools/tune_air.py "$@" > $OUT/stats.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
   "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
   "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
   "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
   i=$((i+1))
   rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pmc$i -o p --output-format csv -- python $R/tools/tune_air.py "$@" --steps 4 > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT
