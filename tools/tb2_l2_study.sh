#!/bin/bash
# L2 / fabric counters of the pair kernel k_tb2_reg on the headline workload, for the grids as allocated (debug 0x8000:
# no placement search) and as placed.  One rocprofv3 pass per counter group (TCC has four slots); rocpd databases land in
# gpurun_out/l2_<tag>/pass<i>/ and tools/rocpd_pmc.py prints the medians.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
for TAG in placed asalloc; do
  EXTRA=""; [ $TAG = asalloc ] && EXTRA="--debug 0x8000"
  OUT=$R/gpurun_out/l2_$TAG; mkdir -p $OUT
  i=0
  for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
             "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum" \
             "TCC_EA0_RDREQ TCC_EA0_WRREQ" \
             "TCC_BUBBLE_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pass$i -o p -- python $R/bench.py --variant 40 $EXTRA \
       --no-rigid-run --no-selfcheck --no-cpu-baseline --steps 8 --warmup 4 --repeats 1 > $OUT/pass$i.log 2>&1
    echo "$TAG pass$i rc=$? : $PMC"
    tail -1 $OUT/pass$i.log | cut -c1-300
  done
  for db in $(find $OUT -name '*.db' | sort); do python $R/tools/rocpd_pmc.py k_tb2_reg $db; done > $OUT/summary.txt 2>&1
  find $OUT -name '*.db' ! -path '*pass4*' -delete
done
