#!/bin/bash
# issue / texture-addresser / L1 counters of the wall-region kernels on the headline workload (one pass per group)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=$R/gpurun_out/wallpmc; rm -rf $OUT; mkdir -p $OUT
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_FLAT" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pass$i -o p -- python $R/bench.py --debug 0x8000 --no-pmc \
       --no-rigid-run --no-selfcheck --no-cpu-baseline --steps 8 --warmup 4 --repeats 1 > $OUT/pass$i.log 2>&1
  echo "pass$i rc=$?"
done
for db in $(find $OUT -name '*.db' | sort); do python $R/tools/rocpd_pmc.py k_wall2 $db; done > $OUT/summary.txt 2>&1
find $OUT -name '*.db' -delete
