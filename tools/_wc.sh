R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=$R/gpurun_out/wallpmc2; rm -rf $OUT; mkdir -p $OUT
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pass$i -o p -- python $R/bench.py --variant 40 --debug 0x4008000 \
       --no-rigid-run --no-selfcheck --no-cpu-baseline --steps 8 --warmup 4 --repeats 1 > $OUT/pass$i.log 2>&1
done
for db in $(find $OUT -name '*.db' | sort); do python $R/tools/rocpd_pmc.py k_wall2 $db; done > $OUT/summary.txt 2>&1
find $OUT -name '*.db' -delete
grep "==\|INSTS_VALU\|SQ_WAVES\|WAVE_CYCLES\|duration\|SALU\|WAIT_ANY\|BRANCH\|SMEM\|LDS\|WAIT_INST\|ACTIVE_INST_ANY" $OUT/summary.txt
