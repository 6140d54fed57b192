#!/bin/bash
# One GPU-box call: the round-6 measurement set.  Outputs under gpurun_out/r06/, summaries copied to gpurun_out/profiles_new/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O $R/gpurun_out/profiles_new
cd $R
export PYTHONUNBUFFERED=1
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = n1 ]; then
  COOL_S=15 bash tools/collect_n1_profile.sh r06_bench_n1 > $O/n1.log 2>&1
  # the driver's own command (K = 20), PMC passes inside
  PFFDTD_VERBOSE=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
  cp $O/bench_k20.json $R/gpurun_out/profiles_new/r06_bench_n1_driver_command.json
  # kernel timeline of one steady-state triple
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t --output-format csv -- python $R/bench.py --steps 24 --warmup 6 --repeats 2 --no-rigid-run --no-cpu-baseline --no-selfcheck --no-pmc > /dev/null 2>&1)
  { echo "# a triple of a headline region (no per-launch events):"; python tools/timeline.py $O/tr "false, 3>" 12; echo "# a triple of the last region (per-launch HIP events on: the two readouts are separate launches there):"; python tools/timeline.py $O/tr "false, 3>" 3; } > $R/gpurun_out/profiles_new/r06_triple_timeline.txt 2>&1; rm -rf $O/tr
fi
if [ "$WHAT" = all ] || [ "$WHAT" = wall ]; then
  # issue counters of the shell's kernels (waves, cycles executing / waiting, instructions), each launch alone under --pmc
  OUT=$O/wallpmc; rm -rf $OUT; mkdir -p $OUT
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc $PMC --kernel-trace -d $OUT/pass$i -o p -- python $R/bench.py --debug 0x8000 --no-pmc --no-rigid-run --no-selfcheck --no-cpu-baseline --steps 9 --warmup 3 --repeats 1 > $OUT/pass$i.log 2>&1)
  done
  { echo "# Round 6: the shell's kernels under rocprofv3 --pmc (one counter group per pass, each launch alone), bench.py N=1 1024^3 fp32 Mb=11";
    echo "# k_wall2<10,false,..,3> = x / y regions, three steps per pass; k_wall2<20,true,..,3> = column strips, three steps; k_brick = the frame";
    for db in $(find $OUT -name '*.db' | sort); do python tools/rocpd_pmc.py k_wall2 $db; python tools/rocpd_pmc.py k_brick $db; done; } > $R/gpurun_out/profiles_new/r06_wall_counters.txt 2>&1
  find $OUT -name '*.db' -delete
fi
if [ "$WHAT" = all ] || [ "$WHAT" = variants ]; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc > $O/v_exact.json 2> $O/v_exact.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc --numerics 2 > $O/v_sg.json 2> $O/v_sg.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-pmc --precision double > $O/v_f64.json 2> $O/v_f64.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc --debug 0x400000 > $O/v_r5shell.json 2> $O/v_r5shell.err
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rigid-run --no-selfcheck --no-pmc --debug 0x1000000 > $O/v_2plus1.json 2> $O/v_2plus1.err
  cat $O/v_exact.json $O/v_sg.json $O/v_f64.json $O/v_r5shell.json $O/v_2plus1.json > $R/gpurun_out/profiles_new/r06_variants.jsonl
fi
if [ "$WHAT" = all ] || [ "$WHAT" = fcc ]; then
  python bench.py --fcc --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fcc.json 2> $O/bench_fcc.err
  python bench.py --fcc --precision double --size 1536 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-rigid-run > $O/bench_fcc64_1536.json 2> $O/bench_fcc64_1536.err
  cat $O/bench_fcc.json $O/bench_fcc64_1536.json > $R/gpurun_out/profiles_new/r06_fcc_n1.jsonl
fi
if [ "$WHAT" = all ] || [ "$WHAT" = configs ]; then
  : > $O/reference_configs.jsonl
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/ctk -o s --output-format csv -- python $R/tools/run_config.py ctk_cart_gpu --steps 400 > $O/ctk.log 2>&1)
  grep '^{"config"' $O/ctk.log >> $O/reference_configs.jsonl
  (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/mv -o s --output-format csv -- python $R/tools/run_config.py mv_fcc_gpu --steps 200 > $O/mv.log 2>&1)
  grep '^{"config"' $O/mv.log >> $O/reference_configs.jsonl
  timeout 600 python tools/run_config.py ctk_cart_viz --precision double --energy 2> $O/viz.err | grep '^{"config"' >> $O/reference_configs.jsonl
  cp $O/reference_configs.jsonl $R/gpurun_out/profiles_new/r06_reference_configs.jsonl
fi
if [ "$WHAT" = all ] || [ "$WHAT" = chain ]; then
  bash tools/slab_cost_model.sh > $R/gpurun_out/profiles_new/r06_slab_cost_model.txt 2> $O/cost_model.err
fi
ls -la $R/gpurun_out/profiles_new/
