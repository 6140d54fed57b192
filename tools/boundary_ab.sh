#!/bin/bash
# boundary pass of a room configuration: workgroup order (debug 0x100000 = plain) x neighbour fetch (0x200000 = all) A/B
# usage: tools/boundary_ab.sh <config> [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=${1:-mv_fcc_gpu}; STEPS=${2:-60}
cd /tmp; export TMPDIR=/tmp
for DBG in 0x0 0x100000 0x200000 0x300000; do
  OUT=$R/gpurun_out/bnd_${CFG}_$DBG; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/tools/run_config.py $CFG --steps $STEPS --warmup 10 --debug $DBG > $OUT/run.log 2>&1
  echo "== debug $DBG: $(tail -1 $OUT/run.log | cut -c1-400)"
  grep -h "k_boundary\|k_air" $(find $OUT -name '*kernel_stats.csv') | awk -F'",' '{print substr($1,1,60), $2}' | awk -F, '{print $1, "calls", $(NF-6), "avg_ns", $(NF-4)}' | head -4
  find $OUT -name '*.csv' ! -name '*kernel_stats.csv' -delete
done
