#!/bin/bash
# usage (on the GPU box): tools/collect_n1_profile.sh <tag>   -> profiles/<tag>_{kernel_stats.csv,hbm_traffic.json,summary.md,.json}
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only; --debug 0x8000 = no creation-time measurement, which
# under counter collection can reject the pair path the timed run uses) first, so that the timed run's bench line
# carries the traffic measured on this very build; then the default bench command under --kernel-trace --stats.
set -u
TAG=${1:-r05_bench_n1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o p --output-format csv -- python $R/bench.py --no-pmc --steps 9 --warmup 3 --repeats 1 --no-rigid-run --no-cpu-baseline --no-selfcheck --debug 0x8000 > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o p --output-format csv -- python $R/bench.py --no-pmc --steps 9 --warmup 3 --repeats 1 --no-rigid-run --no-cpu-baseline --no-selfcheck --debug 0x8000 > $O/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats0 -o s --output-format csv -- python $R/bench.py --no-pmc --steps 9 --warmup 3 --repeats 1 --no-rigid-run --no-cpu-baseline --no-selfcheck > $O/stats0.log 2>&1
(cd $R && python tools/make_profile_summary.py $TAG $O/stats0 $O/fetch $O/write > /dev/null)
sleep ${COOL_S:-45}  # the box loses 2-3 % once warm: let it idle before the timed run
# the committed kernel statistics: the time loop only (no self-check run, no second scene; creation-time probes run under
# their own kernel name), ...
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python $R/bench.py --steps 48 --warmup 6 --no-selfcheck --no-pmc --no-rigid-run --no-cpu-baseline > $O/stats.log 2>&1
sleep ${COOL_S:-45}
# ... and the bench line itself from a plain run of the default command
(cd $R && timeout 900 python bench.py --steps 48 --warmup 6 > $O/plain.log 2> $O/plain.err)
grep '^{"metric"' $O/plain.log | tail -1 > $O/bench.json
(cd $R && python tools/make_profile_summary.py $TAG $O/stats $O/fetch $O/write $O/bench.json && cp $O/bench.json profiles/$TAG.json && mkdir -p gpurun_out/profiles_new && cp profiles/$TAG* gpurun_out/profiles_new/)
cat $O/bench.json
