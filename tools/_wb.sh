R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_hip_tb2.py -x -q -m gpu 2>&1 | tail -1
python $R/bench.py --no-cpu-baseline --no-rigid-run 2>&1 | tail -1 | cut -c1-330
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mc -o p -- python $R/bench.py --debug 0x4000000 --no-cpu-baseline --no-rigid-run --no-selfcheck --steps 40 --warmup 8 --repeats 1 > /dev/null 2>&1
grep "k_wall2\|k_tb2_reg<float, 3, 4, false, 64, false\|k_boundary" $R/gpurun_out/mc/*kernel_stats.csv | awk -F'",' '{print substr($1,1,66), $2}' | cut -c1-150; rm -rf $R/gpurun_out/mc
