set -x
bash tools/collect_n1_profile.sh r02_bench_n1 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_ctk -o s --output-format csv -- python $R/tools/run_config.py ctk_cart_gpu --steps 400 2>&1 | grep '"config"' > $R/gpurun_out/ctk.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_mv -o s --output-format csv -- python $R/tools/run_config.py mv_fcc_gpu --steps 200 2>&1 | grep '"config"' > $R/gpurun_out/mv.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_fcc -o s --output-format csv -- python $R/bench.py --fcc --steps 30 --warmup 6 --repeats 3 --no-cpu-baseline --no-rigid-run 2>&1 | grep '"metric"' > $R/gpurun_out/fcc.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_fcc40 -o s --output-format csv -- python $R/bench.py --fcc --steps 30 --warmup 6 --repeats 3 --no-cpu-baseline --no-rigid-run --variant 40 2>&1 | grep '"metric"' > $R/gpurun_out/fcc40.json
python $R/tools/run_config.py ctk_cart_viz --precision double --energy 2>&1 | grep '"config"' > $R/gpurun_out/ctk_viz.json
cat $R/gpurun_out/ctk.json $R/gpurun_out/mv.json $R/gpurun_out/ctk_viz.json | cut -c1-400
