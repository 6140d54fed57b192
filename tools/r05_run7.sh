cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
timeout 3000 python -m pytest tests -m gpu -q  > gpurun_out/r05g/t.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05g/t.log
tail -40 gpurun_out/r05g/t.log
