cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -x -q -k "set_fills_is_split" 2>&1 | tail -8 > gpurun_out/r5b/t.txt
