cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
MA_VERBOSE=1 timeout 600 python tools/dbg/cascade_passes.py 2048 > gpurun_out/r5b/cascade.txt 2>&1
