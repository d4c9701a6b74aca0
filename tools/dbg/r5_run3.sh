cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 600 python tools/dbg/prof_insert.py > gpurun_out/r5b/prof.txt 2>&1
