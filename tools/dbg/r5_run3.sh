cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python tools/prof_phases.py 2048 bench > gpurun_out/r5b/prof_phases.txt 2>&1
