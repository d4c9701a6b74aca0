cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python3 tools/prof_phases.py 4096 bench 2>&1 | tail -2 > gpurun_out/r5b/vprof.txt
