cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r5g/r5_final_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r5g/r5_final_gpu_tests.txt 2>&1
