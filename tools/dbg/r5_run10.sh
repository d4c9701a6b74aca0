cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r5g/r5_final_gpu_tests.txt
timeout 1500 python tools/sweep_parity.py 2 > gpurun_out/r5g/r5_sweep_parity.txt 2>&1
timeout 1200 python tools/sweep_parity.py 2 c4 > gpurun_out/r5g/r5_sweep_parity_c4.txt 2>&1
MA_ALIGN_PK=0 timeout 1500 python tools/sweep_parity.py 3 > gpurun_out/r5g/r5_sweep_parity_one_pair_aligner.txt 2>&1
timeout 900 python tools/sweep_poa.py > gpurun_out/r5g/r5_sweep_poa.txt 2>&1
