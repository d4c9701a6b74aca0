# Developer tool: the parity sweeps behind profiles/r5_sweep_* (C3/C2/C5 shapes, the C4 windows, POA shapes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 1500 python3 tools/sweep_parity.py 0 > gpurun_out/r5g/r5_sweep_parity.txt 2>&1
timeout 1500 python3 tools/sweep_parity.py 1000 > gpurun_out/r5g/r5_sweep_parity_shift1000.txt 2>&1
timeout 900 python3 tools/sweep_parity.py 0 c4 > gpurun_out/r5g/r5_sweep_parity_c4.txt 2>&1
timeout 600 python3 tools/sweep_poa.py > gpurun_out/r5g/r5_sweep_poa.txt 2>&1
