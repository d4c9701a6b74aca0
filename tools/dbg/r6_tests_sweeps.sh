# Developer tool: the whole GPU suite + smoke + the parity sweeps (what profiles/r6_final_gpu_tests.txt and r6_sweep_* hold)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r6g/r6_final_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r6g/r6_final_gpu_tests.txt 2>&1
timeout 1500 python3 tools/sweep_parity.py 0 > gpurun_out/r6g/r6_sweep_parity.txt 2>&1
timeout 1500 python3 tools/sweep_parity.py 1000 > gpurun_out/r6g/r6_sweep_parity_shift1000.txt 2>&1
timeout 900 python3 tools/sweep_parity.py 0 c4 > gpurun_out/r6g/r6_sweep_parity_c4.txt 2>&1
timeout 600 python3 tools/sweep_poa.py > gpurun_out/r6g/r6_sweep_poa.txt 2>&1
timeout 600 python3 tools/sweep_poa.py 500 > gpurun_out/r6g/r6_sweep_poa_shift500.txt 2>&1
MA_POA_SCHED=0 timeout 600 python3 tools/sweep_poa.py > gpurun_out/r6g/r6_sweep_poa_host_rounds.txt 2>&1
for s in 1 2 3 4; do MA_SWEEP_SEED=$s timeout 600 python3 -m pytest tests/test_gpu_aligner.py -x -q -m gpu 2>&1 | tail -2; done > gpurun_out/r6g/r6_aligner_seeds.txt 2>&1
for f in gpurun_out/r6g/*.txt; do tail -n 2 $f; done
