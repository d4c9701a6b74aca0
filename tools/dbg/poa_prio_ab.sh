# Developer tool: A/B of the POA rounds on a high-priority stream (MA_POA_PRIORITY), four lanes, cached windows
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -f gpurun_out/r5b/poa_prio.txt
for rep in 1 2; do for v in 1 0; do
  MA_POA_PRIORITY=$v python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('MA_POA_PRIORITY=$v', d['value'], d['ms_per_step'], 'k_msa', d['kernel_ms_per_step'].get('k_msa'), 'k_msa_band', d['kernel_ms_per_step'].get('k_msa_band'))" >> gpurun_out/r5b/poa_prio.txt
done; done
