cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for e in 256 64 128 512 1024 256; do
  MA_POA_MIN_PENDING=$e python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('min_pending=$e', d['value'], d['ms_per_step'], 'k_msa', k.get('k_msa'), 'band', k.get('k_msa_band'))" >> gpurun_out/r5b/ab_poa.txt
done
