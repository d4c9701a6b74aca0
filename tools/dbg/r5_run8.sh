cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
for e in 0 1 0 1; do
  if [ $e = 1 ]; then export MA_ALIGN_PK=1; else unset MA_ALIGN_PK; fi
  python3 bench.py --steps 4 --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pk=$e', 'lanes', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('k_align_reg'), d['kernel_ms_per_step'].get('k_align_tb'))" >> gpurun_out/r5b/ab_pk.txt
done
