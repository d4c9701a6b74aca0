# Developer tool: single-lane and four-lane bench of variant libraries (LIBS="a.so b.so", KERNELS="k_x k_y" to print)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
rm -f gpurun_out/r5b/lib_ab.txt
for rep in 1 2; do for lib in ${LIBS:-libmicroasm.so}; do for L in 1 4; do
  MA_LIB=$PWD/lancet2_amd/$lib MA_STREAMS=$L python3 bench.py --steps 6 --no-cpu --no-also 2>/dev/null | tail -1 | python3 -c "
import sys,json,os
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$lib lanes $L', d['value'], d['ms_per_step'], ' '.join('%s %.2f' % (x, k.get(x, 0)) for x in os.environ.get('KERNELS','k_align_reg k_align_tb').split()))" >> gpurun_out/r5b/lib_ab.txt
done; done; done
