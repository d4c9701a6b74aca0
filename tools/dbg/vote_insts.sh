# Developer tool: instruction counts of k_vote per timing-build variant (counts are additive where times are not)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
export MA_STREAMS=1
rm -f $O/vote_insts.txt
for lib in ${VOTE_LIBS:-libmicroasm.so}; do
  export MA_LIB=$PWD/lancet2_amd/$lib
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --kernel-trace --output-format csv -d $O/pmc_v -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-also > $O/pmc_v.log 2>&1
  echo "== $lib" >> $O/vote_insts.txt
  python3 tools/dbg/pmc_generic.py $O/pmc_v "k_vote" >> $O/vote_insts.txt 2>&1
  rm -rf $O/pmc_v
done
