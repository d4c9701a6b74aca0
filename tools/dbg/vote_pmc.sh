# Developer tool: SQ counters of the genotype kernels (what bounds k_vote: VALU issue, the LDS pipe, or waiting?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b
mkdir -p $O
export MA_BENCH_CACHE=/tmp/mbc
python3 bench.py --no-cpu --no-also --gen-only > /dev/null 2>&1
export MA_STREAMS=1
B="python3 bench.py --steps 1 --warmup 1 --no-cpu --no-also"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $O/pmc_a -- $B > $O/pmc_a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $O/pmc_b -- $B > $O/pmc_b.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
  --kernel-trace --output-format csv -d $O/pmc_c -- $B > $O/pmc_c.log 2>&1
for p in a b c; do python3 tools/dbg/pmc_generic.py $O/pmc_$p "${PMC_KERNELS:-k_vote|k_align_reg|k_support|k_classify|k_msa}" > $O/vote_pmc_$p.txt 2>&1; tail -3 $O/pmc_$p.log >> $O/vote_pmc_$p.txt; rm -rf $O/pmc_$p; done
