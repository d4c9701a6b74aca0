#!/bin/bash
# Developer tool: SQ counters of the POA kernels (POA stage alone).  usage: tools/r3_poa_pmc.sh [tier0] [windows]
set -u
R=$PWD
O=$R/gpurun_out/poa_pmc
rm -rf $O && mkdir -p $O
T=${1:-2}
N=${2:-8192}
cd /tmp && export TMPDIR=/tmp
export MA_POA_TIER0=$T
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/poa_bench.py $N 256 > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU \
  --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/poa_bench.py $N 256 > $O/b.log 2>&1
cd $R
python3 tools/dbg/pmc_generic.py $O/a "k_msa"
python3 tools/dbg/pmc_generic.py $O/b "k_msa"
tail -2 $O/a.log
rm -rf $O/a $O/b
