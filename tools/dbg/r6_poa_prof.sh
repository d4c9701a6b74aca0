#!/bin/bash
# Developer tool (GPU box): where a banded POA fill spends its time (-DMA_PROFILE build of poa.hip: tools/build_prof.sh poa.hip)
set -u
O=gpurun_out/r6_poa_prof
mkdir -p $O
for n in 64 1024 8192; do
  echo "== $n windows" >> $O/prof.txt
  timeout 600 python3 tools/prof_phases.py $n bench 2>&1 | grep -v amdgpu.ids | grep "ma_debug_prof\|k_msa" >> $O/prof.txt
done
cat $O/prof.txt
