#!/bin/bash
# Developer tool: libmicroasm_prof.so = the library with build.hip compiled -DMA_PROFILE (phase clocks: tools/dbg/prof_insert.py)
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build_prof
cd $R/lancet2_amd/csrc
make -j8 -s
for f in *.o; do cp $f $R/build_prof/$f; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DMA_PROFILE -c build.hip -o $R/build_prof/build.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/lancet2_amd/libmicroasm_prof.so $R/build_prof/*.o
