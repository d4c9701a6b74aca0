#!/bin/bash
# Developer tool: libmicroasm_prof.so = the library compiled -DMA_PROFILE (phase clocks: tools/dbg/prof_insert.py, tools/prof_phases.py)
# usage: tools/build_prof.sh [file.hip ...]   (default: build.hip only; "all" = every file)
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build_prof
cd $R/lancet2_amd/csrc
make -j8 -s
FILES="${@:-build.hip}"
[ "$FILES" = "all" ] && FILES=$(ls *.hip)
for f in *.o; do [ -f $R/build_prof/$f ] && [ $R/build_prof/$f -nt $f ] && [[ " $FILES " != *" ${f%.o}.hip "* ]] || cp $f $R/build_prof/$f; done
for h in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DMA_PROFILE -c $h -o $R/build_prof/${h%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/lancet2_amd/libmicroasm_prof.so $R/build_prof/*.o
