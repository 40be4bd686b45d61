#!/bin/bash
# Builds of the library that differ in eval.hip only (compiler flags / probe macros), for tools/hazard/scorer_repro.cpp.
#   tools/hazard/build_variants.sh name:"flags" ...     ->  tools/hazard/build/lib_<name>.so
set -e
cd "$(dirname "$0")/../../elimrec_amd/csrc"
OUT=../../tools/hazard/build
mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
OTHERS=$(ls ../lib/obj/*.o | grep -v '/eval.o')
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc $BASE $flags -c eval.hip -o $OUT/eval_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $OUT/eval_$name.o $OTHERS -ldl &&
    rm $OUT/eval_$name.o && echo built $name ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc -O2 -w ../../tools/hazard/scorer_repro.cpp -o $OUT/scorer_repro -ldl
