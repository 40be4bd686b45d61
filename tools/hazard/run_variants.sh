#!/bin/bash
# On the GPU box: every library build under tools/hazard/build against the repeat-launch harness (hm / TIE / recdim 64).
cd "$(dirname "$0")/build"
mkdir -p ../../../gpurun_out/hazard
for lib in lib_*.so; do
  for round in 1 2 3; do
    timeout 300 ./scorer_repro ./$lib ${REPS:-1000} 1 2 64 ${ASYNC:-1} 2>&1 | tail -${TAIL:-14}
  done
done 2>&1 | tee ../../../gpurun_out/hazard/variants_$(date +%s).log
