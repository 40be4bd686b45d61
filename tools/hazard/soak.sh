#!/bin/bash
# On the GPU box, appended to other GPU sessions of the round: the builds of eval.hip that carry the SLP vectoriser's packed
# forms (the round-3 source that failed and today's) through the repeat-launch harness, with the box's GPU id beside the
# counts -- how often the intermittent wrong tile shows, and on which devices.
cd "$(dirname "$0")/build" || exit 0
mkdir -p ../../../gpurun_out/hazard
{
  echo "== $(date -u +%FT%TZ) gpu unique_id: $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | tr '\n' ' ') visible ${ROCR_VISIBLE_DEVICES:-?}/${HIP_VISIBLE_DEVICES:-?} host $(hostname)"
  for lib in lib_old_slp.so lib_slp.so; do
    [ -f $lib ] || continue
    timeout 200 ./scorer_repro ./$lib ${REPS:-1500} 1 2 64 1 2>&1 | tail -8
    timeout 200 ./scorer_repro ./$lib ${REPS:-1500} 1 2 64 0 2>&1 | tail -8
  done
} 2>&1 | tee -a ../../../gpurun_out/hazard/soak_$(date +%s).log | tail -6
