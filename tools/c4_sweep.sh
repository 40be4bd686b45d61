#!/bin/bash
# On the GPU box: the C4-shape hop (I = 1.2 M, recdim 128), tile hop against tile hop + window sweep, a few window sizes.
cd "$(dirname "$0")/.."
export SHAPE=c4
python tools/hop_only.py 128 6 2>&1 | tail -2
for win in ${WINDOWS:-12288}; do
  echo "== sweep, window $win"
  SWEEP=1 ELIMREC_SWEEP_WINDOW=$win python tools/hop_only.py 128 6 2>&1 | tail -5
done
