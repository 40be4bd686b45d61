#!/bin/bash
# On the GPU box: the scorer stability test of the suite against another build of the library (copied over the in-tree one
# in the box's scratch copy only).
set -e
cd "$(dirname "$0")/../.."
cp tools/hazard/build/lib_$1.so elimrec_amd/lib/libelimrec_hip.so
python -m pytest tests/test_hip_parity.py -q -k "test_bf16x3_scorer_is_stable" 2>&1 | tail -${TAIL:-25}
