#!/bin/bash
# On the GPU box: bench.py's batch_sweep line (the step at B = 2048 .. 32768 on one GPU) under the environment given, one line per size.
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 40 --warmup 10 --no-eval --no-cpu-baseline --no-reference-work --no-reduced-precision 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline %.4f ms' % d['ms_per_step'], ' '.join('B=%d: %.3f ms' % (s['batch'], s['ms_per_step']) for s in d['batch_sweep']['sizes']))"
