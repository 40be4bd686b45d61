#!/bin/bash
# On the GPU box: the multi-rank path over a one-rank RCCL group under a few switches, ms per step each.
run() { env "$@" ELIMREC_SHARD_MULTI=1 python bench.py --steps 200 --warmup 20 --no-eval --no-cpu-baseline --no-reference-work --no-b-sweep --no-projection $FLAGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.4f ms  %s' % ('$*' + ' ' + '$FLAGS', d['ms_per_step'], d['config']['step_issue'][:30]))"; }
for spec in "$@"; do FLAGS="--feature-shard row" run $spec; done
