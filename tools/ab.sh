#!/bin/bash
# On the GPU box: the headline step under two settings of one environment switch, interleaved, ms per step of each run.
#   tools/ab.sh NAME "v1 v2 ..." [rounds] [extra bench flags]
name=$1; vals=$2; rounds=${3:-2}; shift 3
for r in $(seq $rounds); do
  for v in $vals; do
    env $name=$v python bench.py --steps ${STEPS:-300} --warmup 20 --no-eval --no-cpu-baseline --no-reference-work --no-b-sweep --no-projection "$@" 2>/dev/null | tail -1 |
      python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$name=$v ms_per_step %.4f hop_us %.2f issue: %s' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['config']['step_issue'][:40]))"
  done
done
