#!/bin/bash
# A/B of two builds of the library on one GPU: the same bench.py command alternately with MMM_LIB_PATH = lib A and lib B, N rounds.
# usage: tools/ab_lib.sh <libA.so> <libB.so> <rounds> -- <bench.py arguments>
A=$1; B=$2; N=$3; shift 4
for i in $(seq 1 $N); do
  for L in $A $B; do
    MMM_LIB_PATH=$PWD/$L python3 bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', 'ms_per_step %.5f' % r['ms_per_step'], 'events %.5f' % (r.get('ms_per_step_events') or 0), {k: round(v, 2) for k, v in r['iteration']['kernel_us'].items()})"
  done
done
