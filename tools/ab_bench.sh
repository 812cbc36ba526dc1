#!/bin/bash
# same-box A/B of library builds: tools/ab_bench.sh libA.so libB.so ... (each run twice, alternating)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "HEAD" ]; then unset LDMSEG_HIP_LIB; else export LDMSEG_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
    python bench.py --no-cpu-baseline --no-extras --no-images --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['roofline']['families_ms_per_step']
print('$lib', 'ms/step %.3f' % d['ms_per_step'], 'value %.1f' % d['value'], {k: round(v,3) for k,v in f.items()})"
  done
done
