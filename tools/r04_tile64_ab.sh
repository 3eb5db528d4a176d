#!/bin/bash
# A/B of the 64 x 64-tile forward / dX kernels (MLGGD_TILE64=1 auto / 0 off) at the shapes that select them.
# usage: tools/r04_tile64_ab.sh > gpurun_out/r04_tile64_ab.txt
for cfg in "--hidden 4096 --nhid 6 --bunch 512 --loss ml --steps 40 --warmup 10" \
           "--hidden 2048 --nhid 3 --bunch 512 --loss mmse --steps 200 --warmup 20" \
           "--hidden 4096 --nhid 6 --bunch 256 --loss ml --steps 60 --warmup 10" \
           "--hidden 2048 --nhid 3 --bunch 256 --loss mmse --steps 200 --warmup 20"; do
  for t in 1 0; do
    echo "# MLGGD_TILE64=$t bench.py $cfg"
    MLGGD_TILE64=$t python bench.py $cfg --no-cpu-baseline --no-dp-rehearsal --no-ml --windows 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'step_roofline_frac':d['step_roofline_frac'],'by_class':d.get('dp_breakdown',{}).get('compute_us_by_class')}))"
  done
done
