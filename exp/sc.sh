#!/bin/bash
for C in 1 0 1 0; do
for W in dragon grid; do MIRT_SHADOW_CULL=$C timeout -k 10 300 python3 bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cull $C $W', round(d['value']), d['ms_per_step'], d['rays_walked_per_frame'])" || exit 1; done; done
