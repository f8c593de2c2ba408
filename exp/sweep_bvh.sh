#!/bin/bash
for LM in 1 2; do for CP in 1.5 2.5 4 6; do
  echo "== LEAF_MIN $LM COST_PRIM $CP"
  MIRT_BVH_LEAF_MIN=$LM MIRT_BVH_COST_PRIM=$CP timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dragon', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
  MIRT_BVH_LEAF_MIN=$LM MIRT_BVH_COST_PRIM=$CP timeout -k 10 200 python3 scripts/prof_frame.py --workload grid --reps 3 | tail -1 || exit 1
done; done
