#!/bin/bash
# A/B of library builds on the GPU box: exp/ab.sh <lib>...   (each: dragon 4K bench, grid frame times)
for L in "$@"; do
  echo "== $L"
  MIRT_LIB=$L timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dragon', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
  MIRT_LIB=$L timeout -k 10 200 python3 scripts/prof_frame.py --workload grid --reps 3 | tail -1 || exit 1
done
