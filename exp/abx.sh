#!/bin/bash
for L in "$@"; do
  echo "== $L"
  for W in dragon grid; do
  MIRT_LIB=$L timeout -k 10 300 python3 bench.py --workload $W --steps 30 --warmup 8 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
  done
  MIRT_LIB=$L timeout -k 10 200 python3 scripts/prof_frame.py --workload grid --reps 3 | tail -1 || exit 1
done
