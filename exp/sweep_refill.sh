#!/bin/bash
for R in 36 48 64; do
  echo "== grid refillMin $R"
  MIRT_REFILL_MIN=$R timeout -k 10 200 python3 scripts/prof_frame.py --workload grid --reps 3 | tail -1 || exit 1
done
for R in 32 40 48 56 64; do
  echo "== dragon refillMin $R"
  MIRT_REFILL_MIN=$R timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dragon', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
done
