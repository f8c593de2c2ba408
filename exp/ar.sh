#!/bin/bash
for R in 64 60 48; do
  echo "== always-refill path, refillMin $R"
  MIRT_LIB=exp/libmirt_ar.so MIRT_REFILL_MIN=$R timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dragon', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
  MIRT_LIB=exp/libmirt_ar.so MIRT_REFILL_MIN=$R timeout -k 10 200 python3 scripts/prof_frame.py --reps 5 | tail -16 || exit 1
done
