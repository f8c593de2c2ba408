#!/bin/bash
# dragon, spheres, cornell bench lines per library
for L in "$@"; do
  echo "== $L"
  for W in dragon spheres cornell; do
  MIRT_LIB=$L timeout -k 10 200 python3 bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', round(d['value']), 'Mrays/s', d['ms_per_step'])" || exit 1
  done
done
