#!/bin/bash
for P in 1 2 4 6 8 12 16; do
  timeout -k 10 200 python3 bench.py --steps 48 --warmup 16 --pipeline $P --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline $P dragon', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4))" || exit 1
done
