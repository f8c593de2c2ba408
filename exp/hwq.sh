#!/bin/bash
for Q in 16 24; do for P in 8 10 12 16; do
  GPU_MAX_HW_QUEUES=$Q timeout -k 10 200 python3 bench.py --steps 64 --warmup 20 --pipeline $P --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq $Q pipeline $P dragon', round(d['value']), 'Mrays/s', round(d['ms_per_step'],4))" || exit 1
done; done
