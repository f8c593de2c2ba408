#!/bin/bash
# A/B of library builds on the GPU box, Stanford dragon only, interleaved: exp/abd.sh <rounds> <lib>...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    MIRT_LIB=$L timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), 'steady', round(d['steady_state']['value']), 'alone', round(d['frame_alone_ms'],3), 'sync', round(d['sync_host_ms'],3))" || exit 1
  done
done
