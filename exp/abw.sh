#!/bin/bash
# A/B of library builds on the GPU box over several workloads: exp/abw.sh "<workloads>" <lib>...
W=$1; shift
for w in $W; do
  for L in "$@"; do
    MIRT_LIB=$L timeout -k 10 300 python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $L', round(d['value']), 'steady', round(d['steady_state']['value']), 'alone', round(d['frame_alone_ms'],3), 'sync', round(d['sync_host_ms'],3))" || exit 1
  done
done
