#!/bin/bash
# A/B of an environment switch on the GPU box: exp/abenv.sh <rounds> "<workloads>" VAR val1 val2 ...
R=$1; W=$2; V=$3; shift 3
for i in $(seq $R); do
  for w in $W; do
    for x in "$@"; do
      env $V=$x timeout -k 10 300 python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $V=$x', round(d['value']), 'steady', round(d['steady_state']['value']), 'alone', round(d['frame_alone_ms'],3), 'sync', round(d['sync_host_ms'],3))" || exit 1
    done
  done
done
