for LM in 1 2 3; do for CP in 1 1.5 2 2.5 3.5 5; do
  MIRT_BVH_LEAF_MIN=$LM MIRT_BVH_COST_PRIM=$CP timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LEAF_MIN $LM COST_PRIM $CP dragon', round(d['value']), 'steady', round(d['steady_state']['value']), 'alone', round(d['frame_alone_ms'],3))" || exit 1
done; done
