mkdir -p gpurun_out/r02c
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_streams.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02c/pytest2.log 2>&1 || { tail -20 gpurun_out/r02c/pytest2.log; exit 1; }
tail -2 gpurun_out/r02c/pytest2.log
for p in 1 2 4 8; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --pipeline $p > gpurun_out/r02c/bench_single_p$p.json 2> gpurun_out/r02c/bench_single_p$p.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c/bench_single*.json')):
    d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_frame'],3) for k,v in d['kernels_alone'].items()})
PY
