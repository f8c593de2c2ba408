mkdir -p gpurun_out/r02d
timeout -k 10 400 python bench.py --workload grid --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02d/bench_grid_refill.json 2> gpurun_out/r02d/bench_grid_refill.err || exit 1
MIRT_REFILL_MIN=64 timeout -k 10 400 python bench.py --workload grid --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02d/bench_grid_frame.json 2> gpurun_out/r02d/bench_grid_frame.err || exit 1
python3 -c "
import json
for n in ('refill','frame'):
    d=json.load(open('gpurun_out/r02d/bench_grid_%s.json'%n)); print(n, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_frame'],3) for k,v in d['kernels_alone'].items()})"
