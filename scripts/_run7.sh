mkdir -p gpurun_out/r02g
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r02g/pytest.log 2>&1 || { tail -20 gpurun_out/r02g/pytest.log; exit 1; }
tail -1 gpurun_out/r02g/pytest.log
MIRT_VERBOSE=1 timeout -k 10 300 python exp/create_phases.py 2>&1 | tail -9
timeout -k 10 300 python bench.py --workload cornell --trace-mode fast --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02g/bench_cornell_fast.json 2> gpurun_out/r02g/err || exit 1
python3 -c "
import json;d=json.load(open('gpurun_out/r02g/bench_cornell_fast.json')); print('cornell fast', round(d['value']), d['rays_walked_per_frame'], d['rays_per_frame'])"
