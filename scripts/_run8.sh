mkdir -p gpurun_out/r02g
for lib in raytracer_amd/libmirt.so exp/libmirt_fw8_s17.so exp/libmirt_fw8_s16.so raytracer_amd/libmirt.so; do
for p in 1 8; do
  MIRT_LIB=$lib timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --pipeline $p > gpurun_out/r02g/bench_x.json 2> gpurun_out/r02g/bench_x.err || exit 1
  python3 -c "
import json;d=json.load(open('gpurun_out/r02g/bench_x.json')); print('$lib', $p, round(d['value']), round(d['ms_per_step'],3), round(d['frame_alone_ms'],3))"
done; done
