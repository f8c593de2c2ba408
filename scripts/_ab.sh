mkdir -p gpurun_out/ab
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload ${W:-dragon} --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err || { echo "$label FAILED"; tail -3 gpurun_out/ab/b.err; return; }
  python3 -c "
import json;d=json.load(open('gpurun_out/ab/b.json')); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['frame_alone_ms'],3))"
}
for i in 1 2 3; do run prev MIRT_LIB=exp/libmirt_prev.so; run bands MIRT_LIB=exp/libmirt_bands.so; run xcc MIRT_LIB=exp/libmirt_xcc.so; done
