mkdir -p gpurun_out/r02h
run() { # label, env...
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02h/b.json 2> gpurun_out/r02h/b.err || { echo "$label FAILED"; tail -3 gpurun_out/r02h/b.err; return; }
  python3 -c "
import json;d=json.load(open('gpurun_out/r02h/b.json')); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['frame_alone_ms'],3), {k:round(v['ms_per_frame'],3) for k,v in d['kernels_alone'].items()})"
}
run frame MIRT_X=0
for r in 64 48 32 24 16 8; do run perdepth_refill$r MIRT_SINGLE_LAUNCH=0 MIRT_REFILL_MIN=$r; done
