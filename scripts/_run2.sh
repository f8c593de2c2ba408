mkdir -p gpurun_out/r02k
run() { label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload grid --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/r02k/b.json 2> gpurun_out/r02k/b.err || { echo "$label FAILED"; tail -3 gpurun_out/r02k/b.err; return; }
  python3 -c "
import json;d=json.load(open('gpurun_out/r02k/b.json')); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['frame_alone_ms'],3), {k:round(v['ms_per_frame'],3) for k,v in d['kernels_alone'].items()})"
}
run frame_new MIRT_REFILL_MIN=64
run frame_prev MIRT_REFILL_MIN=64 MIRT_LIB=exp/libmirt_prev.so
run perdepth MIRT_X=1
