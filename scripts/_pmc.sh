set -o pipefail
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
W=$1; T=gpurun_out/pmc_$W; R=$(pwd); mkdir -p $T
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$T/pmc_fetch" -- python3 scripts/prof_frame.py --workload $W --reps 1 > "$T/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$T/pmc_write" -- python3 scripts/prof_frame.py --workload $W --reps 1 > "$T/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$T/pmc_sq" -- python3 scripts/prof_frame.py --workload $W --reps 1 > "$T/pmc_sq.log" 2>&1 || exit 1
timeout -k 10 120 python3 scripts/summarize_profile.py pmc_$W "$T/sum" pmc || exit 1
tail -12 $T/pmc_sq.log
