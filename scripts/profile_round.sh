#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats of the same command + PMC passes.
# Usage: scripts/profile_round.sh <tag>   -> files under gpurun_out/<tag>/ (copy the summaries into profiles/)
set -o pipefail
export TMPDIR=/tmp
# bench.py asks for 16 HIP hardware queues itself, but under rocprofv3 HIP is initialised before Python runs: say it here
export GPU_MAX_HW_QUEUES=16
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
T=gpurun_out/$1
mkdir -p "$T"
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 > "$T/bench.json" 2> "$T/bench.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$T/stats" -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$T/bench_under_rocprof.json" 2>> "$T/bench.err" || exit 1
# counters: one pass per block group, kernel-trace only (see MI355X guide: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$T/pmc_fetch" -- python3 scripts/prof_frame.py --reps 1 > "$T/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$T/pmc_write" -- python3 scripts/prof_frame.py --reps 1 > "$T/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$T/pmc_sq" -- python3 scripts/prof_frame.py --reps 1 > "$T/pmc_sq.log" 2>&1 || exit 1
echo done
