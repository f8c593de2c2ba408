#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats of the same command + PMC passes.
# Usage: scripts/profile_round.sh <tag> [round]  -> raw files under gpurun_out/<tag>/ and the summaries under profiles/<round>/
# (copy profiles/<round>/ back from gpurun_out/<tag>/profiles/: the GPU box keeps nothing else).
set -o pipefail
export TMPDIR=/tmp
# bench.py asks for 16 HIP hardware queues itself, but under rocprofv3 HIP is initialised before Python runs: say it here
export GPU_MAX_HW_QUEUES=16
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
T=gpurun_out/$1
ROUND=${2:-r02}
mkdir -p "$T"
# counters first, one pass per block group, kernel-trace only (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2): bench.py reads their summary
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$T/pmc_fetch" -- python3 scripts/prof_frame.py --workload dragon --reps 1 > "$T/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$T/pmc_write" -- python3 scripts/prof_frame.py --workload dragon --reps 1 > "$T/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$T/pmc_sq" -- python3 scripts/prof_frame.py --workload dragon --reps 1 > "$T/pmc_sq.log" 2>&1 || exit 1
timeout -k 10 120 python3 scripts/summarize_profile.py "$1" "$T/profiles/$ROUND" pmc || exit 1
mkdir -p "profiles/$ROUND" && cp "$T/profiles/$ROUND/pmc_frame.json" "profiles/$ROUND/pmc_frame.json"
# the bench line (reads profiles/<round>/pmc_frame.json for traffic and VALU issue)
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 > "$T/bench.json" 2> "$T/bench.err" || exit 1
# kernel trace of the same command, and of the same frames rendered one at a time (--pipeline 1): the average launch duration
# of the second must agree with bench.json's kernels_alone / roofline.launch_ms_alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$T/stats" -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$T/bench_under_rocprof.json" 2>> "$T/bench.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$T/stats_alone" -- python3 bench.py --steps 30 --warmup 5 --pipeline 1 --no-cpu-baseline > "$T/bench_alone_under_rocprof.json" 2>> "$T/bench.err" || exit 1
timeout -k 10 120 python3 scripts/summarize_profile.py "$1" "$T/profiles/$ROUND" stats || exit 1
echo done
