#!/bin/bash
# Run on the GPU box (through gpurun): counters, bench line and kernel trace of ONE workload.
# Usage: scripts/profile_config.sh <tag> <workload> [round]     workload: dragon | dragon8k | cornell | spheres | grid
#   -> raw files under gpurun_out/<tag>/<workload>/, summaries under gpurun_out/<tag>/profiles/<round>/ (copy them into
#      profiles/<round>/: the GPU box keeps nothing else) and, for bench.py to read in the same run, under profiles/<round>/.
# Every counter pass is a run of its own with --pmc only (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; at most 8 SQ,
# 4 TCP, 2 TA, 2 TD, 2 GRBM counters per pass), the program directly behind `--`.
set -o pipefail
export TMPDIR=/tmp
# bench.py asks for 16 HIP hardware queues itself, but under rocprofv3 HIP is initialised before Python runs: say it here
export GPU_MAX_HW_QUEUES=16
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
W=$2
ROUND=${3:-r03}
T=gpurun_out/$1/$W
D=gpurun_out/$1/profiles/$ROUND
mkdir -p "$T" "$D" "profiles/$ROUND"
LIM=120
[ "$W" = grid ] && LIM=400
pass() { # name, counters...
	local name=$1; shift
	timeout -s KILL $LIM rocprofv3 --pmc "$@" --output-format csv -d "$R/$T/pmc_$name" -- python3 scripts/prof_frame.py --workload "$W" --reps 1 > "$T/pmc_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$T/pmc_$name.log"; exit 1; }
}
timeout -k 10 $LIM python3 scripts/prof_frame.py --workload "$W" --reps 3 > "$T/frame.log" 2>&1 || exit 1
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
pass l1a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass l1b TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
timeout -k 10 120 python3 scripts/summarize_config.py "$1" "$W" "$D" pmc || exit 1
cp "$D/pmc_$W.json" "profiles/$ROUND/pmc_$W.json"
# the bench line of this workload (reads profiles/<round>/pmc_<workload>.json), then the kernel trace of the same command
# and of the same frames rendered one at a time
STEPS=30
[ "$W" = grid ] && STEPS=10
timeout -k 10 600 python3 bench.py --workload "$W" --steps $STEPS --warmup 5 > "$T/bench.json" 2> "$T/bench.err" || { tail -5 "$T/bench.err"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$T/stats" -- python3 bench.py --workload "$W" --steps $STEPS --warmup 5 --no-cpu-baseline > "$T/bench_under_rocprof.json" 2>> "$T/bench.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$T/stats_alone" -- python3 bench.py --workload "$W" --steps $STEPS --warmup 5 --pipeline 1 --no-cpu-baseline > "$T/bench_alone_under_rocprof.json" 2>> "$T/bench.err" || exit 1
timeout -k 10 120 python3 scripts/summarize_config.py "$1" "$W" "$D" stats || exit 1
echo "done $W"
