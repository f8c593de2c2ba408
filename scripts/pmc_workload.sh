#!/bin/bash
# Counter passes over ONE frame of a generated workload, on the GPU box.  Usage: scripts/pmc_workload.sh <tag> <workload>
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
T=gpurun_out/$1
W=$2
mkdir -p "$T"
timeout -k 10 200 python3 scripts/prof_frame.py --workload "$W" --reps 3 > "$T/frame.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$T/pmc_fetch" -- python3 scripts/prof_frame.py --workload "$W" --reps 1 > "$T/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$T/pmc_write" -- python3 scripts/prof_frame.py --workload "$W" --reps 1 > "$T/pmc_write.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$R/$T/pmc_sq" -- python3 scripts/prof_frame.py --workload "$W" --reps 1 > "$T/pmc_sq.log" 2>&1 || exit 1
echo done
