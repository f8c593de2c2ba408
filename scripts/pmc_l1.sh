#!/bin/bash
# L1 / texture-path counters of one 4K Dragon frame (two separate passes; see profile_round.sh for the rules).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
T=gpurun_out/$1
mkdir -p "$T"
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d "$R/$T/pmc_l1a" -- python3 scripts/prof_frame.py --reps 1 > "$T/pmc_l1a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$R/$T/pmc_l1b" -- python3 scripts/prof_frame.py --reps 1 > "$T/pmc_l1b.log" 2>&1 || exit 1
echo done
