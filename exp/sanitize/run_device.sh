#!/bin/bash
# On a GPU machine, after build_device.sh:  bash exp/sanitize/run_device.sh [random-per-kind] [damaged]
# Leak checking is off (the HIP runtime keeps its own allocations until exit); everything else the sanitizers see ends the run.
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d /tmp/mirt_san.XXXXXX)
python3 exp/sanitize/make_texts.py "$T/texts" "${1:-25}" "${2:-40}"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
for chunk in $(ls "$T"/texts/* | xargs -n 60 echo | tr ' ' ','); do
    timeout -k 10 170 exp/sanitize/build/driver_device $(echo "$chunk" | tr ',' ' ')
done
rm -rf "$T"
