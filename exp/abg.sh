#!/bin/bash
# grid frame times only: exp/abg.sh <lib>...
for L in "$@"; do
  echo "== $L"
  MIRT_LIB=$L timeout -k 10 200 python3 scripts/prof_frame.py --workload grid --reps 3 | tail -1 || exit 1
done
