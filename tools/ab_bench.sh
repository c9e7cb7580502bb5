#!/bin/bash
# A/B timing of two builds of libpfhip.so inside ONE GPU session (box-to-box spread is larger than most kernel changes):
#   tools/ab_bench.sh build/ab/libpfhip_old.so [rounds]      -> alternates the in-tree library (B) with the given one (A)
# Prints ms_per_step of every run; bench.py flags as in the driver's command, without the CPU baseline and streaming legs.
set -e
A=$(realpath "$1"); N=${2:-3}
for i in $(seq 1 "$N"); do
  for which in A B; do
    if [ $which = A ]; then export PFHIP_LIB=$A; else unset PFHIP_LIB; fi
    python3 bench.py --steps 20 --warmup 5 --no-streaming --no-cpu-baseline --no-profile 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', 'back to back', round(d['ms_per_step_one_in_flight'],3), 'ms;', d['config']['in_flight'], 'in flight', round(d['ms_per_step'],3), 'ms')"
  done
done
