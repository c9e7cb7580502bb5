#!/bin/bash
# Dev probe (GPU): sample the shader clock and power while a GEMM probe runs a sustained loop.
cd "$(dirname "$0")"
./gemm_solo 6 $1 &
PID=$!
sleep 2
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' '; echo
  sleep 0.6
done
wait $PID
