#!/bin/bash
# Dev tool (GPU): bench.py's headline leg with 2 / 3 / 4 / 6 batches in flight on the one handle
for n in 2 3 4 6; do
  timeout -k 10 200 python bench.py --in-flight $n --no-streaming --no-cpu-baseline --no-c4c5 --steps 24 --warmup 6 2>/dev/null > /tmp/infl_$n.json
  N=$n python3 -c "import os,json; d=json.load(open('/tmp/infl_'+os.environ['N']+'.json')); print('in flight', os.environ['N'], 'ms/step', round(d['ms_per_step'],2), 'value', round(d['value']), 'one in flight ms', round(d['ms_per_step_one_in_flight'],2))"
done
