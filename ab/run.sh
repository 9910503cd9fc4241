#!/bin/bash
# A/B of library variants on one box: bash ab/run.sh A B ...   (two interleaved rounds)
cd "$(dirname "$0")/.."
cp dctn_amd/libdctn_amd.so /tmp/lib_keep.so
for round in 1 2; do
  for v in "$@"; do
    cp ab/lib$v.so dctn_amd/libdctn_amd.so
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline $ABARGS > /tmp/ab_$v.json 2> /tmp/ab_$v.err || { echo "$v failed"; tail -3 /tmp/ab_$v.err; }
    python -c "
import json,sys; d=json.loads(open('/tmp/ab_$v.json').read().strip().splitlines()[-1]); print('$v round $round: step %.2f us  kernels %s  one-step-per-launch %.2f' % (d['ms_per_step']*1e3, {k: round(v,2) for k,v in d['roofline']['kernels_us'].items()}, d['config']['us_per_step_at_one_step_per_graph_launch']))"
  done
done
cp /tmp/lib_keep.so dctn_amd/libdctn_amd.so
