set -e
timeout -k 10 900 python -m pytest tests -q -m gpu -k "plus_linear or head or q2 or register_family or eps_golden" 2>&1 | tail -2
python bench.py --configs cfg2_f32 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print('headline', d['ms_per_step']*1e3, d['roofline']['kernels_us'], d['run'].get('us_per_step_at_one_step_per_graph_launch'))
        print('f32', d['side_summary'])
"
