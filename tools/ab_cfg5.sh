set -e
timeout -k 10 900 python -m pytest tests -q -m gpu -k "band or sbs or convsbs" 2>&1 | tail -2
python bench.py --skip-headline --configs cfg4_r16,cfg4_r8,cfg4_r4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        for e in d['configs']: print(e['workload'][:12], e['ms_per_step'])
"
timeout -k 10 300 python tools/time_sbs_classifier.py 8 16 2>&1 | grep "reference form"
