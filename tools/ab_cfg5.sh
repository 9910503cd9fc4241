set -e
run() { python bench.py --skip-headline --configs cfg5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{') or l.startswith('['):
        d=json.loads(l); e=d[0] if isinstance(d,list) else d
        e = e.get('configs',[e])[0] if isinstance(e,dict) and 'configs' in e else e
        r=e.get('roofline',{})
        print('$1', e.get('ms_per_step'), r.get('launch_us'), r.get('fwd_us'))
"; }
timeout -k 10 300 python -m pytest tests -q -m gpu -k "logmatmulexp or lme or fold or head or linear" 2>&1 | tail -2
run DEFAULT
run DEFAULT
python bench.py --skip-headline --configs cfg3b --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{') or l.startswith('['):
        d=json.loads(l); e=d[0] if isinstance(d,list) else d
        e = e.get('configs',[e])[0] if isinstance(e,dict) and 'configs' in e else e
        print('cfg3b', e.get('ms_per_step'))
"
