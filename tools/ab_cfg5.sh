set -e
timeout -k 10 900 python -m pytest tests -q -m gpu -k "bigcore or eps_golden or eps_vs_oracle or cfg3 or fullsize" 2>&1 | tail -2
for c in cfg3b cfg3a; do
python bench.py --skip-headline --configs $c --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{') or l.startswith('['):
        d=json.loads(l); e=d[0] if isinstance(d,list) else d
        e = e.get('configs',[e])[0] if isinstance(e,dict) and 'configs' in e else e
        print('$c', e.get('ms_per_step'))
"
done
