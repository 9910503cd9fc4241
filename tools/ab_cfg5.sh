set -e
run() { python bench.py --skip-headline --configs cfg4_r4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{') or l.startswith('['):
        d=json.loads(l); e=d[0] if isinstance(d,list) else d
        e = e.get('configs',[e])[0] if isinstance(e,dict) and 'configs' in e else e
        print('$1', e.get('ms_per_step'), (e.get('roofline') or {}).get('launch_us'))
"; }
cp dctn_amd/libdctn_amd.so /tmp/base.so
run BASE
cp tools/variants/lib_slp.so dctn_amd/libdctn_amd.so; run SLP; run SLP
timeout -k 10 600 python -m pytest tests -q -m gpu -k "sbs or convsbs" 2>&1 | tail -2
timeout -k 10 300 python tools/time_sbs_classifier.py 2 4 2>&1 | grep "reference form"
cp /tmp/base.so dctn_amd/libdctn_amd.so
run BASE
timeout -k 10 300 python tools/time_sbs_classifier.py 2 4 2>&1 | grep "reference form"
