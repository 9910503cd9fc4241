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
cp dctn_amd/libdctn_amd.so /tmp/base.so
run BASE
for v in A B C D E; do cp tools/lme_variants/lib_$v.so dctn_amd/libdctn_amd.so; run $v; done
cp /tmp/base.so dctn_amd/libdctn_amd.so
run BASE
