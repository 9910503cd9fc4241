set -e
run() { python bench.py --skip-headline --configs cfg3b,cfg3a,cfg4_eps36 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print('$1', [(e['workload'][:10], round(e['ms_per_step'],4)) for e in d['configs']])
"; }
cp dctn_amd/libdctn_amd.so /tmp/base.so
run BASE
cp tools/variants/lib_bc_slp.so dctn_amd/libdctn_amd.so; run BCSLP
cp /tmp/base.so dctn_amd/libdctn_amd.so; run BASE
cp tools/variants/lib_bc_slp.so dctn_amd/libdctn_amd.so; run BCSLP
