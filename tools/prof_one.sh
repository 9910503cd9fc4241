#!/bin/bash
# kernel-trace + stats of one bench invocation:  bash tools/prof_one.sh <tag> <bench.py args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/prof_$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o k --output-format csv -- python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete
python3 - "$O" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:90].replace("\n"," "), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
tail -c 600 $O/bench.json
