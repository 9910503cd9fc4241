#!/bin/bash
# kernel-trace + stats of one python invocation:  bash tools/prof_py.sh <tag> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
O=gpurun_out/prof_$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o k --output-format csv -- python3 "$@" > $O/out.txt 2> $O/err.txt
python3 - "$O" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g=(r["Kernel_Name"][:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or "")
    acc[g].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
rows=sorted(acc.items(), key=lambda kv:-sum(kv[1]))
for (n,g,w),v in rows[:28]:
    print(f"{n:70s} grid={g:>9s} n={len(v):4d} avg={sum(v)/len(v):8.1f} us  total={sum(v)/1e3:8.2f} ms")
PY
find $O -name "*kernel_trace.csv" -delete
tail -5 $O/out.txt
