cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_sbs; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/a -o p --output-format csv -- python3 bench.py --skip-headline --configs cfg4_r16 --no-cpu-baseline > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d $O/b -o p --output-format csv -- python3 bench.py --skip-headline --configs cfg4_r16 --no-cpu-baseline > /dev/null 2> $O/b.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace -d $O/c -o p --output-format csv -- python3 bench.py --skip-headline --configs cfg4_r16 --no-cpu-baseline > /dev/null 2> $O/c.err
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,collections
for t in "abc":
    f=glob.glob(f"gpurun_out/pmc_sbs/{t}/**/*counter_collection.csv",recursive=True)
    if not f: print(t,"no file"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "convsbs" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in acc.items():
        print(k, {c: round(sum(v)/len(v)) for c,v in cs.items()}, len(next(iter(cs.values()))))
PY
