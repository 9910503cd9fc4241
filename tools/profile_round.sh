#!/bin/bash
# Per-round profiles (run on the MI355X box through gpurun from the repository root):  ROUND=r05 bash tools/profile_round.sh
#   kernel-trace + stats of the headline bench and of every side config, then separate PMC passes
#   (FETCH_SIZE, WRITE_SIZE: HBM-side traffic; SQ busy counters for the cfg2 kernels).
# rocprofv3 gets `python3 bench.py ...` directly after `--` (no env/bash hop); the side configs are profiled with
# --skip-headline so that they run in the profiled process itself (bench.py's default runs them in child processes).
# Outputs go to gpurun_out/prof_$ROUND/; `python tools/condense_round.py $ROUND` turns them into profiles/${ROUND}_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${ROUND:-r05}
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT
python3 - > $OUT/stamp.json <<'PY'
import hashlib, json
print(json.dumps({"so_sha256": hashlib.sha256(open("dctn_amd/libdctn_amd.so", "rb").read()).hexdigest()}))
PY
CFGS=${CFGS:-"cfg2_f32 cfg1 cfg3a cfg3a_bf16 cfg3b cfg4_r4 cfg4_r8 cfg4_r16 cfg4_eps36 cfg5"}   # CFGS="cfg4_r4 cfg4_r16" HEADLINE=0 re-takes only those
HEADLINE=${HEADLINE:-1}
if [ "$HEADLINE" = 1 ]; then
echo "[profile] headline kernel stats"
rocprofv3 --kernel-trace --stats -d $OUT/headline -o h --output-format csv -- python3 bench.py --configs none --no-cpu-baseline --steps 200 --warmup 20 > $OUT/headline.json 2> $OUT/headline.err || echo "headline stats failed"
fi
for cfg in $CFGS; do
  echo "[profile] $cfg kernel stats"
  rocprofv3 --kernel-trace --stats -d $OUT/$cfg -o k --output-format csv -- python3 bench.py --skip-headline --configs $cfg --no-cpu-baseline > $OUT/$cfg.json 2> $OUT/$cfg.err || echo "$cfg stats failed"
done
echo "[profile] PMC traffic passes"
if [ "$HEADLINE" = 1 ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/headline_fetch -o p --output-format csv -- python3 bench.py --configs none --no-cpu-baseline --steps 40 --warmup 20 --graph 0 > /dev/null 2> $OUT/headline_fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/headline_write -o p --output-format csv -- python3 bench.py --configs none --no-cpu-baseline --steps 40 --warmup 20 --graph 0 > /dev/null 2> $OUT/headline_write.err || echo "write failed"
fi
for cfg in $CFGS; do
  echo "[profile] $cfg traffic"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${cfg}_fetch -o p --output-format csv -- python3 bench.py --skip-headline --configs $cfg --no-cpu-baseline > /dev/null 2> $OUT/${cfg}_fetch.err || echo "$cfg fetch failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${cfg}_write -o p --output-format csv -- python3 bench.py --skip-headline --configs $cfg --no-cpu-baseline > /dev/null 2> $OUT/${cfg}_write.err || echo "$cfg write failed"
done
if [ "$HEADLINE" = 1 ]; then
echo "[profile] SQ busy counters, cfg2 kernels at B = 1024"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/headline_sq1 -o p --output-format csv -- python3 bench.py --configs none --no-cpu-baseline --steps 40 --warmup 20 --graph 0 > /dev/null 2> $OUT/headline_sq1.err || echo "sq1 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace -d $OUT/headline_sq2 -o p --output-format csv -- python3 bench.py --configs none --no-cpu-baseline --steps 40 --warmup 20 --graph 0 > /dev/null 2> $OUT/headline_sq2.err || echo "sq2 failed"
fi
SQCFGS=${SQCFGS:-"cfg2_f32 cfg4_r4 cfg4_r16 cfg3a cfg5"}
for cfg in $SQCFGS; do
  echo "[profile] SQ counters, $cfg"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${cfg}_sq1 -o p --output-format csv -- python3 bench.py --skip-headline --configs $cfg --no-cpu-baseline > /dev/null 2> $OUT/${cfg}_sq1.err || echo "$cfg sq1 failed"
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d $OUT/${cfg}_sq2 -o p --output-format csv -- python3 bench.py --skip-headline --configs $cfg --no-cpu-baseline > /dev/null 2> $OUT/${cfg}_sq2.err || echo "$cfg sq2 failed"
done
find $OUT -name "*kernel_trace.csv" -size +3M -delete   # keep the merge under the 64 MiB limit
# condense on the box (the raw counter files of ten configurations exceed what gpurun merges back) and keep only the summary
mkdir -p gpurun_out/profiles_$ROUND
python3 tools/condense_round.py $ROUND $OUT gpurun_out/profiles_$ROUND > gpurun_out/profiles_$ROUND/condense.log 2>&1 || echo "condense failed"
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT gpurun_out/profiles_$ROUND
echo "[profile] done"
