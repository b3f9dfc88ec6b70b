#!/bin/bash
# Round-4 GPU-box visits: stages picked on the command line, everything lands under gpurun_out/.
#   tools/gpu_r4.sh "bwdtest rsab"        (see the case labels)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
STAGES=${1:-"test"}
prof() {  # prof <name> <pmc counters or ""> -- cmd...   (counters in their own pass, kernel-trace only)
  local name=$1; shift
  local ctrs=$1; shift
  shift
  rm -rf $O/$name
  if [ -n "$ctrs" ]; then
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log)
    python tools/prof_summary.py pmc $O/$name > $O/$name.json 2>> $O/$name.log
  else
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log)
    python tools/prof_summary.py stats $O/$name > $O/$name.txt 2>> $O/$name.log
  fi
  find $O/$name -name "*.csv" -size +2M -delete
}
ab() {   # ab "option:v0,v1,.." "cases"  -> one line per case: the times under each value
  echo "== $1 (sweep cases $2)"
  SWEEP_AB=$1 timeout 900 python tools/sweep.py --big --cases $2 2>&1 | python -c "
import sys,json
last=None
for l in sys.stdin:
  if not l.startswith('{'):
    if 'rror' in l or 'Traceback' in l: print(l.strip()[:300])
    continue
  d=json.loads(l)
  if 'ab' in d: last=d
  elif last: print(d['case'][:78].ljust(78), last['values'], last['us']); last=None"
}
for st in $STAGES; do
  case $st in
    test)
      timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/test.log 2>&1; echo "pytest rc=$?" >> $O/test.log; tail -25 $O/test.log;;
    bwdtest)
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "backward or random" --durations=10 > $O/bwdtest.log 2>&1; echo "pytest rc=$?" >> $O/bwdtest.log; tail -40 $O/bwdtest.log;;
    rstest)   # the row-sorted buckets only (first contact)
      timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowsort" --durations=5 > $O/rstest.log 2>&1; echo "pytest rc=$?" >> $O/rstest.log; tail -40 $O/rstest.log;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log;;
    rsab)   # row-sorted buckets on / off inside one process (ragged, config-5 shape), then C ABI
      (ab "bwd_rowsort_ratio:0,8,0,8" "b,h"
       ab "bwd_rowsort_ratio:0,16,0,16" "c"
       for ratio in 0 8; do
         for w in r d; do
           HBK_BWD_ROWSORT_RATIO=$ratio timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk " | sed "s/^/rowsort_ratio=$ratio  /"
         done
       done) > $O/rsab.log 2>&1; cut -c1-260 $O/rsab.log;;
    rsprof)
      prof prof_ragged "" -- python $R/tools/sweep.py --cases b
      grep -E "bwd_|kernel  " $O/prof_ragged.txt | head -20;;
    rsstamps)   # per-phase trace of the row-sorted reduce (probe build with stamps)
      (for w in R r d; do HBK_STAMP_NAMES=rowsort LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops $w; done
       HBK_BWD_ROWSORT_RATIO=0 LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops R) > $O/rsstamps.log 2>&1; cut -c1-600 $O/rsstamps.log;;
    rspmc)
      prof pmc_rs_sq1 "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" -- $R/tools/bin/bench_ops R
      prof pmc_rs_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- $R/tools/bin/bench_ops R
      prof pmc_rs_sq3 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" -- $R/tools/bin/bench_ops R
      prof pmc_rs_tcc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_HIT_sum" -- $R/tools/bin/bench_ops R
      for f in pmc_rs_sq1 pmc_rs_sq2 pmc_rs_sq3 pmc_rs_tcc; do echo "== $f"; tail -1 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  if 'bwd_' in k: print(k[:60].ljust(60), {c:round(x['mean']) for c,x in v.items()})
PY
      done;;
    rsab2)   # flat walk + staged scatter: in-process toggles, then the probe builds through the C ABI
      (ab "bwd_rowsort_ratio:0,8,0,8" "b,h"
       ab "bwd_scatter_staged:0,1,0,1" "b"
       ab "bwd_xcd:0,1,0,1" "b"
       for v in tools/bin/v_*; do
         for w in R r d; do
           LD_LIBRARY_PATH=$R/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s|^|$(basename $v)  |"
         done
       done
       HBK_BWD_ROWSORT_RATIO=0 timeout 300 tools/bin/bench_ops r 2>&1 | grep group_lookup_bwd | sed "s|^|hashed  |") > $O/rsab2.log 2>&1; cut -c1-260 $O/rsab2.log;;
    rstcc)   # L2 hit rate / fabric requests of the ragged case (3 calls per pass)
      export HBK_BENCH_ITERS=2
      prof pmc_rs_tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- $R/tools/bin/bench_ops R
      unset HBK_BENCH_ITERS
      for f in pmc_rs_tcc1; do echo "== $f"; tail -1 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  if 'bwd_' in k: print(k[:60].ljust(60), {c:round(x['mean']) for c,x in v.items()})
PY
      done;;
    rsstamps2)
      (HBK_STAMP_NAMES=rowsort LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops R) > $O/rsstamps2.log 2>&1; grep "reduce kernel" $O/rsstamps2.log | cut -c1-600;;
    profbh)   # kernel stats of the ragged case and of the config-5 shape
      prof prof_ragged "" -- python $R/tools/sweep.py --cases b
      prof prof_cfg5 "" -- python $R/tools/sweep.py --cases h
      echo "== ragged"; grep -E "bwd_|kernel  " $O/prof_ragged.txt | cut -c1-150 | head -14
      echo "== cfg5"; grep -E "bwd_|kernel  " $O/prof_cfg5.txt | cut -c1-150 | head -40;;
    rsab3)   # the ratio that picks row-sorted buckets; probe builds of the walk width
      (ab "bwd_rowsort_ratio:8,16,32,8,16,32" "c,h"
       ab "bwd_rowsort_ratio:8,16,8,16" "d,f"
       for v in tools/bin/v_*; do
         for w in R r d; do
           LD_LIBRARY_PATH=$R/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s|^|$(basename $v)  |"
         done
       done) > $O/rsab3.log 2>&1; cut -c1-260 $O/rsab3.log;;
    rsab4)   # job size by row width
      (ab "bwd_rowsort_pos:0,32,64,0,32,64" "h"
       ab "bwd_rowsort_pos:0,32,64,0,32,64" "d,f"
       for pos in 0 32 64; do
         HBK_BWD_ROWSORT_POS=$pos timeout 300 tools/bin/bench_ops r 2>&1 | grep group_lookup_bwd | sed "s|^|pos=$pos  |"
         HBK_BWD_ROWSORT_POS=$pos timeout 300 tools/bin/bench_ops d 2>&1 | grep group_lookup_bwd | sed "s|^|pos=$pos  |"
         HBK_BWD_ROWSORT_POS=$pos timeout 300 tools/bin/bench_ops w 2>&1 | grep group_lookup_bwd | sed "s|^|pos=$pos  |"
       done) > $O/rsab4.log 2>&1; cut -c1-260 $O/rsab4.log;;
    shardtest)
      timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_configs.py -q -m gpu -x --durations=8 > $O/shardtest.log 2>&1; echo "pytest rc=$?" >> $O/shardtest.log; tail -25 $O/shardtest.log;;
    deduptest)
      timeout 900 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x -k "dedup" --durations=8 > $O/deduptest.log 2>&1; echo "pytest rc=$?" >> $O/deduptest.log; tail -25 $O/deduptest.log;;
    dedupsweep)
      timeout 900 python tools/sweep.py --big --cases k > $O/dedupsweep.log 2>&1; echo "rc=$?" >> $O/dedupsweep.log; cut -c1-400 $O/dedupsweep.log;;
    rsab5)
      (ab "bwd_rowsort_ratio:0,8,0,8" "b,h,c") > $O/rsab5.log 2>&1; cut -c1-260 $O/rsab5.log;;
    pysweep)   # the Python forms with fresh tensors
      timeout 900 python tools/sweep.py --cases e > $O/pysweep.log 2>&1; echo "rc=$?" >> $O/pysweep.log; cut -c1-200 $O/pysweep.log;;
    traffic)   # profiles/hbm_traffic.json of this round (bench.py's roofline.traffic)
      timeout 900 python tools/hbm_traffic.py r04 $O/hbm_traffic.json > $O/traffic.log 2>&1; echo "rc=$?" >> $O/traffic.log; tail -3 $O/traffic.log | cut -c1-600;;
    bitmap)
      timeout 120 tools/bin/bitmap_probe > $O/bitmap_probe.log 2>&1; cat $O/bitmap_probe.log;;
    benchsharded)
      timeout 600 python bench.py --gpus 1 --sharded --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" > $O/bench_sharded.jsonl; cut -c1-1500 $O/bench_sharded.jsonl;;
    hottest)
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -q -m gpu -x -k "follow_the_data or hot_row or call_cache" --durations=5 > $O/hottest.log 2>&1; echo "pytest rc=$?" >> $O/hottest.log; tail -15 $O/hottest.log;;
    *) echo "unknown stage $st";;
  esac
done
