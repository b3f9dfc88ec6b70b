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
    adavar)   # probe builds of the Adagrad walk on the config-5 shape (two passes each)
      (for rep in 1 2; do for v in tools/bin/v_*; do
         HBK_LIBRARY=$R/$v/libhbk_core.so timeout 600 python tools/sweep.py --cases h 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$(basename $v)'.ljust(14), d['case'][:60].ljust(60), d['us'])"
       done; done) > $O/adavar.log 2>&1; cat $O/adavar.log;;
    intpmc)   # what bounds the integer ops: SQ / TCC counters of partition (reference shape, 26 x 65536 at P = 8 / 64) and unique
      export HBK_BENCH_ITERS=3
      for w in q p P u; do
        prof pmc_int_${w}_1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- $R/tools/bin/bench_ops $w
        prof pmc_int_${w}_2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" -- $R/tools/bin/bench_ops $w
        prof pmc_int_${w}_3 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" -- $R/tools/bin/bench_ops $w
        prof prof_int_${w} "" -- $R/tools/bin/bench_ops $w
      done
      unset HBK_BENCH_ITERS
      python - $O > $O/r04_integer_counters.txt <<'PY'
import json,sys,os
O=sys.argv[1]
names={'q':'partition, reference benchmark shape (100 x 100000 int32, P = 8)','p':'partition 26 x 65536 int64, P = 8 (one launch)','P':'partition 26 x 65536 int64, P = 64','u':'unique 26 x 65536 int64'}
print("SQ / TCC counters (mean per launch; one --pmc pass per group, kernel-trace only) and kernel durations of the integer ops, tools/bin/bench_ops <q|p|P|u>")
for w,title in names.items():
  print("==", title)
  rows={}
  for k in (1,2,3):
    f=f'{O}/pmc_int_{w}_{k}.json'
    if not os.path.exists(f): continue
    for kern,v in json.load(open(f)).items():
      if kern.startswith('hbk::') or 'hbk::' in kern: rows.setdefault(kern[:70],{}).update({c:round(x['mean']) for c,x in v.items()})
  for kern,v in sorted(rows.items()): print(' ',kern.ljust(70), v)
  t=f'{O}/prof_int_{w}.txt'
  if os.path.exists(t):
    for l in open(t):
      if 'hbk::' in l or l.startswith('kernel'): print('  ',l.rstrip()[:150])
PY
      cat $O/r04_integer_counters.txt | cut -c1-330;;
    final)   # the round's evidence, copied to profiles/r04_* afterwards
      prof prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      cp $O/prof_bench.txt $O/r04_bench_kernel_stats.txt
      find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bench_rocprofv3_kernel_stats.csv
      timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/r04_bench_lines.jsonl
      for w in fp32 fp16; do
        timeout 300 python bench.py --gpus 1 --sharded --wire $w --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" >> $O/r04_bench_lines.jsonl
      done
      timeout 300 tools/bin/bench_ops > $O/r04_bench_ops.txt 2>&1
      for ratio in 8 0; do for w in R r d; do HBK_BWD_ROWSORT_RATIO=$ratio timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk " | sed "s/^/bwd_rowsort_ratio=$ratio  /"; done; done >> $O/r04_bench_ops.txt
      prof prof_bwd "" -- $R/tools/bin/bench_ops b
      prof prof_bwd_step "" -- $R/tools/bin/bench_ops s
      (echo "== config-2 backward, C ABI (tools/bin/bench_ops b): row-sorted buckets (rows = 15 x ids, dim 16: ratio 16)"; cat $O/prof_bwd.txt; echo; echo "== + fused SGD step / step only (bench_ops s)"; cat $O/prof_bwd_step.txt) > $O/r04_bwd_kernel_stats.txt
      prof prof_ragged "" -- python $R/tools/sweep.py --cases b
      cp $O/prof_ragged.txt $O/r04_ragged_kernel_stats.txt
      prof prof_cfg5 "" -- python $R/tools/sweep.py --cases h
      cp $O/prof_cfg5.txt $O/r04_cfg5_bwd_kernel_stats.txt
      export HBK_BENCH_ITERS=3
      prof pmc_rs_sq1 "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" -- $R/tools/bin/bench_ops R
      prof pmc_rs_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- $R/tools/bin/bench_ops R
      prof pmc_rs_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- $R/tools/bin/bench_ops R
      HBK_BWD_SCATTER_STAGED=0 prof pmc_rs_tcc_direct "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" -- $R/tools/bin/bench_ops R
      unset HBK_BENCH_ITERS
      python - $O > $O/r04_rowsort_counters.txt <<'PY'
import json,sys
O=sys.argv[1]
print("SQ / TCC counters of the ragged backward's kernels (tools/bin/bench_ops R: 26 x 524288 ids, 8 per segment, mean, dim 16, 1M rows; mean per launch; one --pmc pass per group)")
for f in ('pmc_rs_sq1','pmc_rs_sq2','pmc_rs_tcc'):
  d=json.load(open(f'{O}/{f}.json'))
  for k,v in sorted(d.items()):
    if 'bwd_' in k: print(k[:64].ljust(64), {c:round(x['mean']) for c,x in v.items()})
print("== the same with the direct pair scatter (HBK_BWD_SCATTER_STAGED=0)")
d=json.load(open(f'{O}/pmc_rs_tcc_direct.json'))
for k,v in sorted(d.items()):
  if 'scatter' in k: print(k[:64].ljust(64), {c:round(x['mean']) for c,x in v.items()})
PY
      (HBK_STAMP_NAMES=rowsort LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops R; HBK_STAMP_NAMES=rowsort LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops d) > $O/r04_rowsort_trace.txt 2>&1
      prof prof_sharded "" -- python $R/bench.py --gpus 1 --sharded --steps 30 --warmup 5 --cpu-seconds 0 --tune-steps 0 --no-secondary
      cp $O/prof_sharded.txt $O/r04_sharded_w1_kernel_stats.txt
      timeout 2400 python tools/sweep.py --big --cases a,b,c,d,e,f,g,h,i,j,k,l 2>/dev/null | grep "^{" > $O/r04_sweep.jsonl
      (echo "Library options toggled INSIDE one process on the same tensors (tools/sweep.py SWEEP_AB=option:values): case, values in the order run, microseconds per call under each."
       ab "bwd_rowsort_ratio:0,8,0,8" "b,h,c"
       ab "bwd_scatter_staged:0,1,0,1" "b"
       ab "bwd_rowsort_pos:0,64,0,64" "h") > $O/r04_inprocess_ab.txt 2>&1
      timeout 120 tools/bin/bitmap_probe > $O/r04_bitmap_probe.txt 2>&1
      ls -la $O/r04_*;;
    cfg4chk)
      (for rep in 1 2; do for ratio in 8 0; do
         HBK_BWD_ROWSORT_RATIO=$ratio timeout 600 python tools/sweep.py --big --cases d 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l)
  if 'bwd' in d['case']: print('ratio=$ratio rep=$rep', d['case'][:70].ljust(70), d['us'])"
       done; done) > $O/cfg4chk.log 2>&1; cat $O/cfg4chk.log;;
    fuzz)
      timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x --durations=5 > $O/fuzz.log 2>&1; echo "pytest rc=$?" >> $O/fuzz.log; tail -15 $O/fuzz.log;;
    fuzzmore)   # fresh random draws (hypothesis seed from the clock), three rounds
      for k in 1 2 3; do HBK_FUZZ_RANDOM=1 HBK_FUZZ_SCALE=10 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -p no:cacheprovider > $O/fuzz_$k.log 2>&1; echo "pytest rc=$?" >> $O/fuzz_$k.log; tail -4 $O/fuzz_$k.log; done;;
    variants)   # probe builds through the C ABI, two passes
      (for rep in 1 2; do for v in tools/bin/v_*; do for w in R r d b; do
         LD_LIBRARY_PATH=$R/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s|^|$(basename $v)  |"
       done; done; done) > $O/variants.log 2>&1; cut -c1-200 $O/variants.log;;
    rsab6)
      (ab "bwd_group_cols:0,13,7,0,13,7" "b"
       for gc in 0 13 7; do HBK_BWD_GROUP_COLS=$gc timeout 300 tools/bin/bench_ops R 2>&1 | grep group_lookup_bwd | sed "s|^|group_cols=$gc  |"; done
       for gc in 0 8 4; do HBK_BWD_GROUP_COLS=$gc timeout 300 tools/bin/bench_ops r 2>&1 | grep group_lookup_bwd | sed "s|^|group_cols=$gc  |"; done) > $O/rsab6.log 2>&1; cut -c1-200 $O/rsab6.log;;
    synctest)
      timeout 900 python -m pytest tests/test_gpu_sync.py -q -m gpu --durations=5 > $O/synctest.log 2>&1; echo "pytest rc=$?" >> $O/synctest.log; tail -30 $O/synctest.log;;
    autosweep)
      timeout 900 python tools/sweep.py --big --cases l > $O/autosweep.log 2>&1; echo "rc=$?" >> $O/autosweep.log; cut -c1-300 $O/autosweep.log;;
    rsab7)
      (ab "bwd_group_cols:0,45,34,0,45,34" "h") > $O/rsab7.log 2>&1; cut -c1-200 $O/rsab7.log;;
    ilab)
      (ab "fwd_interleave:0,1,2,0,1,2" "i,l"; python -m pytest tests/test_gpu_parity.py -q -x -k "dense or block or stride" 2>&1 | tail -3) > $O/ilab.log 2>&1; cut -c1-220 $O/ilab.log;;
    ilab2)
      (ab "fwd_interleave:2,3,2,3" "a,b,d") > $O/ilab2.log 2>&1; cut -c1-220 $O/ilab2.log;;
    iltest)
      (python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -8
       python tools/sweep.py --cases i,l) > $O/iltest.log 2>&1; cut -c1-250 $O/iltest.log;;
    tail)
      (python tools/scratch/bench_tail.py; for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 | cut -c1-260; done) > $O/tail.log 2>&1; cut -c1-300 $O/tail.log;;
    tail2)
      (for i in 1 2 3; do python tools/scratch/bench_tail.py fresh; done; python tools/scratch/bench_tail.py fresh per_step_events
       for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0.5; done) > $O/tail2.log 2>&1; cut -c1-400 $O/tail2.log;;
    tail3)
      (for i in 1 2 3 4 5 6 7 8; do python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0.5; done) > $O/tail3.log 2>&1;;
    tail4)
      (for i in 1 2 3 4 5 6 7 8; do python tools/scratch/bench_tail.py fresh flags; done) > $O/tail4.log 2>&1;;
    tail5)
      cd /tmp; export TMPDIR=/tmp
      for i in 1 2 3 4 5 6; do
        rocprofv3 --kernel-trace --output-format csv -d $O/tail5_$i -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > $O/tail5_$i.log 2>&1
      done
      cd $R;;
    tail6)
      (for i in 1 2 3 4 5 6 7 8 9 10; do HBK_BENCH_STAMPS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | cut -c1-420; done) > $O/tail6.log 2>&1;;
    tail7)
      (for v in "HSA_ENABLE_INTERRUPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "GPU_MAX_HW_QUEUES=1"; do
         echo "== $v"
         for i in 1 2 3 4 5 6 7 8; do env $v HBK_BENCH_STAMPS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | cut -c1-420; done
       done) > $O/tail7.log 2>&1;;
    pftest)
      (python -m pytest tests/test_gpu_sharded.py -q -x -k "prefetch" 2>&1 | grep -E "passed|failed|rror" | tail -5
       python tools/sweep.py --cases e 2>&1 | grep "Sharded" | cut -c1-200) > $O/pftest.log 2>&1; cat $O/pftest.log;;
    dftest)
      python -m pytest tests/test_gpu_sharded.py -q -x -k "dense_features or prefetch" 2>&1 | grep -E "passed|failed|rror" | tail -5;;
    final2)   # late-round refresh: the bench lines, the headline kernel's stats, the sweep
      prof prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      cp $O/prof_bench.txt $O/r04_bench_kernel_stats.txt
      find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bench_rocprofv3_kernel_stats.csv
      timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/r04_bench_lines.jsonl
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" >> $O/r04_bench_lines.jsonl
      for w in fp32 fp16; do
        timeout 300 python bench.py --gpus 1 --sharded --wire $w --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" >> $O/r04_bench_lines.jsonl
      done
      timeout 2400 python tools/sweep.py --big --cases a,b,c,d,e,f,g,h,i,j,k,l 2>/dev/null | grep "^{" > $O/r04_sweep.jsonl
      wc -l $O/r04_bench_lines.jsonl $O/r04_sweep.jsonl; head -12 $O/r04_bench_kernel_stats.txt | cut -c1-160;;
    tail8)
      (for v in noevents evsync streamsync; do
         echo "== HBK_BENCH_PROBE=$v"
         for i in 1 2 3 4 5 6 7 8 9 10; do HBK_BENCH_PROBE=$v HBK_BENCH_STAMPS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | cut -c1-420; done
       done) > $O/tail8.log 2>&1;;
    tail9)
      (for i in 1 2 3 4 5 6 7 8 9 10; do HBK_BENCH_STAMPS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | cut -c1-420; done
       python bench.py --gpus 1 --sharded --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" | cut -c1-300) > $O/tail9.log 2>&1;;
    lines)   # the bench lines alone
      timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/r04_bench_lines.jsonl
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" >> $O/r04_bench_lines.jsonl
      for w in fp32 fp16; do
        timeout 300 python bench.py --gpus 1 --sharded --wire $w --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" >> $O/r04_bench_lines.jsonl
      done
      cut -c1-200 $O/r04_bench_lines.jsonl;;
    *) echo "unknown stage $st";;
  esac
done
