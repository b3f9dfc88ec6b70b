#!/bin/bash
# Round-3 GPU-box visits: stages picked on the command line, everything lands under gpurun_out/.
#   tools/gpu_r3.sh "bwdtest bwdops"        (see the case labels)
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
for st in $STAGES; do
  case $st in
    test)
      timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/test.log 2>&1; echo "pytest rc=$?" >> $O/test.log; tail -25 $O/test.log;;
    bwdtest)
      timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -k "backward or bwd or fuzz" --durations=10 > $O/bwdtest.log 2>&1; echo "pytest rc=$?" >> $O/bwdtest.log; tail -40 $O/bwdtest.log;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log;;
    ops)
      timeout 300 tools/bin/bench_ops > $O/bench_ops.log 2>&1; echo "ops rc=$?" >> $O/bench_ops.log; cat $O/bench_ops.log;;
    bwdops)   # the backward through the C ABI: dense (row-range) buckets vs hashed ones
      for dense in 1 0; do
        for w in ${BWD_CASES:-b s d r w}; do
          HBK_BWD_DENSE=$dense timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk " | sed "s/^/dense=$dense  /"
        done
      done > $O/bwdops.log 2>&1; cat $O/bwdops.log;;
    bwdstamps)
      for w in b s d; do LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops $w; done > $O/bwdstamps.log 2>&1; cat $O/bwdstamps.log;;
    profbwd)
      prof prof_bwd "" -- $R/tools/bin/bench_ops b
      prof prof_bwd_step "" -- $R/tools/bin/bench_ops s
      cat $O/prof_bwd.txt $O/prof_bwd_step.txt;;
    pmcbwd)
      prof pmc_bwd_sq1 "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" -- $R/tools/bin/bench_ops b
      prof pmc_bwd_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- $R/tools/bin/bench_ops b
      prof pmc_bwd_tcc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops b
      for f in pmc_bwd_sq1 pmc_bwd_sq2 pmc_bwd_tcc; do echo "== $f"; tail -2 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  print(k[:70].ljust(70), {c:round(x['mean']) for c,x in v.items()})
PY
      done;;
    variants)   # probe builds of the dense backward (tools/scratch/bwd_dense_variants.sh)
      for v in tools/bin/v_*; do
        for w in ${VARIANT_CASES:-b s}; do
          LD_LIBRARY_PATH=$R/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk " | sed "s|^|$(basename $v)  |"
        done
      done > $O/variants.log 2>&1; cut -c1-400 $O/variants.log;;
    pmcvariants)   # SQ counters of the probe builds
      for v in tools/bin/v_*; do
        n=$(basename $v)
        LD_LIBRARY_PATH=$R/$v prof pmc_${n}_1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" -- $R/tools/bin/bench_ops b
        LD_LIBRARY_PATH=$R/$v prof pmc_${n}_2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_INST_CYCLES_VMEM" -- $R/tools/bin/bench_ops b
        LD_LIBRARY_PATH=$R/$v prof pmc_${n}_3 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" -- $R/tools/bin/bench_ops b
        for k in 1 2 3; do f=pmc_${n}_$k; echo "== $f"; tail -1 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  if 'dense_kernel' in k or 'reduce_kernel' in k: print(k[:60].ljust(60), {c:round(x['mean']) for c,x in v.items()})
PY
        done
      done;;
    shardtest)
      timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_configs.py tests/test_gpu_golden.py -q -m gpu -k "not config4" --durations=8 > $O/shardtest.log 2>&1; echo "pytest rc=$?" >> $O/shardtest.log; tail -25 $O/shardtest.log;;
    shardbench)   # one rank, through RCCL: fp32 wire, fp16 wire fused / with the two cast passes; the three forms
      for w in fp32 fp16; do
        for fused in 1 0; do
          [ $w = fp32 ] && [ $fused = 0 ] && continue
          HBK_SHARDED_WIRE_FUSED=$fused timeout 300 python bench.py --gpus 1 --sharded --wire $w --steps 30 --warmup 5 --cpu-seconds 0 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('wire=$w fused=$fused', 'ms_per_step', d['ms_per_step'], 'value', d['value'], d['config'].get('sharded_form'), d['config'].get('sharded_form_probe_ms_per_step'))"
        done
      done > $O/shardbench.log 2>&1; cat $O/shardbench.log;;
    synctest)
      timeout 900 python -m pytest tests/test_gpu_sync.py -q -m gpu --durations=5 > $O/synctest.log 2>&1; echo "pytest rc=$?" >> $O/synctest.log; tail -30 $O/synctest.log;;
    hottest)
      timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "hot_row or group_lookup" --durations=5 > $O/hottest.log 2>&1; echo "pytest rc=$?" >> $O/hottest.log; tail -15 $O/hottest.log;;
    hotsweep)
      timeout 900 python tools/sweep.py --big --cases j > $O/hotsweep.log 2>&1; echo "rc=$?" >> $O/hotsweep.log; cut -c1-300 $O/hotsweep.log;;
    profhot)
      prof prof_hot "" -- python $R/tools/sweep.py --big --cases j
      cat $O/prof_hot.txt
      export SWEEP_J_KINDS="Zipf(1.2)"
      prof pmc_hot_tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" -- python $R/tools/sweep.py --big --cases j
      prof pmc_hot_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" -- python $R/tools/sweep.py --big --cases j
      unset SWEEP_J_KINDS
      for f in pmc_hot_tcc pmc_hot_tcp; do echo "== $f"; tail -1 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  if 'group_lookup_fwd' in k: print(k[:70].ljust(70), {c:(round(x['mean']), x.get('n')) for c,x in v.items()})
PY
      done;;
    profsweep)   # kernel stats of sweep cases, dense (row-range) buckets on / off
      for dense in 1 0; do
        export HBK_BWD_DENSE=$dense
        prof prof_sweep_dense$dense "" -- python $R/tools/sweep.py --big --cases ${SWEEP_CASES:-d,h}
        grep "^{" $O/prof_sweep_dense$dense.log | cut -c1-200
        grep -E "bwd_|kernel  " $O/prof_sweep_dense$dense.txt | head -24
      done
      unset HBK_BWD_DENSE;;
    final)   # the round's evidence, copied to profiles/r03_* afterwards
      prof prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      cp $O/prof_bench.txt $O/r03_bench_kernel_stats.txt
      find $O/prof_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_bench_rocprofv3_kernel_stats.csv
      timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/r03_bench_lines.jsonl
      for w in fp32 fp16; do
        timeout 300 python bench.py --gpus 1 --sharded --wire $w --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" >> $O/r03_bench_lines.jsonl
      done
      HBK_SHARDED_WIRE_FUSED=0 timeout 300 python bench.py --gpus 1 --sharded --wire fp16 --steps 30 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" | sed 's/^{/{"unfused_fp16_wire": true, /' >> $O/r03_bench_lines.jsonl
      timeout 300 tools/bin/bench_ops > $O/r03_bench_ops.txt 2>&1
      for dense in 1 0; do for w in d r; do HBK_BWD_DENSE=$dense timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk " | sed "s/^/bwd_dense=$dense  /"; done; done >> $O/r03_bench_ops.txt
      prof prof_bwd "" -- $R/tools/bin/bench_ops b
      prof prof_bwd_step "" -- $R/tools/bin/bench_ops s
      HBK_BWD_DENSE=0 prof prof_bwd_hash "" -- $R/tools/bin/bench_ops b
      (echo "== config-2 backward, C ABI (tools/bin/bench_ops b): dense (row-range) buckets"; cat $O/prof_bwd.txt; echo; echo "== + fused SGD step / step only (bench_ops s)"; cat $O/prof_bwd_step.txt; echo; echo "== the same backward with hashed buckets (HBK_BWD_DENSE=0)"; cat $O/prof_bwd_hash.txt) > $O/r03_bwd_kernel_stats.txt
      prof pmc_bwd_sq1 "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" -- $R/tools/bin/bench_ops b
      prof pmc_bwd_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- $R/tools/bin/bench_ops b
      prof pmc_bwd_tcc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops b
      python - $O > $O/r03_bwd_sq_counters.txt <<'PY'
import json,sys
O=sys.argv[1]
print("SQ / TCC counters of the config-2 backward's kernels (tools/bin/bench_ops b; mean per launch; one --pmc pass per group)")
for f in ('pmc_bwd_sq1','pmc_bwd_sq2','pmc_bwd_tcc'):
  d=json.load(open(f'{O}/{f}.json'))
  for k,v in sorted(d.items()):
    if 'bwd_' in k: print(k[:64].ljust(64), {c:round(x['mean']) for c,x in v.items()})
PY
      LD_LIBRARY_PATH=$R/tools/bin/stamps timeout 300 tools/bin/bench_ops b > $O/r03_bwd_reduce_trace.txt 2>&1
      prof prof_cfg5 "" -- python $R/tools/sweep.py --cases h
      cp $O/prof_cfg5.txt $O/r03_cfg5_bwd_kernel_stats.txt
      prof prof_sharded "" -- python $R/bench.py --gpus 1 --sharded --steps 30 --warmup 5 --cpu-seconds 0 --tune-steps 0
      cp $O/prof_sharded.txt $O/r03_sharded_w1_kernel_stats.txt
      prof prof_unique "" -- $R/tools/bin/bench_ops u
      cp $O/prof_unique.txt $O/r03_unique_kernel_stats.txt
      timeout 1500 python tools/sweep.py --big --cases a,b,c,d,e,f,g,h,i,j 2>/dev/null | grep "^{" > $O/r03_sweep.jsonl
      prof prof_hot "" -- python $R/tools/sweep.py --big --cases j
      export SWEEP_J_KINDS="Zipf(1.2)"
      prof pmc_hot_tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" -- python $R/tools/sweep.py --big --cases j
      prof pmc_hot_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" -- python $R/tools/sweep.py --big --cases j
      unset SWEEP_J_KINDS
      (echo "== config-4 forward, fwd_hot_rows = 0 / 1 / 2 x Zipf(1.2) / uniform / one row (tools/sweep.py --big --cases j): kernel durations"; grep "group_lookup_fwd\|kernel  " $O/prof_hot.txt; echo; echo "== L2 / L1 counters, Zipf(1.2) ids only, mean per launch (the hot kernel's mean covers modes 1 and 2)"; python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ('pmc_hot_tcc','pmc_hot_tcp'):
  d=json.load(open(f'{O}/{f}.json'))
  for k,v in sorted(d.items()):
    if 'group_lookup_fwd' in k: print(k[:70].ljust(70), {c:round(x['mean']) for c,x in v.items()})
PY
      ) > $O/r03_hot_rows.txt
      ls -la $O/r03_*;;
    abfinal)   # policies toggled inside one process on the same tensors (tools/sweep.py SWEEP_AB) + the cache-policy probe builds
      (echo "Library options toggled INSIDE one process on the same tensors (tools/scratch/ab_options.sh: tools/sweep.py with SWEEP_AB=option:values);"
       echo "every line: case, the option values in the order they were run, microseconds per call under each."
       timeout 1500 tools/scratch/ab_options.sh "bwd_xcd:0,1,0,1 b,c,d,h" "bwd_xcd:3,1,3,1 b" "bwd_dense:0,1,0,1 c" "bwd_wide:0,1,2,0,1,2 d,h" "bwd_onepass:0,1,0,1 c,h" "fwd_xcd:0,2,0,2 a,b,h"
       echo "== fwd_xcd x fwd_hot_rows, config 4 (SWEEP_J_XCD=0,2,0,2)"
       SWEEP_J_XCD=0,2,0,2 timeout 600 python tools/sweep.py --big --cases j 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print(d['case'][9:].ljust(70), d['us'])") > $O/r03_inprocess_ab.txt 2>&1
      (echo "Probe builds of the backward (tools/scratch/bwd_cache_variants.sh), C ABI, two passes: shipped (plain gradient loads) vs the non-temporal loads of rounds 1-2"
       for rep in 1 2; do for v in tools/bin/v_*; do for w in b s r; do LD_LIBRARY_PATH=$R/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s|^|$(basename $v)  |"; done; done; done) > $O/r03_bwd_cache_variants.txt 2>&1
      ls -la $O/r03_inprocess_ab.txt $O/r03_bwd_cache_variants.txt;;
    sweep)
      timeout 1200 python tools/sweep.py --big --cases ${SWEEP_CASES:-a,b,c,d,e,f,g,h,i} > $O/sweep.log 2>&1; echo "sweep rc=$?" >> $O/sweep.log; cut -c1-400 $O/sweep.log;;
    *) echo "unknown stage $st";;
  esac
done
