#!/bin/bash
# Round-5 GPU-box visits: stages picked on the command line, everything lands under gpurun_out/.
#   tools/gpu_r5.sh "fetch place"        (see the case labels)
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
    (cd /tmp && export TMPDIR=/tmp && timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log)
    python tools/prof_summary.py pmc $O/$name > $O/$name.json 2>> $O/$name.log
  else
    (cd /tmp && export TMPDIR=/tmp && timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1; echo "rc=$?" >> $O/$name.log)
    python tools/prof_summary.py stats $O/$name > $O/$name.txt 2>> $O/$name.log
  fi
}
trim() { find $O/$1 -name "*.csv" -size +2M -delete; }
pmc_table() {   # pmc_table <json> <kernel-name filter>: one line per kernel, counters side by side
  python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  if sys.argv[2] in k: print(k[:44].ljust(44), {c.replace('TCC_EA0_','').replace('_sum',''):round(x['mean']) for c,x in v.items()})
PY
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
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log;;
    counters)   # which TLB / translation counters this rocprofv3 knows
      (cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L 2>&1 | grep -o -i -E "[A-Z0-9_]*(UTCL|TLB|XNACK|TRANSLATION)[A-Za-z0-9_]*" | sort -u) > $O/counters_tlb.txt 2>&1
      wc -l $O/counters_tlb.txt; head -60 $O/counters_tlb.txt;;
    fetch)      # request size of a 64-byte row fetch per load path / memory kind
      timeout 90 tools/bin/fetch_probe > $O/fetch_times.txt 2>&1; cat $O/fetch_times.txt
      prof pmc_fetch "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $R/tools/bin/fetch_probe --quick
      tail -1 $O/pmc_fetch.log; pmc_table $O/pmc_fetch.json fetch_; trim pmc_fetch;;
    place)      # the same launch on tables allocated under different policies
      timeout 900 tools/bin/placement_probe > $O/place_times.txt 2>&1; cat $O/place_times.txt
      timeout 900 tools/bin/placement_probe > $O/place_times2.txt 2>&1; cat $O/place_times2.txt;;
    placepmc)
      prof pmc_place1 "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" -- $R/tools/bin/placement_probe --quick
      tail -1 $O/pmc_place1.log
      python tools/prof_summary.py chunks $O/pmc_place1 group_lookup_fwd 4 malloc,slab,vmm_1g,malloc_rev,frag,vmm_2m,malloc,slab,vmm_1g,malloc,slab,vmm_1g,malloc_rev,frag,vmm_2m,malloc,slab,vmm_1g | tee $O/place_counters.txt; trim pmc_place1;;
    bwdbase)    # where the backward family stands on this box before round 5's changes
      (for w in b s R r d w; do timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk "; done) > $O/bwdbase.log 2>&1; cut -c1-200 $O/bwdbase.log;;
    bwdab)      # round 5 levers (a) packed pair words, (b) segments found inside the grouping kernels, (c) scaling in the histogram launch
      (for cfg in "0 0 0" "1 1 0" "1 1 1" "0 0 0" "1 1 1"; do set -- $cfg
         for w in R r; do
           HBK_BWD_PAIRS_PACKED=$1 HBK_BWD_SEG_INLINE=$2 HBK_BWD_SCALE_FUSED=$3 timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/packed=$1 seg_inline=$2 scale_fused=$3  /"
         done
       done) > $O/bwdab.log 2>&1; cut -c1-200 $O/bwdab.log;;
    rsprof)     # kernel times + memory-side counters of the ragged backward (bench_ops R) after round 5's levers
      export HBK_BENCH_ITERS=6
      prof prof_ragged "" -- $R/tools/bin/bench_ops R
      grep -E "bwd_|kernel  " $O/prof_ragged.txt | cut -c1-150 | head -14
      prof pmc_rs_tcc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" -- $R/tools/bin/bench_ops R
      prof pmc_rs_sq "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" -- $R/tools/bin/bench_ops R
      unset HBK_BENCH_ITERS
      for f in pmc_rs_tcc pmc_rs_sq; do echo "== $f"; tail -1 $O/$f.log; pmc_table $O/$f.json bwd_; trim $f; done
      trim prof_ragged;;
    evidence)   # the round's evidence run: bench lines, kernel stats, traffic, C-ABI ops, config-5 shape x 3 processes
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/ev_bench50.log 2>&1; grep "^{" $O/ev_bench50.log | tail -1 > $O/ev_bench_lines.jsonl
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/ev_bench20.log 2>&1; grep "^{" $O/ev_bench20.log | tail -1 >> $O/ev_bench_lines.jsonl
      timeout 600 python bench.py --sharded --steps 50 --warmup 10 --cpu-seconds 0 > $O/ev_bench_sh.log 2>&1; grep "^{" $O/ev_bench_sh.log | tail -1 >> $O/ev_bench_lines.jsonl
      cut -c1-400 $O/ev_bench_lines.jsonl
      prof ev_prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      grep -E "group_lookup|kernel  " $O/ev_prof_bench.txt | cut -c1-150 | head -5
      timeout 900 python tools/hbm_traffic.py r05 $O/hbm_traffic.json > $O/ev_traffic.log 2>&1; tail -3 $O/ev_traffic.log; cat $O/hbm_traffic.json | head -30
      timeout 600 tools/bin/bench_ops > $O/ev_bench_ops.txt 2>&1; (for w in R r d; do timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk "; done) >> $O/ev_bench_ops.txt; cut -c1-170 $O/ev_bench_ops.txt
      for i in 1 2 3; do timeout 600 python tools/sweep.py --cases h 2>/dev/null | grep "^{" > $O/ev_sweep_h_$i.jsonl; done
      HBK_BWD_PAIRS_PACKED=0 HBK_BWD_SEG_INLINE=0 HBK_BWD_SCALE_FUSED=0 timeout 600 python tools/sweep.py --cases h 2>/dev/null | grep "^{" > $O/ev_sweep_h_old.jsonl
      for f in $O/ev_sweep_h_1.jsonl $O/ev_sweep_h_2.jsonl $O/ev_sweep_h_3.jsonl $O/ev_sweep_h_old.jsonl; do echo "== $f"; python -c "
import sys,json
for l in open('$f'):
  d=json.loads(l); print('  ',d['case'][:70].ljust(70), d['us'])"; done
      timeout 900 python tools/sweep.py --big --cases b,c,d 2>/dev/null | grep "^{" > $O/ev_sweep_bcd.jsonl; python -c "
import json
for l in open('$O/ev_sweep_bcd.jsonl'):
  d=json.loads(l)
  if 'case' in d: print('  ',d['case'][:90].ljust(90), d['us'])";;
    evidence2)  # the backward part of the evidence again (after the whole-line stores): C-ABI ops, sweeps b / c / h
      timeout 600 tools/bin/bench_ops > $O/ev_bench_ops.txt 2>&1; (for w in R r d; do timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk "; done) >> $O/ev_bench_ops.txt; grep group_lookup_bwd $O/ev_bench_ops.txt | cut -c1-170
      for i in 1 2 3; do timeout 600 python tools/sweep.py --cases h 2>/dev/null | grep "^{" > $O/ev_sweep_h_$i.jsonl; done
      timeout 900 python tools/sweep.py --big --cases b,c,d 2>/dev/null | grep "^{" > $O/ev_sweep_bcd.jsonl
      for f in $O/ev_sweep_h_1.jsonl $O/ev_sweep_h_2.jsonl $O/ev_sweep_h_3.jsonl $O/ev_sweep_bcd.jsonl; do echo "== $f"; python -c "
import sys,json
for l in open('$f'):
  d=json.loads(l)
  if 'case' in d: print('  ',d['case'][:90].ljust(90), d['us'])"; done;;
    rsstats)    # kernel times of the ragged backward (final build)
      export HBK_BENCH_ITERS=6
      prof prof_ragged "" -- $R/tools/bin/bench_ops R
      unset HBK_BENCH_ITERS
      grep -E "bwd_|kernel  " $O/prof_ragged.txt | cut -c1-150 | head -14; trim prof_ragged;;
    c5stats)    # kernel times of the config-5 shape (forward, backward + SGD, step only, + Adagrad)
      prof prof_cfg5 "" -- python $R/tools/sweep.py --big --cases h
      grep -E "^\{" $O/prof_cfg5.log | cut -c1-200; head -40 $O/prof_cfg5.txt | cut -c1-170; trim prof_cfg5;;
    rscounters) # memory-side counters of the ragged backward's kernels (two short passes)
      export HBK_BENCH_ITERS=2
      prof pmc_rs_tcc_a "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops R
      prof pmc_rs_tcc_b "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- $R/tools/bin/bench_ops R
      unset HBK_BENCH_ITERS
      for f in pmc_rs_tcc_a pmc_rs_tcc_b; do echo "== $f"; tail -1 $O/$f.log; pmc_table $O/$f.json bwd_; trim $f; done;;
    p2pprof)    # kernel times of the sharded step at one rank: exchange form (inline) and p2p form
      prof prof_p2p "" -- python $R/bench.py --sharded --steps 30 --warmup 5 --cpu-seconds 0 --no-secondary --tune-steps 0 --p2p off
      grep -E "hbk|kernel  " $O/prof_p2p.txt | cut -c1-150 | head -16
      HBK_BENCH_FORCE_P2P=1 true
      prof prof_p2p_on "" -- python $R/bench.py --sharded --steps 30 --warmup 5 --cpu-seconds 0 --no-secondary --tune-steps 4 --p2p on
      grep -E "hbk|kernel  " $O/prof_p2p_on.txt | cut -c1-150 | head -24; trim prof_p2p; trim prof_p2p_on;;
    overlap)    # the step's forms under an artificial wire (one rank, full per-rank work)
      timeout 600 python tools/overlap_model.py > $O/overlap_model.txt 2> $O/overlap_model.err; echo "rc=$?"; cat $O/overlap_model.txt; tail -5 $O/overlap_model.err;;
    bwdtest)
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "backward or random or graph" --durations=8 > $O/bwdtest.log 2>&1; echo "pytest rc=$?" >> $O/bwdtest.log; tail -25 $O/bwdtest.log;;
    t_*)        # t_<file stem>[:<-k expression>]: one test file, e.g. t_test_gpu_sync or t_test_gpu_parity:rowsort
      spec=${st#t_}; f=${spec%%:*}; k=""; [ "$spec" != "$f" ] && k=${spec#*:}
      timeout 1500 python -m pytest tests/$f.py -x -q -m gpu ${k:+-k "$k"} --durations=5 > $O/$f.log 2>&1; echo "pytest rc=$?" >> $O/$f.log; tail -15 $O/$f.log;;
    *) echo "unknown stage $st";;
  esac
done
