#!/bin/bash
# Round-6 GPU-box visits: stages picked on the command line, everything lands under gpurun_out/.
#   tools/gpu_r6.sh "test seed400"        (see the case labels)
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
for st in $STAGES; do
  case $st in
    test)       # the driver's command (with -x), then the tail
      timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/test.log 2>&1; echo "pytest rc=$?" >> $O/test.log; tail -25 $O/test.log;;
    repeats)    # the whole -m gpu suite 5 x in fresh processes WITHOUT -x
      : > $O/gputest_repeats.txt
      for i in 1 2 3 4 5; do
        timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/test_rep$i.log 2>&1; rc=$?
        echo "== run $i: rc=$rc" >> $O/gputest_repeats.txt; grep -E "passed|failed" $O/test_rep$i.log | tail -1 >> $O/gputest_repeats.txt
        grep -E "^(FAILED|ERROR)" $O/test_rep$i.log >> $O/gputest_repeats.txt
      done; cat $O/gputest_repeats.txt;;
    seed400)    # the r05 failure (dedup, seed 400) in 24 fresh processes, worst |diff| / sum|g| logged
      rm -f $O/seed400_ratios.txt
      for i in $(seq 1 24); do
        d=$(mktemp -d); HBK_TEST_RATIO_LOG=$O/seed400_ratios.txt timeout 300 python tests/support/multi_worker.py --rank 0 --world 1 --dir $d --cases dedup > $O/seed400_last.log 2>&1 || { echo "process $i FAILED"; tail -20 $O/seed400_last.log; }
        rm -rf $d
      done
      python - <<PY | tee $O/seed400_summary.txt
import re
rows=[l for l in open('$O/seed400_ratios.txt')]
ratio=lambda l: float(re.search(r'worst_ratio=([0-9.e+-]+)',l).group(1))
pids={l.split()[0] for l in rows}
f16=[l for l in rows if 'fp16' in l]; f32=[l for l in rows if 'fp16' not in l]
print(f'{len(rows)} checks in {len(pids)} fresh processes (tests/support/multi_worker.py --world 1 --cases dedup)')
print(f'fp32 checks: {len(f32)}, worst |diff| / sum|terms| = {max(map(ratio,f32)):.3e} (bound 1e-5 + 1e-6 floor)')
print(f'fp16-wire checks: {len(f16)}, worst = {max(map(ratio,f16)):.3e} (bound 2e-3)')
bd=[l for l in rows if 'backward' in l and 'Zipf' in l]
rb=sorted(map(ratio,bd))
print(f"seed-400 backward checks (the r05 failure, 'requester-side dedup, Zipf ids'): {len(bd)}, worst ratio {rb[-1]:.3e}, median {rb[len(rb)//2]:.3e}")
w=max(bd,key=ratio); print('worst line:', w.strip())
PY
      ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log;;
    stamps)     # where the host returns from the launches of the bench's steps (headline + the backward cases)
      HBK_BENCH_STAMPS=1 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > $O/stamps.log 2>&1; echo "rc=$?" >> $O/stamps.log; grep -E "bench stamps|^rc" $O/stamps.log | cut -c1-400;;
    benchsh)    # the sharded step at one rank through RCCL: forms, p2p, three plans pipelined, references
      timeout 900 python bench.py --sharded --steps 50 --warmup 10 --cpu-seconds 0 > $O/bench_sh.log 2>&1; echo "rc=$?" >> $O/bench_sh.log; tail -2 $O/bench_sh.log | cut -c1-3000;;
    det)        # the deterministic backward: parity, then its cost next to the default on the same box
      timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -x -q -m gpu -k "deterministic or launch_repeats or interleaved" --durations=5 > $O/det_test.log 2>&1; echo "pytest rc=$?" >> $O/det_test.log; tail -12 $O/det_test.log
      (for w in b s R Q d w; do for det in 0 1 2; do HBK_BWD_DETERMINISTIC=$det timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/deterministic=$det  /"; done; done) > $O/det_cost.txt 2>&1; cut -c1-200 $O/det_cost.txt;;
    detab)      # what the pieces of the deterministic row-sorted jobs cost (probe builds: tools/bin/variants/<name>/)
      (for rep in 1 2; do
         for w in ${WORK:-b R}; do
           HBK_BWD_DETERMINISTIC=0 timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/default   /"
           HBK_BWD_DETERMINISTIC=1 timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/det       /"
           for v in ${VARIANTS:-nosort atomic atomxcd all}; do
             HBK_BWD_DETERMINISTIC=1 LD_LIBRARY_PATH=$R/tools/bin/variants/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/$(printf %-10s $v)/"
           done
         done
       done) > $O/detab.txt 2>&1; cut -c1-150 $O/detab.txt;;
    detquick)   # the deterministic mode next to the default: config 2, ragged, the duplicate-heavy cases (C ABI), twice
      (for rep in 1 2; do for w in ${WORK:-b s R d}; do for det in 0 1; do HBK_BWD_DETERMINISTIC=$det timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/deterministic=$det  /"; done; done; done) > $O/detquick.txt 2>&1; cut -c1-150 $O/detquick.txt;;
    defvar)     # the DEFAULT backward of the current build next to probe builds (tools/bin/variants/<name>/), alternating in one visit
      (for rep in 1 2 3; do for w in ${WORK:-b s R}; do
         timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/current   /"
         for v in ${VARIANTS:-splitoc}; do LD_LIBRARY_PATH=$R/tools/bin/variants/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/$(printf %-10s $v)/"; done
       done; done) > $O/defvar.txt 2>&1; cut -c1-140 $O/defvar.txt;;
    optab)      # a library option A/B-ed per process, alternating: OPT=HBK_BWD_SIMPLE VALS="1 0" WORK="b s R"
      (for rep in 1 2 3; do for w in ${WORK:-b s R}; do for v in ${VALS:-1 0}; do
         env ${OPT:-HBK_BWD_SIMPLE}=$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/${OPT:-HBK_BWD_SIMPLE}=$v  /"
       done; done; done) > $O/optab.txt 2>&1; cut -c1-150 $O/optab.txt;;
    detvar)     # the deterministic mode of the current build next to probe builds of the library (tools/bin/variants/<name>/), alternating in one visit
      (for rep in 1 2 3; do for w in ${WORK:-b s R}; do
         HBK_BWD_DETERMINISTIC=1 timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/current   /"
         for v in ${VARIANTS:-prev}; do HBK_BWD_DETERMINISTIC=1 LD_LIBRARY_PATH=$R/tools/bin/variants/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/$(printf %-10s $v)/"; done
       done; done) > $O/detvar.txt 2>&1; cut -c1-140 $O/detvar.txt;;
    detskew)    # the deterministic modes under skew: dim 128, 26 x 65536 ids -- uniform / Zipf(1.2) / one row with 20 % / one row with all
      (for det in ${DETS:-0 1 2}; do HBK_BWD_DETERMINISTIC=$det timeout 600 python tools/sweep.py --cases f 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l)
  if 'case' in d: print('deterministic=$det  ', d['case'].ljust(40), d['us'])"; done) > $O/detskew.txt 2>&1; cat $O/detskew.txt;;
    detsharded) # the sharded step (W = 1 through RCCL) forward + backward under the deterministic modes: the owner's backward reads segmented inputs
      (for det in ${DETS:-0 1 2}; do HBK_BWD_DETERMINISTIC=$det timeout 600 python tools/sweep.py --cases g 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l)
  if 'case' in d and 'bwd' in d['case'] or 'step only' in d.get('case',''): print('deterministic=$det  ', d['case'].ljust(75), d['us'])"; done) > $O/detsharded.txt 2>&1; cat $O/detsharded.txt;;
    scaleab)    # ragged mean columns: the scaling inside the histogram launch (1) vs in front (0) (vs 2 = on a side stream: a probe build of round 6, removed)
      (for rep in 1 2; do for w in ${WORK:-R Q r}; do for v in 1 2 0; do HBK_BWD_SCALE_FUSED=$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/scale_fused=$v  /"; done; done; done) > $O/scaleab.txt 2>&1; cut -c1-150 $O/scaleab.txt;;
    detprof)    # kernel times of the deterministic backward (config 2 emit, ragged)
      export HBK_BENCH_ITERS=4
      for det in ${DETS:-0 1}; do
        HBK_BWD_DETERMINISTIC=$det prof prof_det${det}_b "" -- $R/tools/bin/bench_ops b
        HBK_BWD_DETERMINISTIC=$det prof prof_det${det}_R "" -- $R/tools/bin/bench_ops R
        echo "== deterministic=$det"; head -12 $O/prof_det${det}_b.txt | cut -c1-160; head -14 $O/prof_det${det}_R.txt | cut -c1-160; trim prof_det${det}_b; trim prof_det${det}_R
      done
      unset HBK_BENCH_ITERS;;
    p2pprof)    # kernel times of the sharded step at one rank in the p2p form
      prof prof_p2p_on "" -- python $R/bench.py --sharded --steps 30 --warmup 5 --cpu-seconds 0 --no-secondary --tune-steps 4 --p2p on
      grep -E "hbk|kernel  " $O/prof_p2p_on.txt | cut -c1-150 | head -24; trim prof_p2p_on;;
    fuzzhunt)   # the fuzz tests with fresh random draws, 6 x the committed example counts
      HBK_FUZZ_RANDOM=1 HBK_FUZZ_SCALE=6 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu --durations=6 > $O/fuzzhunt.log 2>&1; echo "pytest rc=$?" >> $O/fuzzhunt.log; tail -14 $O/fuzzhunt.log;;
    pairab)     # paired output stores of the row-sorted reduce: rows up to HBK_RS_PAIR_DIST positions apart (probe builds)
      (for rep in 1 2; do for v in ${VARIANTS:-base d2 d3 d3w10 d2w10}; do
         for w in R Q b; do LD_LIBRARY_PATH=$R/tools/bin/variants/$v timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd | sed "s/^/$v  /"; done
       done; done) > $O/pairab.txt 2>&1; cut -c1-150 $O/pairab.txt;;
    pairpmc)    # write / read requests of the ragged backward's reduce kernel per variant
      export HBK_BENCH_ITERS=2
      for v in ${VARIANTS:-base d3}; do
        LD_LIBRARY_PATH=$R/tools/bin/variants/$v prof pmc_pair_$v "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/bench_ops R
        echo "== $v"; tail -1 $O/pmc_pair_$v.log; pmc_table $O/pmc_pair_$v.json bwd_rowsort; trim pmc_pair_$v
      done
      unset HBK_BENCH_ITERS;;
    hosttime)   # host time of the backward entry per call (C ABI, no Python)
      (for w in b s R r; do timeout 300 tools/bin/bench_ops $w 2>&1 | grep group_lookup_bwd; done) > $O/hosttime.txt 2>&1; cut -c1-200 $O/hosttime.txt;;
    evidence)   # the round's evidence run: bench lines, kernel stats of the same command, traffic, C-ABI ops
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/ev_bench50.log 2>&1; grep "^{" $O/ev_bench50.log | tail -1 > $O/ev_bench_lines.jsonl
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/ev_bench20.log 2>&1; grep "^{" $O/ev_bench20.log | tail -1 >> $O/ev_bench_lines.jsonl
      timeout 600 python bench.py --sharded --steps 50 --warmup 10 --cpu-seconds 0 > $O/ev_bench_sh.log 2>&1; grep "^{" $O/ev_bench_sh.log | tail -1 >> $O/ev_bench_lines.jsonl
      cut -c1-300 $O/ev_bench_lines.jsonl
      prof ev_prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      head -16 $O/ev_prof_bench.txt | cut -c1-160
      timeout 900 python tools/hbm_traffic.py r06 $O/hbm_traffic.json > $O/ev_traffic.log 2>&1; tail -3 $O/ev_traffic.log; head -30 $O/hbm_traffic.json
      timeout 600 tools/bin/bench_ops > $O/ev_bench_ops.txt 2>&1; (for w in R r d; do timeout 300 tools/bin/bench_ops $w 2>&1 | grep -v "^hbk "; done) >> $O/ev_bench_ops.txt; cut -c1-190 $O/ev_bench_ops.txt
      trim ev_prof_bench;;
    cfg5prof)   # kernel times of the config-5 shape sweep (forward, backward + SGD, step only, + Adagrad x 3)
      prof prof_cfg5 "" -- python $R/tools/sweep.py --cases h
      head -40 $O/prof_cfg5.txt | cut -c1-170; trim prof_cfg5;;
    cfg5)       # config-5 shape x 3 processes (sweep h): forward, backward + SGD, step only, + Adagrad
      for i in 1 2 3; do timeout 600 python tools/sweep.py --cases h 2>/dev/null | grep "^{" > $O/ev_sweep_h_$i.jsonl; done
      for i in 1 2 3; do echo "== process $i"; python -c "
import sys,json
for l in open('$O/ev_sweep_h_$i.jsonl'):
  d=json.loads(l)
  if 'case' in d: print('  ',d['case'][:90].ljust(90), d['us'])"; done;;
    t_*)        # t_<file stem>[:<-k expression>]: one test file, e.g. t_test_gpu_sync or t_test_gpu_parity:rowsort
      spec=${st#t_}; f=${spec%%:*}; k=""; [ "$spec" != "$f" ] && k=${spec#*:}
      timeout 1500 python -m pytest tests/$f.py -x -q -m gpu ${k:+-k "$k"} --durations=5 > $O/$f.log 2>&1; echo "pytest rc=$?" >> $O/$f.log; tail -15 $O/$f.log;;
    *) echo "unknown stage $st";;
  esac
done
