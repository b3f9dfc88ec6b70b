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
    *) echo "unknown stage $st";;
  esac
done
