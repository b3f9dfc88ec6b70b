#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, tuning probe, rocprofv3 kernel stats.
# Everything lands under gpurun_out/ (merged back by gpurun).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
STAGES=${1:-"test smoke bench tune prof"}
for st in $STAGES; do
  case $st in
    test)
      timeout 900 python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; echo "pytest rc=$?" >> $O/test.log; tail -5 $O/test.log;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log
      timeout 300 python bench.py --steps 50 --warmup 10 --batch 262144 --cpu-seconds 0 > $O/bench_b262144.log 2>&1; tail -2 $O/bench_b262144.log;;
    ops)
      timeout 120 tools/bin/bench_ops > $O/bench_ops.log 2>&1; echo "ops rc=$?" >> $O/bench_ops.log; cat $O/bench_ops.log;;
    tune)
      timeout 600 tools/bin/tune_lookup > $O/tune.log 2>&1; echo "tune rc=$?" >> $O/tune.log; cat $O/tune.log;;
    prof)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 > $O/prof.log 2>&1; echo "prof rc=$?" >> $O/prof.log)
      tail -3 $O/prof.log; find $O/prof -name "*stats*" | head;;
    sweep)
      timeout 900 python tools/sweep.py --big --cases ${SWEEP_CASES:-a,b,c,d,e,f,g,h,i} > $O/sweep.log 2>&1; echo "sweep rc=$?" >> $O/sweep.log; cat $O/sweep.log | cut -c1-400;;
    profsweep)
      (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sweep -o sweep -- python $R/tools/sweep.py --cases ${SWEEP_CASES:-c,e} > $O/prof_sweep.log 2>&1; echo "rc=$?" >> $O/prof_sweep.log)
      tail -3 $O/prof_sweep.log;;
    pmc)
      # counters in their own passes (never combined with sys/hip/hsa traces)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$ctr -o bench -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/pmc_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_$ctr.log)
        (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_tune_$ctr -o tune -- $R/tools/bin/tune_lookup --quick > $O/pmc_tune_$ctr.log 2>&1; echo "rc=$?" >> $O/pmc_tune_$ctr.log)
        tail -2 $O/pmc_$ctr.log $O/pmc_tune_$ctr.log
      done
      find $O -name "*counter_collection*" | head;;
  esac
done
