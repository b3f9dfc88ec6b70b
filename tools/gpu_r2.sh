#!/bin/bash
# Round-2 GPU-box visits: stages picked on the command line, everything lands under gpurun_out/.
#   tools/gpu_r2.sh "test ceil"        (see the case labels)
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
  # keep the merge small: the condensed summaries are what we read
  find $O/$name -name "*.csv" -size +2M -delete
}
for st in $STAGES; do
  case $st in
    test)
      timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/test.log 2>&1; echo "pytest rc=$?" >> $O/test.log; tail -25 $O/test.log;;
    testnew)
      timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_configs.py -q -m gpu --durations=10 > $O/testnew.log 2>&1; echo "pytest rc=$?" >> $O/testnew.log; tail -40 $O/testnew.log;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -3 $O/bench.log;;
    benchn)   # the self-launching path on whatever GPUs the box has (1): must fail fast and say why
      timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_n2.log 2>&1; echo "bench rc=$?" >> $O/bench_n2.log; tail -3 $O/bench_n2.log
      timeout 300 python bench.py --gpus 1 --sharded --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_sharded1.log 2>&1; echo "bench rc=$?" >> $O/bench_sharded1.log; tail -3 $O/bench_sharded1.log;;
    ops)
      timeout 300 tools/bin/bench_ops > $O/bench_ops.log 2>&1; echo "ops rc=$?" >> $O/bench_ops.log; cat $O/bench_ops.log;;
    ceilpmc)
      prof pmc_ceil_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $R/tools/bin/ceiling_probe --quick
      prof pmc_ceil_hit "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/ceiling_probe --quick
      for f in pmc_ceil_rd pmc_ceil_hit; do echo "== $f"; tail -2 $O/$f.log; python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d.items()):
  print(k[:60].ljust(60), {c:round(x['mean']) for c,x in v.items()})
PY
      done;;
    ceil)
      timeout 600 tools/bin/ceiling_probe > $O/ceiling.log 2>&1; echo "rc=$?" >> $O/ceiling.log; cat $O/ceiling.log
      prof pmc_ceil_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- $R/tools/bin/ceiling_probe --quick
      prof pmc_ceil_hit "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- $R/tools/bin/ceiling_probe --quick
      prof pmc_ceil_req "TCC_REQ_sum TCC_READ_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum" -- $R/tools/bin/ceiling_probe --quick
      prof pmc_bench_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0
      prof pmc_bench_hit "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0
      prof pmc_bench_fetch "FETCH_SIZE" -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0
      prof pmc_bench_write "WRITE_SIZE" -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0
      for f in pmc_ceil_rd pmc_ceil_hit pmc_ceil_req pmc_bench_rd pmc_bench_hit pmc_bench_fetch pmc_bench_write; do echo "== $f"; tail -2 $O/$f.log; head -c 1500 $O/$f.json; done;;
    profbench)
      prof prof_bench "" -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0
      cat $O/prof_bench.txt;;
    profsharded)
      prof prof_sharded "" -- python $R/bench.py --gpus 1 --sharded --steps 30 --warmup 5 --cpu-seconds 0 ${SHARDED_ARGS:-}
      cat $O/prof_sharded.txt
      HBK_SHARDED_TRACE=1 timeout 120 python bench.py --gpus 1 --sharded --steps 8 --warmup 2 --cpu-seconds 0 ${SHARDED_ARGS:-} 2>&1 | tail -6;;
    profops)
      prof prof_ops "" -- $R/tools/bin/bench_ops ${OPS_ARGS:-}
      cat $O/prof_ops.txt;;
    sweep)
      timeout 1200 python tools/sweep.py --big --cases ${SWEEP_CASES:-a,b,c,d,e,f,g,h,i} > $O/sweep.log 2>&1; echo "sweep rc=$?" >> $O/sweep.log; cut -c1-400 $O/sweep.log;;
    profsweep)
      prof prof_sweep "" -- python $R/tools/sweep.py --cases ${SWEEP_CASES:-c,e}
      cat $O/prof_sweep.txt;;
    *) echo "unknown stage $st";;
  esac
done
