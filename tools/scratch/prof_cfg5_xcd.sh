cd /tmp && export TMPDIR=/tmp
for x in 1 4; do
  HBK_BWD_XCD=$x rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$x -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py --cases h > /dev/null 2>&1
  echo "== bwd_xcd=$x"; python $GRAFT_REPO_ROOT/tools/prof_summary.py stats $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$x | grep "bwd_" | cut -c1-150 | head -14
done
