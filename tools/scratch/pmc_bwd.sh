#!/bin/bash
# SQ counters of the backward kernels, one --pmc pass per group (never combined with traces other than kernel-trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_bwd; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- $R/tools/bin/bench_ops ${OPS_MODE:-bwd} > $O/g$i.log 2>&1
done
python - <<PY
import csv,glob,collections,re
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(bwd_\w+|partition_\w+|unique_\w+|fillBuffer\w*)(<[^>]*>)?', r["Kernel_Name"])
        d[m.group(0) if m else r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} {sum(vals)/len(vals):16.0f}")
PY
