"""Per-kernel durations from a rocprofv3 kernel-trace CSV: python tools/trace_summary.py <kernel_trace.csv> [filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.defaultdict(list)
for r in rows:
  d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
  if flt in k:
    name = k.replace('(anonymous namespace)::', '').replace('hbk::', '')[:90]
    print(f'{name:92s} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100*sum(v)/tot:6.1f}')
