"""Condense rocprofv3 output (csv) under gpurun_out/ into the small text/json summaries that
are committed under profiles/.

  python tools/prof_summary.py stats  gpurun_out/prof            -> per-kernel time table
  python tools/prof_summary.py pmc    gpurun_out/pmc_FETCH_SIZE  -> per-kernel counter means
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
  name = name.replace('hbk::(anonymous namespace)::', 'hbk::')
  name = name.split('(')[0] if name.startswith('hbk::') else name
  return name if len(name) <= 90 else name[:87] + '...'


def stats(d):
  files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
  agg = defaultdict(list)
  for f in files:
    for row in csv.DictReader(open(f)):
      dur = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
      agg[short(row['Kernel_Name'])].append(dur)
  total = sum(sum(v) for v in agg.values())
  print(f'{"kernel":<92}{"calls":>7}{"total_us":>12}{"avg_us":>10}{"min_us":>10}{"max_us":>10}{"pct":>7}')
  for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:<92}{len(v):>7}{sum(v):>12.1f}{sum(v)/len(v):>10.2f}{min(v):>10.2f}{max(v):>10.2f}'
          f'{100*sum(v)/total:>7.1f}')


def pmc(d):
  files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
  agg = defaultdict(lambda: defaultdict(list))
  for f in files:
    for row in csv.DictReader(open(f)):
      agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
  out = {}
  for k, ctrs in agg.items():
    out[k] = {c: {'mean': sum(v) / len(v), 'n': len(v)} for c, v in ctrs.items()}
  print(json.dumps(out, indent=1, sort_keys=True))


def chunks(d, pattern, per_chunk, labels=''):
  """Per-dispatch counter rows of the kernels whose name contains `pattern`, in dispatch order, cut
  into runs of `per_chunk` launches (one run per policy of tools/placement_probe): mean per run."""
  files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
  rows = defaultdict(dict)
  for f in files:
    for row in csv.DictReader(open(f)):
      if pattern in row['Kernel_Name']:
        rows[int(row['Dispatch_Id'])][row['Counter_Name']] = float(row['Counter_Value'])
  order = sorted(rows)
  per_chunk = int(per_chunk)
  names = labels.split(',') if labels else []
  for i in range(0, len(order), per_chunk):
    run = [rows[k] for k in order[i:i + per_chunk]]
    ctrs = sorted({c for r in run for c in r})
    label = names[i // per_chunk] if i // per_chunk < len(names) else f'run {i // per_chunk}'
    print(f'{label:<14}' + '  '.join(
        f'{c}={sum(r.get(c, 0.0) for r in run) / len(run):.0f}' for c in ctrs))


if __name__ == '__main__':
  {'stats': stats, 'pmc': pmc, 'chunks': chunks}[sys.argv[1]](*sys.argv[2:])
