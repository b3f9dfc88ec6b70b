"""Regenerates profiles/hbm_traffic.json (what bench.py prints as roofline.traffic) from two
rocprofv3 --pmc passes over bench.py on the GPU box -- run once per round:

  python tools/hbm_traffic.py r04 gpurun_out/hbm_traffic.json     (GPU box; tools/gpu_r4.sh traffic)
  cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json         (here, then commit)

HBM bytes per launch of the dominant kernel = read requests by size (TCC_EA0_RDREQ_128B x 128 +
_64B x 64 + _32B x 32) + TCC_EA0_WRREQ_64B x 64 (MI355X_MICROARCH.md: request counters of the L2's
memory side; separate passes, kernel-trace only)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = (('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_128B_sum'),
          ('TCC_EA0_WRREQ_sum', 'TCC_EA0_WRREQ_64B_sum', 'TCC_HIT_sum', 'TCC_MISS_sum'))
BENCH = ['python', os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--cpu-seconds', '0',
         '--no-secondary']   # (the forward kernel is what is counted: the backward cases behind it are skipped)


def main():
  rnd, out_path = sys.argv[1], sys.argv[2]
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  import prof_summary  # noqa: E402  pylint: disable=import-outside-toplevel
  means = {}
  kernel = None
  for k, ctrs in enumerate(PASSES):
    d = os.path.join(ROOT, 'gpurun_out', f'pmc_traffic_{k}')
    subprocess.run(['rm', '-rf', d])
    env = dict(os.environ, TMPDIR='/tmp')
    subprocess.run(['rocprofv3', '--kernel-trace', '--pmc', *ctrs, '--output-format', 'csv', '-d', d,
                    '-o', 'p', '--', *BENCH], cwd='/tmp', env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    import csv
    from collections import defaultdict
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
      for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, c in agg.items():
      if 'group_lookup_fwd_kernel' in name:
        kernel = name
        for ctr, v in c.items():
          means[ctr] = sum(v) / len(v)
  rd = means['TCC_EA0_RDREQ_128B_sum'] * 128 + means['TCC_EA0_RDREQ_64B_sum'] * 64 + \
      means['TCC_EA0_RDREQ_32B_sum'] * 32
  wr = means['TCC_EA0_WRREQ_64B_sum'] * 64
  entry = {
    'kernel': kernel, 'hbm_bytes_per_launch': int(rd + wr), 'measured_in': rnd,
    'read_bytes': int(rd), 'write_bytes': int(wr),
    'source': 'tools/hbm_traffic.py: rocprofv3 --kernel-trace --pmc ' + ' / --pmc '.join(
      ' '.join(p) for p in PASSES) + ' (separate passes) -- ' + ' '.join(BENCH[1:]).replace(ROOT + '/', ''),
    **{k: round(v, 1) for k, v in sorted(means.items())}}
  json.dump({'c26_r1000000_d16_b65536_n1': entry}, open(out_path, 'w'), indent=1)
  print(json.dumps(entry))


if __name__ == '__main__':
  main()
