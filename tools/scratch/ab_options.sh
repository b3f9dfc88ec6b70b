#!/bin/bash
# In-process A/B of library options over sweep cases (tools/sweep.py SWEEP_AB): every measurement
# repeated under each value on the same tensors.   usage: ab_options.sh "opt:v0,v1,.. cases" ...
show() { python -c "
import sys,json
last=None
for l in sys.stdin:
  if not l.startswith('{'):
    if 'rror' in l: print(l.strip()[:200])
    continue
  d=json.loads(l)
  if 'ab' in d: last=d
  elif last: print(d['case'][:78].ljust(78), last['values'], last['us']); last=None"; }
for spec in "$@"; do
  set -- $spec
  echo "== $1 (cases $2)"
  SWEEP_AB=$1 python tools/sweep.py --big --cases $2 2>&1 | show
done
