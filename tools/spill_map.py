#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit: inside which loops of its ISA.
  hipcc ... -S --cuda-device-only -o /tmp/x.s file.hip ; tools/spill_map.py /tmp/x.s <mangled-name substring> ...
For every matching function: its scratch_load / scratch_store instructions grouped by the innermost
loop (back edge target .. back edge) they fall into, with that loop's global loads as a hint of
which loop it is (the walk of the row-sorted reduce requests W gradient rows per trip)."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
for key in sys.argv[2:]:
  starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l)]
  for st in starts:
    end = st
    while not lines[end].startswith('.Lfunc_end'): end += 1
    body = lines[st + 1:end]
    labs = {}
    for k, l in enumerate(body):
      m = re.match(r'^(\.LBB\S+):', l)
      if m: labs[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
      m = re.search(r's_cbranch_\w+ (\.LBB\S+)|s_branch (\.LBB\S+)', l)
      if m:
        t = labs.get(m.group(1) or m.group(2))
        if t is not None and t < k: loops.append((t, k))
    scr = [k for k, l in enumerate(body) if 'scratch_' in l]
    print(lines[st].split(':')[0][:110], '--', len(body), 'lines,', len(scr), 'scratch instructions')
    out = 0
    by = {}
    for k in scr:
      inner = [lp for lp in loops if lp[0] <= k <= lp[1]]
      if not inner: out += 1; continue
      lp = min(inner, key=lambda x: x[1] - x[0])
      by.setdefault(lp, []).append(k)
    print('   outside any loop:', out)
    for lp, ks in sorted(by.items()):
      seg = body[lp[0]:lp[1] + 1]
      gl = sum('global_load' in l for l in seg)
      ds = sum(l.strip().startswith('ds_') for l in seg)
      print(f'   loop lines {lp[0]}..{lp[1]} ({lp[1]-lp[0]+1} instr, {gl} global loads, {ds} LDS ops): {len(ks)} scratch instructions')
