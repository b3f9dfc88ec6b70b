"""Stress of the backward's row counts: repeated calls, both grouping paths; prints mismatches."""
import sys
import numpy as np
import torch
import hybridbackend_amd as hb
from hybridbackend_amd import _lib

DEV = torch.device('cuda:0')
rng = np.random.RandomState(77)
shapes = ((16, 5000, 20000), (128, 700, 6000), (6, 90, 3000), (32, 100000, 4000), (16, 50, 0))
rows = [r for _, r, _ in shapes]
bad = 0
for onepass in (1, 0, 1):
  _lib.set_option('bwd_onepass', onepass)
  for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    ids = [rng.randint(0, 2**40, size=n).astype(np.int64) for _, _, n in shapes]
    grads = [rng.randn(n, d).astype(np.float32) for d, _, n in shapes]
    tabs = [torch.zeros(r, d, device=DEV) for d, r, _ in shapes]
    lookup = hb.embedding.GroupLookup(tabs, rows, 'sum')
    grad = hb.embedding.GroupLookupGrad(lookup)
    res = grad([torch.from_numpy(i).to(DEV) for i in ids], [torch.from_numpy(g).to(DEV) for g in grads],
               apply_lr=0.05 if it % 2 else 0.0)
    torch.cuda.synchronize()
    for c in range(len(shapes)):
      k = int(res[c][2].item())
      want = np.unique(ids[c] % rows[c])
      if k != want.size:
        bad += 1
        ur = res[c][0][:k].cpu().numpy()
        u, cnt = np.unique(ur, return_counts=True)
        print(f'onepass={onepass} it={it} col={c} got {k} want {want.size} dup rows {u[cnt > 1][:5]} '
              f'missing {np.setdiff1d(want, u)[:5]} extra {np.setdiff1d(u, want)[:5]}', flush=True)
print('mismatches', bad)
