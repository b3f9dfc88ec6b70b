"""Stress of the split / merge path of the backward under the forced-split hook: are rows unique?"""
import sys
import numpy as np
import torch
import hybridbackend_amd as hb
from hybridbackend_amd import _lib

DEV = torch.device('cuda:0')
rng = np.random.RandomState(5)
split, log2p = int(sys.argv[2]) if len(sys.argv) > 2 else 96, int(sys.argv[3]) if len(sys.argv) > 3 else 2
_lib.set_option('bwd_split_pairs', split)
_lib.set_option('bwd_buckets_log2', log2p)
shapes = ((16, 5000, 20000), (128, 700, 6000), (6, 90, 3000), (32, 100000, 4000))
rows = [r for _, r, _ in shapes]
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
  ids = [rng.randint(0, 2**40, size=n).astype(np.int64) for _, _, n in shapes]
  grads = [rng.randn(n, d).astype(np.float32) for d, _, n in shapes]
  tabs = [torch.zeros(r, d, device=DEV) for d, r, _ in shapes]
  lookup = hb.embedding.GroupLookup(tabs, rows, 'sum')
  grad = hb.embedding.GroupLookupGrad(lookup)
  res = grad([torch.from_numpy(i).to(DEV) for i in ids], [torch.from_numpy(g).to(DEV) for g in grads])
  torch.cuda.synchronize()
  for c in range(len(shapes)):
    k = int(res[c][2].item())
    local = ids[c] % rows[c]
    want = np.unique(local)
    ur = res[c][0][:k].cpu().numpy()
    u, cnt = np.unique(ur, return_counts=True)
    if k != want.size or (cnt > 1).any():
      bad += 1
      dup = u[cnt > 1]
      msg = f'it={it} col={c} got {k} want {want.size} dup rows {dup[:4]} (pairs of them: {[int((local == r).sum()) for r in dup[:4]]})'
      print(msg, flush=True)
print('calls with duplicates', bad)
