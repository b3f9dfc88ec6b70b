"""Randomised stress of the one-launch partition / unique kernels (run by hand on a GPU box): random
column counts, lengths and shard counts, two host threads on their own streams at once."""
import sys
import threading
import time
import numpy as np
import torch
import hybridbackend_amd as hb
from oracle import partition_by_modulo, unique as oracle_unique

DEV = torch.device('cuda:0')
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 30.0)
bad = []
count = [0, 0]


def worker(w):
  rng = np.random.RandomState(100 + w)
  stream = torch.cuda.Stream()
  with torch.cuda.stream(stream):
    while time.time() < t_end:
      count[w] += 1
      n_cols = int(rng.randint(1, 40))
      lens = [int(rng.choice([0, 1, 63, 1024, 1025, 5000, 65536, 70000, 262144])) for _ in range(n_cols)]
      if sum(lens) > 3000000:
        lens = [min(x, 70000) for x in lens]
      P = int(rng.choice([1, 2, 3, 5, 8]))
      xs = [rng.randint(-2**40, 2**40, size=n).astype(np.int64) for n in lens]
      dv = [torch.from_numpy(x).to(DEV, non_blocking=False) for x in xs]
      ys, sizes, idxs = hb.distribute.partition_by_modulo_n(dv, P)
      res = hb.embedding.unique_n(dv[:8])
      stream.synchronize()
      for c, x in enumerate(xs):
        oy, os_, oi = partition_by_modulo(x, P)
        if not (np.array_equal(ys[c].cpu().numpy(), oy) and np.array_equal(sizes[c].cpu().numpy(), os_)
                and np.array_equal(idxs[c].cpu().numpy(), oi)):
          bad.append(f'w{w} it{count[w]} partition col {c} len {lens[c]} P {P}')
      for c, x in enumerate(xs[:8]):
        ou, oidx = oracle_unique(x)
        u, idx, nu = res[c]
        k = int(nu.item())
        if k != ou.size or not np.array_equal(u.cpu().numpy()[:k], ou) or not np.array_equal(idx.cpu().numpy(), oidx):
          bad.append(f'w{w} it{count[w]} unique col {c} len {lens[c]}')


ts = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
for t in ts:
  t.start()
for t in ts:
  t.join()
print(f'{count} iterations, {len(bad)} mismatches', bad[:5])
