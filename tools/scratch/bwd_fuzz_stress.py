"""Randomised stress of the backward (not a test: run by hand on a GPU box).  Random column sets, id
distributions, dims, combiners, optimizer modes and bucket options; checks distinct rows, exact counts,
sums against float64."""
import sys
import time
import numpy as np
import torch
import hybridbackend_amd as hb
from hybridbackend_amd import _lib

DEV = torch.device('cuda:0')
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
bad = it = 0
while time.time() < t_end:
  it += 1
  for name, val in (('bwd_onepass', int(rng.randint(0, 2))),
                    ('bwd_buckets_log2', int(rng.choice([-1, -1, -1, 0, 2, 5]))),
                    ('bwd_split_pairs', int(rng.choice([0, 0, 0, 96, 700])))):
    _lib.set_option(name, val)
  n_cols = int(rng.randint(1, 6))
  cols = []
  for _ in range(n_cols):
    d = int(rng.choice([4, 6, 16, 32, 128]))
    rows = int(rng.choice([7, 90, 700, 5000, 100000, 3000000]))
    n = int(rng.choice([0, 1, 37, 3000, 20000, 70000]))
    kind = rng.choice(['uniform', 'zipf', 'hot', 'same'])
    if kind == 'uniform':
      ids = rng.randint(0, 2**40, size=n)
    elif kind == 'zipf':
      ids = rng.zipf(1.2, size=n)
    elif kind == 'hot':
      ids = np.where(rng.rand(n) < 0.5, 3, rng.randint(0, 2**40, size=n))
    else:
      ids = np.full(n, 11)
    cols.append((d, rows, ids.astype(np.int64)))
  mode = rng.choice(['none', 'sgd', 'adagrad', 'sgd_step', 'adagrad_step'])
  tabs = [rng.uniform(-1, 1, size=(r, d)).astype(np.float32) for d, r, _ in cols]
  accs = [np.full((r, d), 0.1, np.float32) for d, r, _ in cols]
  grads = [rng.randn(i.size, d).astype(np.float32) for d, _, i in cols]
  t_dev = [torch.from_numpy(t.copy()).to(DEV) for t in tabs]
  a_dev = [torch.from_numpy(a.copy()).to(DEV) for a in accs]
  lookup = hb.embedding.GroupLookup(t_dev, [r for _, r, _ in cols], 'sum')
  grad = hb.embedding.GroupLookupGrad(lookup, accums=a_dev if 'adagrad' in mode else None)
  lr = 0.0 if mode == 'none' else 0.05
  res = grad([torch.from_numpy(i).to(DEV) for _, _, i in cols], [torch.from_numpy(g).to(DEV) for g in grads],
             apply_lr=lr, optimizer='adagrad' if 'adagrad' in mode else 'sgd', emit='step' not in mode)
  torch.cuda.synchronize()
  for c, (d, rows, ids) in enumerate(cols):
    local = ids % rows
    want_rows = np.unique(local)
    k = int(res[c][2].item())
    g64 = np.zeros((rows, d), np.float64)
    np.add.at(g64, local, grads[c].astype(np.float64))
    msg = None
    if k != want_rows.size:
      msg = f'count {k} != {want_rows.size}'
    elif 'step' not in mode:
      ur = res[c][0][:k].cpu().numpy()
      if np.unique(ur).size != k or not np.array_equal(np.sort(ur), want_rows):
        msg = 'rows not the distinct set'
      else:
        gr = res[c][1][:k].cpu().numpy()
        scale = np.abs(g64).max() + 1.0
        if not np.allclose(gr, g64[ur], rtol=1e-4, atol=1e-5 * scale * 10):
          msg = f'sums differ by {np.abs(gr - g64[ur]).max()}'
    if msg is None and mode != 'none':
      got = t_dev[c].cpu().numpy()
      if 'adagrad' in mode:
        a64 = accs[c].astype(np.float64) + g64 * g64
        ref = tabs[c].astype(np.float64) - lr * g64 / np.sqrt(a64)
      else:
        ref = tabs[c].astype(np.float64) - lr * g64
      scale = np.abs(g64).max() + 1.0
      if not np.allclose(got, ref, rtol=1e-4, atol=1e-5 * scale * 10):
        msg = f'table after {mode} differs by {np.abs(got - ref).max()}'
    if msg:
      bad += 1
      print(f'it={it} col={c} d={d} rows={rows} n={ids.size} mode={mode} opts='
            f'{[_lib.get_option(o) for o in ("bwd_onepass", "bwd_buckets_log2", "bwd_split_pairs")]}: {msg}', flush=True)
print(f'seed {seed}: {it} iterations, {bad} bad columns')
