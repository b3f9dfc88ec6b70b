"""Secondary measurements on one MI355X (not the bench line): batch sweep, ragged multi-hot,
config-4 (dim 128, one 100M-row table, Zipf ids) forward/backward, and the integer kernels.
Prints one JSON object per case; algorithmic bytes follow SURVEY.md 8(d).

  python tools/sweep.py [--cases a,b,c,d,e] [--big]     (--big allocates the 51 GB table)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hybridbackend_amd as hb  # noqa: E402

DEV = torch.device('cuda', 0)
PEAK = 8000.0


def _timed_once(fn, iters, warmup):
  for i in range(warmup):
    fn(i)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for i in range(iters):
    fn(warmup + i)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / iters  # us


def timed(fn, iters=20, warmup=8):
  """SWEEP_AB="bwd_xcd:0,1,0,1": every measurement is repeated under each value of the option,
  inside this process and on the same tensors (between processes the same setting differs by up
  to 15 % on the large-table cases: where the tables land); the extra line carries all times."""
  ab = os.environ.get('SWEEP_AB')
  if not ab:
    return _timed_once(fn, iters, warmup)
  from hybridbackend_amd import _lib
  name, values = ab.split(':')
  values = [int(v) for v in values.split(',')]
  old = _lib.get_option(name)
  times = []
  try:
    for v in values:
      _lib.set_option(name, v)
      times.append(round(_timed_once(fn, iters, warmup), 2))
  finally:
    _lib.set_option(name, old)
  print(json.dumps(dict(ab=name, values=values, us=times)), flush=True)
  return times[0]


def report(name, us, lookups, nbytes, **extra):
  gbs = nbytes / us / 1e3
  print(json.dumps(dict(case=name, us=round(us, 2), M_lookups_per_s=round(lookups / us, 1),
                        algorithmic_GBps=round(gbs, 1), frac_of_8TBps=round(gbs / PEAK, 4),
                        **extra)), flush=True)


def uniform_tables(n, rows, dim):
  return [torch.empty(rows, dim, device=DEV).uniform_(-1e-3, 1e-3) for _ in range(n)]


def zipf_ids(n, rows, alpha, gen, perm_mult):
  """Zipf(alpha) ranks by inverse CDF over [1, rows], scattered over the table by a fixed
  multiplicative permutation ('inverse-CDF on a fixed permutation, seed 7')."""
  u = torch.rand(n, device=DEV, dtype=torch.float64, generator=gen)
  a = 1.0 - alpha
  # continuous approximation of the Zipf CDF: F(x) ~ (x^a - 1) / (R^a - 1)
  x = ((u * (float(rows) ** a - 1.0)) + 1.0) ** (1.0 / a)
  rank = x.floor().clamp_(1, rows).to(torch.int64) - 1
  return (rank * perm_mult) % rows


def case_batch_sweep():
  tables = uniform_tables(26, 1000000, 16)
  for B in (4096, 16384, 65536, 262144, 1048576):
    nb = 8
    batches = [[torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)] for _ in range(nb)]
    outs = [torch.empty(B, 16, device=DEV) for _ in range(26)]
    plans = []
    for b in range(nb):
      gl = hb.embedding.GroupLookup(tables, [1000000] * 26, 'sum')
      gl.bind(batches[b], None, outs)
      plans.append(gl)
    us = timed(lambda i: plans[i % nb].launch(), iters=30)
    report(f'cfg2 fwd H=1 dim16 B={B}', us, 26 * B, 26 * B * 136)
    del batches, outs, plans


def case_ragged():
  tables = uniform_tables(26, 1000000, 16)
  S = 65536
  g = torch.Generator(device=DEV)
  g.manual_seed(5)
  nb = 4
  for comb in ('sum', 'mean'):
    plans, n_tot = [], 0
    for b in range(nb):
      ids, sps = [], []
      for c in range(26):
        lens = torch.poisson(torch.full((S,), 8.0, device=DEV), generator=g).clamp_(0, 32)
        sp = torch.zeros(S + 1, dtype=torch.int32, device=DEV)
        sp[1:] = torch.cumsum(lens, 0).to(torch.int32)
        n = int(sp[-1].item())
        ids.append(torch.randint(0, 1 << 40, (n,), device=DEV))
        sps.append(sp)
        if b == 0:
          n_tot += n
      gl = hb.embedding.GroupLookup(tables, [1000000] * 26, comb)
      gl.bind(ids, sps, None)
      plans.append(gl)
    us = timed(lambda i: plans[i % nb].launch(), iters=20)
    nbytes = n_tot * 8 + n_tot * 64 + 26 * S * 64 + 26 * (S + 1) * 4
    report(f'cfg2 fwd ragged Poisson(8) dim16 S={S} {comb}', us, n_tot, nbytes, ids=n_tot)
    if comb == 'mean':
      gl = plans[0]
      ids0, sps0, _ = gl._keep
      grad = hb.embedding.GroupLookupGrad(gl)
      gouts = [torch.randn(S, 16, device=DEV) for _ in range(26)]
      us = timed(lambda i: grad(ids0, gouts, sps0), iters=10, warmup=2)
      report(f'cfg2 bwd ragged Poisson(8) dim16 S={S} {comb} (IndexedSlices only)', us, n_tot,
             n_tot * 8 + n_tot * 64 + 26 * S * 64, ids=n_tot)


def case_backward_cfg2():
  tables = uniform_tables(26, 1000000, 16)
  B = 65536
  nb = 4
  lookup = hb.embedding.GroupLookup(tables, [1000000] * 26, 'sum')
  grad = hb.embedding.GroupLookupGrad(lookup)
  batches = [[torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)] for _ in range(nb)]
  gouts = [torch.randn(B, 16, device=DEV) for _ in range(26)]
  res = grad(batches[0], gouts)
  u = sum(int(r[2].item()) for r in res)
  for lr, emit, tag in ((0.0, True, 'IndexedSlices only'), (0.01, True, 'fused SGD apply'),
                        (0.01, False, 'SGD step only, no IndexedSlices')):
    us = timed(lambda i: grad(batches[i % nb], gouts, apply_lr=lr, emit=emit), iters=10)
    nbytes = 26 * B * 8 + 26 * B * 64 + u * ((2 * 64 if lr else 0) + (72 if emit else 0))
    report(f'cfg2 bwd H=1 dim16 B={B} ({tag})', us, 26 * B, nbytes, unique_rows=u)


def case_cfg4(big):
  dim, B = 128, 65536
  rows = [1000000] * 25 + [100000000 if big else 10000000]
  tables = [torch.empty(r, dim, device=DEV).uniform_(-1e-3, 1e-3) for r in rows]
  g = torch.Generator(device=DEV)
  g.manual_seed(7)
  nb = 4
  batches = [[zipf_ids(B, rows[c], 1.2, g, 2654435761 % rows[c] | 1) for c in range(26)]
             for _ in range(nb)]
  lookup = hb.embedding.GroupLookup(tables, None, 'sum')
  outs = [torch.empty(B, dim, device=DEV) for _ in range(26)]
  plans = []
  for b in range(nb):
    gl = hb.embedding.GroupLookup(tables, None, 'sum')
    gl.bind(batches[b], None, outs)
    plans.append(gl)
  us = timed(lambda i: plans[i % nb].launch(), iters=20)
  uniq = sum(int(torch.unique(batches[0][c]).numel()) for c in range(26))
  report(f'cfg4 fwd Zipf(1.2) dim128 B={B} big_rows={rows[-1]}', us, 26 * B,
         26 * B * (8 + 512 + 512), unique_rows=uniq,
         dedup_aware_GBps=round((26 * B * (8 + 512) + uniq * 512) / us / 1e3, 1))
  # the store-bound floor of this shape: every id the same row (all row reads hit L1)
  same = [torch.zeros(B, dtype=torch.int64, device=DEV) for _ in range(26)]
  gl = hb.embedding.GroupLookup(tables, None, 'sum')
  gl.bind(same, None, outs)
  us = timed(lambda i: gl.launch(), iters=20)
  report(f'cfg4 fwd dim128 B={B}, one row per column (output stores only)', us, 26 * B,
         26 * B * (8 + 512))
  grad = hb.embedding.GroupLookupGrad(lookup)
  gouts = [torch.randn(B, dim, device=DEV) for _ in range(26)]
  for lr, emit, tag in ((0.0, True, 'IndexedSlices only'), (0.01, True, 'fused SGD apply'),
                        (0.01, False, 'SGD step only, no IndexedSlices')):
    us = timed(lambda i: grad(batches[i % nb], gouts, apply_lr=lr, emit=emit), iters=10)
    nbytes = 26 * B * 8 + 26 * B * 512 + uniq * ((2 * 512 if lr else 0) + (520 if emit else 0))
    report(f'cfg4 bwd Zipf(1.2) dim128 B={B} ({tag})', us, 26 * B, nbytes, unique_rows=uniq)


def case_cfg4_hot_rows(big):
  """Config-4 forward under option fwd_hot_rows: 0 per-wave gather, 1 256-segment tiles with
  repeated rows staged in LDS, 2 the tiles alone; Zipf(1.2), uniform and one-row ids."""
  from hybridbackend_amd import _lib
  dim, B = 128, 65536
  rows = [1000000] * 25 + [100000000 if big else 10000000]
  tables = [torch.empty(r, dim, device=DEV).uniform_(-1e-3, 1e-3) for r in rows]
  g = torch.Generator(device=DEV)
  g.manual_seed(7)
  nb = 4
  kinds = {
      'Zipf(1.2)': [[zipf_ids(B, rows[c], 1.2, g, 2654435761 % rows[c] | 1) for c in range(26)]
                    for _ in range(nb)],
      'uniform': [[torch.randint(0, rows[c], (B,), device=DEV, generator=g) for c in range(26)]
                  for _ in range(nb)],
      'one row': [[torch.zeros(B, dtype=torch.int64, device=DEV) for c in range(26)]] * nb,
  }
  outs = [torch.empty(B, dim, device=DEV) for _ in range(26)]
  only = os.environ.get('SWEEP_J_KINDS')   # e.g. "Zipf(1.2)" for a counter pass over one kind
  for name, batches in kinds.items():
    if only and name not in only.split(','):
      continue
    plans = []
    for b in range(nb):
      gl = hb.embedding.GroupLookup(tables, None, 'sum')
      gl.bind(batches[b], None, outs)
      plans.append(gl)
    uniq = sum(int(torch.unique(batches[0][c]).numel()) for c in range(26))
    # SWEEP_J_XCD="0,2,0,2": the tile -> XCD mapping (option fwd_xcd) toggled inside ONE process --
    # between processes the same setting differs by up to 15 % here (where the tables land)
    xcds = [int(v) for v in os.environ.get('SWEEP_J_XCD', '-1').split(',')]
    for xcd in xcds:
      old_x = _lib.set_option('fwd_xcd', xcd) if xcd >= 0 else None
      for mode in (0, 1, 2):
        old = _lib.set_option('fwd_hot_rows', mode)
        try:
          us = timed(lambda i: plans[i % nb].launch(), iters=20)
        finally:
          _lib.set_option('fwd_hot_rows', old)
        tag = f' fwd_xcd={xcd}' if xcd >= 0 else ''
        report(f'cfg4 fwd {name} dim128 B={B} fwd_hot_rows={mode}{tag}', us, 26 * B,
               26 * B * (8 + 512 + 512), unique_rows=uniq,
               dedup_aware_GBps=round((26 * B * (8 + 512) + uniq * 512) / us / 1e3, 1))
      if old_x is not None:
        _lib.set_option('fwd_xcd', old_x)


def case_integer():
  B = 65536
  ids = [torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)]
  for P in (2, 8):
    us = timed(lambda i: hb.distribute.partition_by_modulo_n(ids, P), iters=20)
    # reads ids twice (histogram + scatter), writes ids + int32 indices
    report(f'partition_by_modulo_n 26 x {B} int64 P={P}', us, 26 * B, 26 * B * (8 + 8 + 8 + 4))
  # the reference's own partition benchmark shape: 100 columns x 100000 ids, 8 partitions
  ids100 = [torch.randint(0, 3 * 100 * 100000, (100000,), device=DEV, dtype=torch.int32)
            for _ in range(100)]
  us = timed(lambda i: hb.distribute.partition_by_modulo_n(ids100, 8), iters=20)
  report('partition_by_modulo_n 100 x 100000 int32 P=8 (reference benchmark shape)', us,
         100 * 100000, 100 * 100000 * (4 + 4 + 4 + 4))
  us = timed(lambda i: hb.embedding.unique_n(ids), iters=10)
  report(f'unique_n 26 x {B} int64', us, 26 * B, 26 * B * (8 + 8 + 4))
  # round 5: the functional forms return plain lists; lazy=True keeps the views unmade
  us = timed(lambda i: hb.distribute.partition_by_modulo_n(ids, 8, lazy=True), iters=20)
  report(f'partition_by_modulo_n(lazy=True) 26 x {B} int64 P=8', us, 26 * B, 26 * B * (8 + 8 + 8 + 4))
  us = timed(lambda i: hb.embedding.unique_n(ids, lazy=True), iters=10)
  report(f'unique_n(lazy=True) 26 x {B} int64', us, 26 * B, 26 * B * (8 + 8 + 4))
  # the bound forms: arguments marshalled once, a call is one foreign call
  for P in (2, 8):
    plan = hb.distribute.PartitionByModuloN(P)
    plan.bind(ids)
    us = timed(lambda i: plan.launch(), iters=20)
    report(f'PartitionByModuloN(bound) 26 x {B} int64 P={P}', us, 26 * B, 26 * B * (8 + 8 + 8 + 4))
    us = timed(lambda i: plan(ids), iters=20)
    report(f'PartitionByModuloN(call, same tensors) 26 x {B} int64 P={P}', us, 26 * B,
           26 * B * (8 + 8 + 8 + 4))
  uplan = hb.embedding.UniqueN()
  uplan.bind(ids)
  us = timed(lambda i: uplan.launch(), iters=10)
  report(f'UniqueN(bound) 26 x {B} int64', us, 26 * B, 26 * B * (8 + 8 + 4))
  # the call forms handed OTHER tensors every step (a training loop's fresh batches): nothing is
  # remembered from the call before, the whole marshalling is paid
  tables = uniform_tables(26, 1000000, 16)
  pool = [[torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)] for _ in range(8)]
  outs = [torch.empty(B, 16, device=DEV) for _ in range(26)]
  lookup = hb.embedding.GroupLookup(tables, [1000000] * 26, 'sum')
  us = timed(lambda i: lookup(pool[i % 8], None, outs), iters=30)
  report(f'GroupLookup.__call__ (new id tensors every call) 26 x {B}', us, 26 * B, 26 * B * 136)
  us = timed(lambda i: lookup(pool[i % 8]), iters=30)
  report(f'GroupLookup.__call__ (new ids, outputs allocated) 26 x {B}', us, 26 * B, 26 * B * 136)
  coll = hb.distribute.Collective(world_size=1, rank=0)
  drv = hb.embedding.ShardedGroupLookup(tables, coll, buckets=[1000000] * 26)
  us = timed(lambda i: drv(pool[i % 8], None, outs), iters=30)
  report(f'ShardedGroupLookup.__call__ W=1 (new id tensors every call) 26 x {B}', us, 26 * B,
         26 * B * 136)

  def call_and_prefetch(i):
    drv(pool[i % 8], None, outs)
    drv.prefetch(pool[(i + 1) % 8])       # the loader knows the next batch
  us = timed(call_and_prefetch, iters=30)
  report(f'ShardedGroupLookup.__call__ W=1 + prefetch(next ids) (new id tensors every call) 26 x {B}',
         us, 26 * B, 26 * B * 136)
  drv.close()
  coll.close()


def case_bwd_probe():
  """Isolates the backward's regimes: dim 128, 26 columns x 65536 ids, 1M-row tables;
  uniform ids vs Zipf ids vs a single hot row."""
  dim, B, rows = 128, 65536, 1000000
  tables = [torch.zeros(rows, dim, device=DEV) for _ in range(26)]
  lookup = hb.embedding.GroupLookup(tables, None, 'sum')
  grad = hb.embedding.GroupLookupGrad(lookup)
  gouts = [torch.randn(B, dim, device=DEV) for _ in range(26)]
  g = torch.Generator(device=DEV)
  g.manual_seed(3)
  kinds = {
    'uniform': lambda: torch.randint(0, rows, (B,), device=DEV),
    'zipf1.2': lambda: zipf_ids(B, rows, 1.2, g, 7919),
    'hot20pct': lambda: torch.where(torch.rand(B, device=DEV) < 0.2,
                                    torch.full((B,), 12345, device=DEV),
                                    torch.randint(0, rows, (B,), device=DEV)),
    'all_same': lambda: torch.full((B,), 777, device=DEV, dtype=torch.int64),
  }
  for name, mk in kinds.items():
    ids = [mk() for _ in range(26)]
    us = timed(lambda i: grad(ids, gouts), iters=5, warmup=2)
    report(f'bwd probe dim128 {name}', us, 26 * B, 26 * B * (8 + 512 + 512))


def case_cfg5():
  """Config 5 shape on one GPU: 200 columns, dims cycling over {4..128}, rows log-uniform in
  [1e3, 1e7], one third of the columns ragged; fused lookup, then backward + SGD apply."""
  rng = np.random.RandomState(5)
  dims_cycle = [4, 8, 12, 16, 24, 32, 36, 48, 64, 80, 128]
  n, B = 200, 65536
  dims = [dims_cycle[c % len(dims_cycle)] for c in range(n)]
  rows = [int(10 ** rng.uniform(3, 7)) for _ in range(n)]
  tables = [torch.empty(rows[c], dims[c], device=DEV).uniform_(-1e-3, 1e-3) for c in range(n)]
  ids, splits, n_ids, n_bytes = [], [], 0, 0
  for c in range(n):
    if c % 3 == 0:
      lens = torch.poisson(torch.full((B,), 4.0, device=DEV)).clamp_(0, 16)
      sp = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
      sp[1:] = torch.cumsum(lens, 0).to(torch.int32)
      k = int(sp[-1].item())
      splits.append(sp)
    else:
      k = B
      splits.append(None)
    ids.append(torch.randint(0, 1 << 40, (k,), device=DEV))
    n_ids += k
    n_bytes += k * 8 + k * 4 * dims[c] + B * 4 * dims[c]
  lookup = hb.embedding.GroupLookup(tables, rows, 'mean')
  outs = lookup.bind(ids, splits, None)
  us = timed(lambda i: lookup.launch(), iters=10)
  report(f'cfg5 fwd 200 cols mixed dims B={B} (1/3 ragged)', us, n_ids, n_bytes, ids=n_ids)
  grad = hb.embedding.GroupLookupGrad(lookup)
  gouts = [torch.randn_like(o) for o in outs]
  us = timed(lambda i: grad(ids, gouts, splits, apply_lr=0.01), iters=5, warmup=2)
  report(f'cfg5 bwd + SGD apply 200 cols mixed dims B={B}', us, n_ids, n_bytes, ids=n_ids)
  us = timed(lambda i: grad(ids, gouts, splits, apply_lr=0.01, emit=False), iters=5, warmup=2)
  report(f'cfg5 bwd SGD step only 200 cols mixed dims B={B}', us, n_ids, n_bytes, ids=n_ids)
  accums = [torch.full_like(t, 0.1) for t in tables]
  grad_a = hb.embedding.GroupLookupGrad(lookup, accums=accums)
  us = timed(lambda i: grad_a(ids, gouts, splits, apply_lr=0.01, optimizer='adagrad'), iters=5,
             warmup=2)
  report(f'cfg5 bwd + Adagrad apply 200 cols mixed dims B={B}', us, n_ids, n_bytes, ids=n_ids)
  # lever (d) of VERDICT r04 / r05: weights and accumulator interleaved row by row (table_pitch = 2 dim)
  del accums, grad_a
  inter = [torch.empty(rows[c], 2 * dims[c], device=DEV) for c in range(n)]
  for c in range(n):
    inter[c][:, :dims[c]].copy_(tables[c])
    inter[c][:, dims[c]:].fill_(0.1)
  grad_i = hb.embedding.GroupLookupGrad(lookup, interleaved=inter)
  us = timed(lambda i: grad_i(ids, gouts, splits, apply_lr=0.01, optimizer='adagrad'), iters=5,
             warmup=2)
  report(f'cfg5 bwd + Adagrad apply, weights + accumulator interleaved per row, B={B}', us, n_ids,
         n_bytes, ids=n_ids)
  us = timed(lambda i: grad_i(ids, gouts, splits, apply_lr=0.01, optimizer='adagrad', emit=False),
             iters=5, warmup=2)
  report(f'cfg5 bwd Adagrad step only, interleaved, B={B}', us, n_ids, n_bytes, ids=n_ids)


def case_dense_block():
  """26 columns written as the blocks of ONE [batch, 26 x 16] tensor in place (out_stride) vs 26
  separate outputs + a concat pass (what DenseFeatures does on top of per-column lookups)."""
  tables = uniform_tables(26, 1000000, 16)
  B = 65536
  lookup = hb.embedding.GroupLookup(tables, [1000000] * 26, 'sum')
  ids = [torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)]
  block = torch.empty(B, 26 * 16, device=DEV)
  views = [block[:, c * 16:(c + 1) * 16] for c in range(26)]
  lookup.bind(ids, None, views)
  us = timed(lambda i: lookup.launch(), iters=30)
  report(f'dense block [B, 416] written in place, 26 cols dim16 B={B}', us, 26 * B, 26 * B * 136)
  outs = lookup.bind(ids, None, None)

  def two_pass(i):
    lookup.launch()
    torch.cat(list(outs), dim=1, out=block)
  us = timed(two_pass, iters=30)
  report(f'26 separate outputs + concat pass, 26 cols dim16 B={B}', us, 26 * B, 26 * B * 136)


def case_sharded_world1():
  """The whole sharded pipeline (partition -> RCCL alltoallv -> owner gather -> alltoallv ->
  stitch) on one GPU with a world-size-1 communicator: kernel chain + host overhead per step."""
  import time
  tables = uniform_tables(26, 1000000, 16)
  B, nb = 65536, 8
  coll = hb.distribute.Collective(world_size=1, rank=0)
  drv = hb.embedding.ShardedGroupLookup(tables, coll, buckets=[1000000] * 26)
  batches = [[torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(26)] for _ in range(nb)]
  outs = [torch.empty(B, 16, device=DEV) for _ in range(26)]
  bound = [drv.bind(b, None, outs) for b in batches]
  for wire in (None, torch.float16):
    drv.close()               # the wire format is fixed when the plan is created
    drv.wire_dtype = wire
    us = timed(lambda i: drv.launch(bound[i % nb]), iters=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
      drv.launch(bound[i % nb])
    host_us = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize()
    # the call's wall time includes its one wait for the device (the sizes); the plan reports the
    # host's own phases of the last step: enqueue partition + size exchange / wait / enqueue rest
    phases = [round(v, 1) for v in drv.last_host_us()]
    report(f'sharded pipeline W=1 fwd dim16 B={B} wire={"fp16" if wire else "fp32"}', us, 26 * B,
           26 * B * 136, call_wall_us=round(host_us, 1), host_enqueue_wait_enqueue_us=phases,
           host_enqueue_us=round(phases[0] + phases[2], 1))
  drv.close()
  drv.wire_dtype = None
  gouts = [torch.randn(B, 16, device=DEV) for _ in range(26)]

  bouts = [(torch.empty(B, dtype=torch.int64, device=DEV), torch.empty(B, 16, device=DEV),
            torch.zeros(1, dtype=torch.int32, device=DEV)) for _ in range(26)]

  def fwd_bwd(i):
    drv.launch(bound[i % nb])
    drv.backward(gouts, apply_lr=0.01, outs=bouts)
  us_fb = timed(fwd_bwd, iters=20)
  report(f'sharded pipeline W=1 fwd + bwd + SGD dim16 B={B} wire=fp32', us_fb, 26 * B,
         26 * B * (136 + 8 + 64 + 128))

  def fwd_step(i):
    drv.launch(bound[i % nb])
    drv.backward(gouts, apply_lr=0.01, emit=False)
  us_fb = timed(fwd_step, iters=20)
  report(f'sharded pipeline W=1 fwd + SGD step only dim16 B={B} wire=fp32', us_fb, 26 * B,
         26 * B * (136 + 8 + 64 + 128))
  coll.close()


def case_sharded_dedup(big):
  """Requester-side dedup of the sharded driver on the config-4 shape (dim 128, Zipf(1.2) ids, one
  large table), one rank through RCCL: ids / rows the step puts "on the wire" (what would cross
  xGMI at W > 1, per rank) and the step time, without and with dedup."""
  dim, B = 128, 65536
  rows = [1000000] * 25 + [100000000 if big else 10000000]
  tables = [torch.empty(r, dim, device=DEV).uniform_(-1e-3, 1e-3) for r in rows]
  g = torch.Generator(device=DEV)
  g.manual_seed(7)
  nb = 4
  batches = [[zipf_ids(B, rows[c], 1.2, g, 2654435761 % rows[c] | 1) for c in range(26)]
             for _ in range(nb)]
  outs = [torch.empty(B, dim, device=DEV) for _ in range(26)]
  gouts = [torch.randn(B, dim, device=DEV) for _ in range(26)]
  coll = hb.distribute.Collective(world_size=1, rank=0)
  for dedup in (False, True):
    drv = hb.embedding.ShardedGroupLookup(tables, coll, buckets=rows, dedup=dedup)
    bound = [drv.bind(b, None, outs) for b in batches]
    us = timed(lambda i: drv.launch(bound[i % nb]), iters=20)
    sent = sum(int(drv._lib.hbk_sharded_owned_ids(drv._plan(), c)) for c in range(26))
    wire = sent * (4 + dim * 4)
    report(f'sharded W=1 cfg4 Zipf(1.2) dim128 fwd dedup={int(dedup)}', us, 26 * B,
           26 * B * (8 + 512 + 512), ids_on_wire=sent, wire_bytes_per_step=wire)

    def fwd_step(i):
      drv.launch(bound[i % nb])
      drv.backward(gouts, apply_lr=0.01, emit=False)
    us = timed(fwd_step, iters=10, warmup=3)
    report(f'sharded W=1 cfg4 Zipf(1.2) dim128 fwd + SGD step only dedup={int(dedup)}', us,
           26 * B, 26 * B * (8 + 512 + 512 + 512), ids_on_wire=sent,
           wire_bytes_per_step=wire + sent * dim * 4)
    drv.close()
  coll.close()


def case_dense_features_auto_hot(big):
  """Config-4 shape through DenseFeatures with NO user flag (EmbeddingColumn's default
  hot_rows='auto'): the forward before any backward (per-wave gather), then after one backward
  over the same kind of ids has told the layer how many distinct rows a batch names."""
  dim, B = 128, 65536
  rows = [1000000] * 25 + [100000000 if big else 10000000]
  cols = [hb.feature_column.EmbeddingColumn(f'c{c}', rows[c], dim, 'sum') for c in range(26)]
  layer = hb.feature_column.DenseFeatures(cols, DEV)
  g = torch.Generator(device=DEV)
  g.manual_seed(7)
  nb = 4
  kinds = {
      'Zipf(1.2)': [{f'c{c}': zipf_ids(B, rows[c], 1.2, g, 2654435761 % rows[c] | 1)
                     for c in range(26)} for _ in range(nb)],
      'uniform': [{f'c{c}': torch.randint(0, rows[c], (B,), device=DEV, generator=g)
                   for c in range(26)} for _ in range(nb)],
  }
  grad = torch.randn(B, 26 * dim, device=DEV)
  for name, batches in kinds.items():
    for phase in ('before any backward', 'after a backward'):
      if phase == 'after a backward':
        layer(batches[0])
        layer.backward(grad)
        torch.cuda.synchronize()
        layer(batches[0])          # (this forward finds the counts landed and switches)
        torch.cuda.synchronize()
      us = timed(lambda i: layer(batches[i % nb]), iters=20)
      hot = sum(int(layer._lookup._cols[c].hot_rows) for c in range(26))
      report(f'DenseFeatures cfg4 fwd {name} dim128 B={B}, no flag, {phase}', us, 26 * B,
             26 * B * (8 + 512 + 512), columns_staging_hot_rows=hot)
    # forget what was learnt: the next kind starts from scratch
    for c in range(26):
      layer._lookup._cols[c].hot_rows = 0
    layer._lookup._auto_state = None


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--cases', default='a,b,c,d,e')
  ap.add_argument('--big', action='store_true')
  args = ap.parse_args()
  torch.manual_seed(0)
  for c in args.cases.split(','):
    {'a': case_batch_sweep, 'b': case_ragged, 'c': case_backward_cfg2,
     'd': lambda: case_cfg4(args.big), 'e': case_integer, 'f': case_bwd_probe, 'g': case_sharded_world1, 'h': case_cfg5, 'i': case_dense_block,
     'j': lambda: case_cfg4_hot_rows(args.big), 'k': lambda: case_sharded_dedup(args.big),
     'l': lambda: case_dense_features_auto_hot(args.big)}[c]()
    torch.cuda.empty_cache()
