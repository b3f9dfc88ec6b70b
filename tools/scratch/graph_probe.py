"""Does replaying the backward from a captured hipGraph shorten its launch gaps?  Config 2 emit and the
ragged Poisson(8) case: K direct calls vs K graph replays (one call captured per graph), HIP events."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
import hybridbackend_amd as hb

DEV = 'cuda:0'
cols, rows, dim, B = 26, 1000000, 16, 65536
tables = [torch.empty(rows, dim, device=DEV).uniform_(-1e-3, 1e-3) for _ in range(cols)]
lookup = hb.embedding.GroupLookup(tables, [rows] * cols, 'mean')


def timed(fn, iters=50, warmup=10):
  for i in range(warmup):
    fn(i)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(iters):
    fn(i)
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e3


def case(name, ids, splits, n_seg):
  grads = [torch.randn(n_seg, dim, device=DEV) for _ in range(cols)]
  obj = hb.embedding.GroupLookupGrad(lookup)
  obj(ids, grads, splits)
  direct = timed(lambda i: obj.launch())
  side = torch.cuda.Stream()
  g_obj = hb.embedding.GroupLookupGrad(lookup)
  with torch.cuda.stream(side):
    g_obj(ids, grads, splits)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=side):
    g_obj.launch()
  replay = timed(lambda i: graph.replay())
  with torch.cuda.stream(side):   # the same multi-launch form, not captured
    os.environ['X'] = '1'
  print(f'{name}: direct {direct:.1f} us, graph replay {replay:.1f} us')


ids = [torch.randint(0, 1 << 40, (B,), device=DEV) for _ in range(cols)]
case('config 2 backward emit', ids, None, B)
rng = np.random.RandomState(4242)
sp, rid = [], []
for c in range(cols):
  lens = rng.poisson(8, size=B).clip(0, 32)
  s = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
  sp.append(torch.from_numpy(s).to(DEV))
  rid.append(torch.randint(0, 1 << 40, (int(s[-1]),), device=DEV))
case('ragged Poisson(8) backward', rid, sp, B)
from hybridbackend_amd import _lib
old = _lib.set_option('bwd_onepass', 0)
case('config 2 backward emit, three-launch grouping (what a capture records)', ids, None, B)
_lib.set_option('bwd_onepass', old)
