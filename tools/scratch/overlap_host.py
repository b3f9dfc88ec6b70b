"""Where does a pipelined step spend its time on the HOST?  (diagnostic for tools/overlap_model.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hybridbackend_amd as hb
from hybridbackend_amd import _lib
dev = torch.device('cuda:0')
cols, rows, dim, batch, links = 26, 1000000, 16, 65536, 8
tables = [torch.empty(rows, dim, device=dev).uniform_(-1e-3, 1e-3) for _ in range(cols)]
nb = 8
batches = [[torch.randint(0, 1 << 40, (batch,), device=dev) for _ in range(cols)] for _ in range(nb)]
tlib = _lib.testing_lib()
_lib.set_option('sharded_copy_self', 1)
_lib.set_option('sharded_groups', 1)
_lib.set_option('sharded_inline', 0)
for gbps in (0.0, 50.0):
  for depth in (2, 3):
    comms = hb.distribute.Collective.local_world(1)
    tlib.hbk_testing_set_wire(comms[0]._world, gbps, 3.0, 1.0 / links, 1)
    drvs = [hb.embedding.ShardedGroupLookup(tables, comms[0], buckets=[rows] * cols) for _ in range(depth)]
    pipe = hb.embedding.PipelinedLookup(drvs)
    outs = [[torch.empty(batch, dim, device=dev) for _ in range(cols)] for _ in range(depth)]
    bounds = [[pipe.bind(k, batches[b], None, outs[k]) for b in range(nb)] for k in range(depth)]
    t_begin = t_end = t_pre = 0.0
    def step(i, rec):
      global t_begin, t_end, t_pre
      k = pipe.next_plan()
      b = bounds[k][i % nb]
      pipe._next = (k + 1) % depth
      t0 = time.perf_counter()
      with torch.cuda.stream(pipe.streams[k]):
        pipe.plans[k].launch_begin(b)
      t1 = time.perf_counter()
      if pipe._open is not None:
        k0, b0, pf = pipe._open
        with torch.cuda.stream(pipe.streams[k0]):
          pipe.plans[k0].launch_end(b0)
          t2 = time.perf_counter()
          pipe.plans[k0].prefetch(pf)
        t3 = time.perf_counter()
      else:
        t2 = t3 = t1
      pipe._open = (k, b, bounds[k][(i + depth) % nb])
      if rec:
        t_begin += t1 - t0; t_end += t2 - t1; t_pre += t3 - t2
    for i in range(8): step(i, False)
    torch.cuda.synchronize()
    n = 40
    w0 = time.perf_counter()
    for i in range(n): step(8 + i, True)
    host = time.perf_counter() - w0
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    print(f'wire {gbps:5.1f} GB/s depth {depth}: wall {wall / n * 1e6:7.1f} us/step, host loop {host / n * 1e6:7.1f}; '
          f'begin {t_begin / n * 1e6:6.1f} end {t_end / n * 1e6:6.1f} prefetch {t_pre / n * 1e6:6.1f}', flush=True)
    k0, b0, pf = pipe._open
    with torch.cuda.stream(pipe.streams[k0]):
      pipe.plans[k0].launch_end(b0)
    pipe._open = None
    torch.cuda.synchronize()
    for d in drvs: d.close()
    comms[0].close()
