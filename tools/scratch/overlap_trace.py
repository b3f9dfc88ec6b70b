"""Kernel timeline of a few pipelined steps under the modelled wire (run under rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hybridbackend_amd as hb
from hybridbackend_amd import _lib
dev = torch.device('cuda:0')
cols, rows, dim, batch, links = 26, 1000000, 16, 65536, 8
tables = [torch.empty(rows, dim, device=dev).uniform_(-1e-3, 1e-3) for _ in range(cols)]
nb = 4
batches = [[torch.randint(0, 1 << 40, (batch,), device=dev) for _ in range(cols)] for _ in range(nb)]
tlib = _lib.testing_lib()
_lib.set_option('sharded_copy_self', 1)
_lib.set_option('sharded_groups', 1)
_lib.set_option('sharded_inline', 0)
depth = 2
comms = hb.distribute.Collective.local_world(1)
tlib.hbk_testing_set_wire(comms[0]._world, 50.0, 3.0, 1.0 / links, 1)
drvs = [hb.embedding.ShardedGroupLookup(tables, comms[0], buckets=[rows] * cols) for _ in range(depth)]
pipe = hb.embedding.PipelinedLookup(drvs)
outs = [[torch.empty(batch, dim, device=dev) for _ in range(cols)] for _ in range(depth)]
bounds = [[pipe.bind(k, batches[b], None, outs[k]) for b in range(nb)] for k in range(depth)]
for i in range(16):
  k = pipe.next_plan()
  pipe.step(bounds[k][i % nb], prefetch=bounds[k][(i + depth) % nb])
pipe.flush()
torch.cuda.synchronize()
