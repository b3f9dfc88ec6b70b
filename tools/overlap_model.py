"""NS3 made falsifiable on one GPU: does the pipelined sharded step hide its
exchanges behind the gathers, and which form should be the default?

One rank runs the FULL per-rank work of config 3 (26 x 1M x dim16 tables, batch 65536: the owner
gather and the stitch of 1.70 M rows each, as at W = 8 with uniform ids) through
`hbk_sharded_lookup_fwd` over the in-process test transport; its own slice is sent through the
exchange (option sharded_copy_self) and the transport appends an ARTIFICIAL WIRE to every exchange on
the communicator's stream: a kernel that waits  latency + bytes / 8 / rate  -- the time the largest
per-peer message of the same step (1/8 of the rows) would spend on one xGMI link at `rate`, all
seven links in parallel.  The forms of the step are timed under it:

  inline        exchanges enqueued on the compute stream (nothing overlaps)
  one_group     one column group, exchanges on the communicator's stream (two stream hops)
  pipelined_G   G column groups: gather(g) beside ids(g+1), stitch(g) beside rows(g+1)
  pipelined_steps  TWO plans on two compute streams share the communicator (hb.embedding.
                PipelinedLookup): begin(step i + 1) -- ids out, owner gather -- is enqueued BEFORE
                end(step i) -- rows back, stitch --, so ids(i + 1) are on the wire ahead of rows(i):
                one plan gathers while the other's rows travel.  The wire itself stays serial (one
                communicator stream).  Forward-only use (no table update between begin and end)

  python tools/overlap_model.py [--steps 40] > profiles/r05_overlap_model.txt
"""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import hybridbackend_amd as hb  # noqa: E402
from hybridbackend_amd import _lib  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=40)
  ap.add_argument('--columns', type=int, default=26)
  ap.add_argument('--rows', type=int, default=1000000)
  ap.add_argument('--dim', type=int, default=16)
  ap.add_argument('--batch', type=int, default=65536)
  ap.add_argument('--links', type=int, default=8, help='the modelled world size')
  a = ap.parse_args()
  dev = torch.device('cuda:0')
  torch.manual_seed(1)
  tables = [torch.empty(a.rows, a.dim, device=dev).uniform_(-1e-3, 1e-3) for _ in range(a.columns)]
  n_batches = 8
  batches = [[torch.randint(0, 1 << 40, (a.batch,), device=dev) for _ in range(a.columns)]
             for _ in range(n_batches)]
  outs = [torch.empty(a.batch, a.dim, device=dev) for _ in range(a.columns)]
  tlib = _lib.testing_lib()
  forms = [('inline', 0, 1), ('one_group', 1, 0), ('pipelined_2', 2, 0), ('pipelined_3', 3, 0),
           ('pipelined_4', 4, 0)]
  wires = [('no wire', 0.0), ('100 GB/s', 100.0), ('50 GB/s', 50.0), ('25 GB/s', 25.0)]
  row_msg = a.columns * a.batch * a.dim * 4 / a.links
  id_msg = a.columns * a.batch * 4 / a.links
  print(f'# one rank, full per-rank work of config 3 ({a.columns} x {a.rows} x dim{a.dim}, batch '
        f'{a.batch}), own slice through the exchange; artificial wire = 3 us + largest per-peer '
        f'message / rate with W = {a.links}: ids {id_msg / 1e6:.2f} MB, rows {row_msg / 1e6:.2f} MB per link')
  _lib.set_option('sharded_copy_self', 1)
  results = {}
  for wire_name, gbps in wires:
    wire_us = (3.0 + id_msg / (gbps * 1e3)) + (3.0 + row_msg / (gbps * 1e3)) if gbps > 0 else 0.0
    line = {}
    for name, groups, inline in forms:
      _lib.set_option('sharded_groups', groups)
      _lib.set_option('sharded_inline', inline)
      comms = hb.distribute.Collective.local_world(1)
      assert tlib.hbk_testing_set_wire(comms[0]._world, gbps, 3.0, 1.0 / a.links, 1) == 0
      drv = hb.embedding.ShardedGroupLookup(tables, comms[0], buckets=[a.rows] * a.columns,
                                            combiners='sum')
      bound = [drv.bind(batches[b], None, outs) for b in range(n_batches)]

      def step(i):
        drv.launch(bound[i % n_batches])
        drv.prefetch(bound[(i + 1) % n_batches])
      for i in range(5):
        step(i)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(a.steps):
        step(5 + i)
      torch.cuda.synchronize()
      us = (time.perf_counter() - t0) / a.steps * 1e6
      line[name] = round(us, 1)
      drv.close()
      comms[0].close()
    # two steps in flight: two plans (one column group each, exchanges on the communicator's
    # stream) through hb.embedding.PipelinedLookup: begin(step i + 1) BEFORE end(step i), so the ids
    # of step i + 1 are on the wire ahead of the rows of step i
    _lib.set_option('sharded_groups', 1)
    _lib.set_option('sharded_inline', 0)
    for depth in (2, 3):
      comms = hb.distribute.Collective.local_world(1)
      assert tlib.hbk_testing_set_wire(comms[0]._world, gbps, 3.0, 1.0 / a.links, 1) == 0
      drvs = [hb.embedding.ShardedGroupLookup(tables, comms[0], buckets=[a.rows] * a.columns,
                                              combiners='sum') for _ in range(depth)]
      pipe = hb.embedding.PipelinedLookup(drvs)
      outs2 = [outs] + [[torch.empty_like(o) for o in outs] for _ in range(depth - 1)]
      bounds = [[pipe.bind(k, batches[b], None, outs2[k]) for b in range(n_batches)]
                for k in range(depth)]

      def step2(i):
        k = pipe.next_plan()
        pipe.step(bounds[k][i % n_batches], prefetch=bounds[k][(i + depth) % n_batches])
      for i in range(6):
        step2(i)
      pipe.flush()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(a.steps):
        step2(6 + i)
      pipe.flush()
      torch.cuda.synchronize()
      line[f'pipelined_steps_{depth}'] = round((time.perf_counter() - t0) / a.steps * 1e6, 1)
      for d in drvs:
        d.close()
      comms[0].close()
    results[wire_name] = line
    best = min(line, key=line.get)
    print(f'{wire_name:<9} (exchanges of a step: {wire_us:6.1f} us on the wire)  ' +
          '  '.join(f'{k} {v:7.1f}' for k, v in line.items()) + f'   us/step   best: {best}')
  print(json.dumps(results))


if __name__ == '__main__':
  main()
