"""bench.py -- M-lookups/sec of the sharded-embedding hot path on MI355X.

Workload (BASELINE.json configs[1]): 26 Criteo-shape categorical columns,
1M-row x dim-16 fp32 tables (uniform(-1e-3, 1e-3), seed 1234+col), int64 ids uniform in
[0, 2^40) (seed 42+col+step) bucketized `% 1_000_000` inside the fused kernel, one id per
sample (Criteo columns are scalar), local batch 65536.  One "step" = one pass of the hot
path over one fresh batch:
  N = 1 : hbk_group_lookup_fwd (bucketize -> HBM row gather -> combiner), one launch.
  N > 1 : tables row-sharded by `id mod N` (configs[2]): bucketize -> stable partition ->
          RCCL alltoallv(ids) -> owner gather -> RCCL alltoallv(rows) -> stitch/combine,
          per-GPU batch fixed (weak scaling); value = lookups of all ranks / max-rank time.
Inputs are resident in HBM before the timed region; every step reads a different id batch
so nothing is served from L2/MALL by repetition (1.66 GB of tables >> 256 MB).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM-bound
gather: algorithmic 136 B/lookup = 8 id + 64 row read + 64 output write) and
`cpu_baseline` (the oracle's C pipeline timed on the host cores, rank 0, N = 1).

Launching.  `python bench.py --gpus N` is enough: without WORLD_SIZE in the environment and
N > 1 the script starts N ranks of itself on a free port (one process per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would) and relays rank 0's
JSON line; launched under `python -m torch.distributed.run --nproc-per-node N ...` it uses the
environment it is given.  The ranks rendezvous over gloo (host TCP): the barrier, the max over
ranks and the broadcast of the 128-byte RCCL id need no second RCCL communicator next to the
library's own.  If fewer than N GPUs are visible a JSON line with an "error" key is printed and
the exit code is 2.  `--dry-run` exercises launcher + rendezvous without touching a GPU.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# dmabuf IPC (the host driver supports nothing else): RCCL between processes needs it, and the
# HSA runtime reads it when the first HIP call initialises it -- so before torch is imported
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: 6.29 TB/s measured with a float4 copy (79 % of spec)


# reference measurements that follow the headline steps at N > 1 (or --sharded), in `config`
SECONDARY_KEYS = ('replicated_M_lookups_per_s', 'replicated_ms_per_step',
                  'other_wire', 'other_wire_M_lookups_per_s', 'other_wire_ms_per_step',
                  'secondary_steps')


# the step forms the probe BEHIND the timed steps runs at N > 1 (or --sharded): the timed steps run in
# the shipped default ('inline') first, and again in the fastest of the first four when the probe
# finds one more than 3 % faster (config.sharded_form says which measurement `value` is);
# 'pipelined_steps_3' (three plans, begin(i + 1) before end(i): outputs arrive one step late,
# forward-only use) is a reference
FORM_KEYS = ('pipelined_2_groups', 'one_group', 'inline', 'p2p', 'pipelined_steps_3')
# what an N > 1 line carries under 'xgmi'
XGMI_KEYS = ('bytes_out_per_rank_per_step', 'links_per_rank', 'achieved_GBps_per_rank_each_way',
             'link_probe', 'link_bound', 'rccl', 'note')


def multi_gpu_skeleton():
  """Every key a measured N > 1 line carries next to the headline, with nothing measured: what
  `--dry-run` prints (tests/test_bench_launcher.py checks the names for N = 2, 4, 8, so the first
  multi-GPU box cannot be met by a line that lacks one)."""
  config = {key: None for key in SECONDARY_KEYS}
  config.update({'rccl_ranks_seen': None, 'sharded_form': None,
                 'sharded_form_probe_ms_per_step': {key: None for key in FORM_KEYS},
                 'value_at_shipped_default_M_lookups_per_s': None, 'wire': None,
                 'prefetch_next_partition': None})
  return config, {key: None for key in XGMI_KEYS}


def link_bound(args, world, probe, lookups_per_step_per_rank):
  """The ceiling the measured links put on the sharded step: uniform ids send 1/W of a
  rank's lookups to every peer over that peer's own link -- int32 ids out, fp32 | fp16 rows back
  (each way carries ids + rows of one of the two ranks) -- at the per-link rate the probe measured."""
  if not probe or not probe.get('GBps_per_link_each_way'):
    return None
  id_b = 4 if args.rows <= 0x7fffffff else 8
  row_b = args.dim * (2 if args.wire == 'fp16' else 4)
  per_peer = lookups_per_step_per_rank / world * (id_b + row_b)
  sec = per_peer / (probe['GBps_per_link_each_way'] * 1e9)
  return {'bytes_per_link_each_way_per_step': int(per_peer), 'us_per_step': round(sec * 1e6, 1),
          'M_lookups_per_s': round(lookups_per_step_per_rank * world / sec / 1e6, 1),
          'from': 'link_probe.GBps_per_link_each_way'}


def parse_args():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=50)
  p.add_argument('--warmup', type=int, default=10)
  p.add_argument('--batch', type=int, default=65536, help='ids per column per GPU and step')
  p.add_argument('--columns', type=int, default=26)
  p.add_argument('--rows', type=int, default=1000000)
  p.add_argument('--dim', type=int, default=16)
  p.add_argument('--wire', choices=['fp32', 'fp16'], default='fp32',
                 help='N > 1: wire dtype of the embedding exchange (comm_wire_dtype)')
  p.add_argument('--cpu-seconds', type=float, default=12.0,
                 help='budget of the host-CPU baseline (0 disables it)')
  p.add_argument('--sharded', action='store_true',
                 help='run the sharded pipeline even at N = 1 (validation of the N > 1 code path)')
  p.add_argument('--no-prefetch', action='store_true',
                 help='N > 1: do not partition the next batch during the current step')
  p.add_argument('--watchdog', type=float, default=900.0,
                 help='N > 1: seconds after which a rank exits instead of waiting for its peers')
  p.add_argument('--id-batches', type=int, default=0,
                 help='distinct id batches kept in HBM (default: steps + warmup, max 64)')
  p.add_argument('--dry-run', action='store_true',
                 help='launcher + rendezvous + barrier + reduction only (no GPU, no kernels)')
  p.add_argument('--tune-steps', type=int, default=8,
                 help='N > 1 (or --sharded): untimed steps per candidate of the pipeline group '
                      'count (option sharded_groups: 2 = exchanges of one column group overlap the '
                      'gather / stitch of the other, 1 = no pipelining, half the cross-stream '
                      'hops); the faster one runs the timed steps.  0: keep the default')
  p.add_argument('--p2p', choices=['auto', 'on', 'off'], default='auto',
                 help='probe the p2p form of the sharded step (owners store rows straight into the '
                      'requester\'s registered outputs: hbk_sharded_p2p_bind) next to the exchange '
                      'forms and run the timed steps in it when it is the fastest.  auto: at one '
                      'rank (--sharded) only -- across processes the form maps peer memory through '
                      'hipIpc*, which has not run on this hardware yet; on: at any N')
  p.add_argument('--no-secondary', action='store_true',
                 help='N = 1: skip the backward family behind the headline (config.secondary_steps: '
                      'emit / + SGD / step only / ragged, ~2 s).  '
                      'N > 1 (or --sharded): skip the two reference measurements that follow the '
                      'headline steps -- the other wire format, and every rank holding ALL tables '
                      '(replicated, no exchange: SURVEY 8e "report replicated as a reference line")')
  p.add_argument('--link-probe-mb', type=float, default=16.0,
                 help='N > 1: MB per peer of the equal-split alltoallv that measures the links '
                      'before the timed steps (0 disables it)')
  return p.parse_args()


def _free_port():
  sock = socket.socket()
  sock.bind(('127.0.0.1', 0))
  port = sock.getsockname()[1]
  sock.close()
  return port


def error_line(args, message):
  """The one JSON line of a run that could not be measured."""
  return json.dumps({
    'metric': 'M-lookups/sec, 26-col Criteo-shape dim16, 1/2/4/8 GPUs; % HBM roofline',
    'value': None, 'unit': 'M-lookups/sec', 'n_gpus': args.gpus, 'steps': args.steps,
    'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak',
    'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'error': message})


def launch_ranks(args):
  """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process
  per GPU), relay rank 0's JSON line, never leave a rank behind."""
  n = args.gpus
  if not args.dry_run:
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < n:
      print(error_line(args, f'--gpus {n} but only {visible} GPU(s) are visible on this node'),
            flush=True)
      return 2
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it
  env['MASTER_ADDR'] = '127.0.0.1'
  env['MASTER_PORT'] = str(_free_port())
  env['WORLD_SIZE'] = env['LOCAL_WORLD_SIZE'] = str(n)
  procs = []
  for r in range(n):
    e = dict(env)
    e['RANK'] = e['LOCAL_RANK'] = str(r)
    procs.append(subprocess.Popen(
      [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
  deadline = time.time() + args.watchdog + 60
  out0 = ''
  rc = 0
  try:
    out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
    rc = procs[0].returncode
    for pr in procs[1:]:
      pr.wait(timeout=max(1.0, deadline - time.time()))
      rc = rc or pr.returncode
  except subprocess.TimeoutExpired:
    rc = 3
  finally:
    for pr in procs:       # exact PIDs only
      if pr.poll() is None:
        pr.kill()
  line = None
  for ln in out0.splitlines():
    if ln.startswith('{'):
      line = ln
  if line is None or rc != 0:
    if line is None:
      line = error_line(args, f'rank processes failed (exit code {rc}) before a result was printed')
    print(line, flush=True)
    return rc or 1
  print(line, flush=True)
  return 0


def link_probe(coll, device, world, mb_per_peer, dist):
  """SURVEY 8e pre-measurement (the equal-split mode of the reference's
  hybridbackend/tensorflow/benchmarks/collective_benchmark.py:74-102): every rank sends
  `mb_per_peer` MB to every peer through hbk_alltoallv_n; returns the achieved GB/s per link and
  direction (slowest rank)."""
  n = int(mb_per_peer * 1e6 / 4)
  if n <= 0 or world < 2:
    return None
  src = torch.ones(n * world, device=device, dtype=torch.float32)
  dst = torch.empty_like(src)
  sizes = [[n] * world]
  for _ in range(2):
    coll.alltoallv_n([src], sizes, sizes, common_sizes=[1], outs=[dst])
  torch.cuda.synchronize()
  dist.barrier()
  iters = 5
  t0 = time.perf_counter()
  for _ in range(iters):
    coll.alltoallv_n([src], sizes, sizes, common_sizes=[1], outs=[dst])
  torch.cuda.synchronize()
  el = time.perf_counter() - t0
  t = torch.tensor([el], dtype=torch.float64)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  per = float(t.item()) / iters
  return {'bytes_per_peer': n * 4, 'us_per_exchange': round(per * 1e6, 1),
          'GBps_per_link_each_way': round(n * 4 / per / 1e9, 2),
          'GBps_out_per_rank': round(n * 4 * (world - 1) / per / 1e9, 2)}


def make_tables(args, device, rank, world):
  """Full tables at N=1; this rank's rows (id mod N == rank, variables.py:107-111) at N>1.
  Row r of column c is generated from (seed 1234+c) over the FULL table and sliced, so a
  sharded run holds exactly the rows of the unsharded one."""
  tables = []
  for c in range(args.columns):
    g = torch.Generator(device=device)
    g.manual_seed(1234 + c)
    full = torch.empty(args.rows, args.dim, device=device, dtype=torch.float32)
    full.uniform_(-1e-3, 1e-3, generator=g)
    if world > 1:
      tables.append(full[rank::world].contiguous())
      del full
    else:
      tables.append(full)
  return tables


def make_id_batches(args, device, rank, n_batches):
  batches = []
  for b in range(n_batches):
    cols = []
    for c in range(args.columns):
      g = torch.Generator(device=device)
      g.manual_seed(42 + c + 1000 * b + 100000 * rank)
      cols.append(torch.randint(0, 1 << 40, (args.batch,), device=device, dtype=torch.int64,
                                generator=g))
    batches.append(cols)
  return batches


def cpu_baseline(args, tables, id_batch, budget_s):
  """The oracle's C restatement of the path (kind "port": bucketize -> partition(P=1) ->
  unique -> gather -> restore/stitch -> combiner, oracle/hbk_oracle.c) on the host cores,
  on a bounded sample of the same workload."""
  import oracle  # the checker / reported baseline only
  cores = os.cpu_count() or 1
  # one task per (column, slice of the batch): the reference's CPU analogue runs the per-column
  # ops on TF's inter-op pool; slicing the batch as well lets the port use all host cores
  cols = args.columns
  slices = max(1, min(cores // max(cols, 1), 16))
  threads = max(1, min(cores, cols * slices))
  h_tab = [t.cpu().numpy() for t in tables[:cols]]
  h_all = [i.cpu().numpy() for i in id_batch[:cols]]
  h_tables, h_ids = [], []
  for c in range(cols):
    for part in np.array_split(h_all[c], slices):
      h_tables.append(h_tab[c])
      h_ids.append(np.ascontiguousarray(part))
  tasks = len(h_ids)
  buckets = [args.rows] * tasks
  comb = ['sum'] * tasks
  oracle.group_lookup_fwd(h_tables, h_ids, [None] * tasks, buckets, comb, n_threads=threads)
  passes, t0, rep = 0, time.perf_counter(), 8
  while True:
    oracle.group_lookup_fwd(h_tables, h_ids, [None] * tasks, buckets, comb, n_threads=threads,
                            repeat=rep)
    passes += rep
    el = time.perf_counter() - t0
    if el >= budget_s or passes >= 5000:
      break
  lookups = passes * cols * args.batch
  # the same pipeline on ONE host thread (both figures are reported; the
  # reference's own CPU partition functor is single-threaded, partition_by_modulo_functors.cc:48-69)
  one_tab, one_ids = h_tab[:cols], h_all[:cols]
  one_args = ([None] * cols, [args.rows] * cols, ['sum'] * cols)
  oracle.group_lookup_fwd(one_tab, one_ids, *one_args, n_threads=1)
  p1, t1 = 0, time.perf_counter()
  while True:
    oracle.group_lookup_fwd(one_tab, one_ids, *one_args, n_threads=1)
    p1 += 1
    el1 = time.perf_counter() - t1
    if el1 >= max(1.0, budget_s / 4) or p1 >= 200:
      break
  one_thread = p1 * cols * args.batch / el1 / 1e6
  return {
    'value': round(lookups / el / 1e6, 3), 'unit': 'M-lookups/sec', 'cores': threads,
    'kind': 'port',
    'sample': f'{passes} passes of one {cols}-column x {args.batch}-id batch '
              f'({lookups} lookups, {el:.1f} s) through oracle/hbk_oracle.c '
              f'orc_group_lookup_fwd, {threads} pthreads over {tasks} (column, batch-slice) '
              f'tasks with per-thread scratch (no allocation, no lock per task), host '
              f'nproc={cores}; thread scaling {lookups / el / 1e6 / max(one_thread, 1e-9):.1f} x '
              f'over one thread: random 64-byte rows out of 1.66 GB of tables are bound by the '
              f'host\'s memory latency, not by its cores',
    'single_thread': {
      'value': round(p1 * cols * args.batch / el1 / 1e6, 3), 'unit': 'M-lookups/sec', 'cores': 1,
      'sample': f'{p1} passes of the same batch on one thread ({el1:.1f} s)'}}


def backward_secondary(args, hb, tables, batches, device, timed_steps, steps=20, warmup=5):
  """The backward family behind the headline, in the driver-timed line (north_star
  names "the backward scatter-add" as a hot-path function): config 2's backward through the same
  bracket as the headline steps --

    bwd_emit       d(combiner) -> duplicate-row reduction -> IndexedSlices (unique rows, summed rows)
    bwd_sgd        the same + the fused sparse SGD step on the tables
    bwd_step_only  the fused step alone, no IndexedSlices written
    bwd_ragged     26 columns x batch segments of Poisson(8) ids clipped to [0, 32] (SURVEY 8d's
                   multi-hot variant), mean combiner, IndexedSlices
    bwd_emit_det / bwd_ragged_det   the first and the last under option bwd_deterministic = 1

  Every call reads another resident id batch (tables 1.66 GB, 4-8 batches: nothing is served from the
  Infinity Cache by repetition).  `frac` = algorithmic bytes / time / 8 TB/s with SURVEY 8(d)'s
  bytes_bwd = n*8 (ids) + S*4D (gradient rows read) + u*(2*4D) (row read-modify-write), u = the
  distinct rows the call itself reported (n_unique); a case that writes IndexedSlices moves
  u*(4D + 8) for them (instead of, or with the step, on top of, the read-modify-write); ragged adds
  the (S+1)*4 row splits."""
  cols, dim, batch = args.columns, args.dim, args.batch
  pool = min(len(batches), 8)
  gen = torch.Generator(device=device)
  gen.manual_seed(777)
  out = {}

  def run(name, ids_pool, splits, n_seg, combiner, lr, emit, deterministic=0):
    from hybridbackend_amd import _lib
    old_det = _lib.set_option('bwd_deterministic', deterministic)
    try:
      run_case(name, ids_pool, splits, n_seg, combiner, lr, emit)
    finally:
      _lib.set_option('bwd_deterministic', old_det)

  def run_case(name, ids_pool, splits, n_seg, combiner, lr, emit):
    grads = [torch.randn(n_seg[c], dim, device=device, generator=gen) for c in range(cols)]
    lookup = hb.embedding.GroupLookup(tables, buckets=[args.rows] * cols, combiners=combiner)
    # one gradient object per resident batch: handed the same tensors again, a call re-binds nothing
    objs = []
    for _ in ids_pool:     # (one workspace for all of them: they run one after the other)
      objs.append(hb.embedding.GroupLookupGrad(lookup, workspace_of=objs[0] if objs else None))

    for k, obj in enumerate(objs):      # binds the batch's tensors (validation + marshalling, once)
      obj(ids_pool[k], grads, splits, apply_lr=lr, emit=emit)

    def step(i):
      # the batches are resident: a step is one C-ABI call, as in the headline
      objs[i % len(ids_pool)].launch(apply_lr=lr)
    el, ms = timed_steps(step, steps, warmup)
    res = objs[0](ids_pool[0], grads, splits, apply_lr=lr, emit=emit)
    torch.cuda.synchronize()
    n = sum(int(i.numel()) for i in ids_pool[0])
    seg = sum(n_seg)
    u = sum(int(r[2].item()) for r in res)
    nbytes = n * 8 + seg * 4 * dim + (0 if splits is None else 4 * (seg + cols))
    if emit:
      nbytes += u * (4 * dim + 8)
    if lr != 0.0:
      nbytes += u * 2 * 4 * dim
    wall_ms = el / steps * 1e3
    out[name] = {'ms': round(wall_ms, 5), 'event_ms': round(ms / steps, 5), 'ids': n,
                 'distinct_rows': u, 'bytes': nbytes,
                 'frac': round(nbytes / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del objs, grads

  flat = [batches[b] for b in range(pool)]
  one = [batch] * cols
  run('bwd_emit', flat, None, one, 'sum', 0.0, True)
  run('bwd_sgd', flat, None, one, 'sum', 1e-4, True)
  run('bwd_step_only', flat, None, one, 'sum', 1e-4, False)
  # ragged: Poisson(8) clipped to [0, 32] ids per segment (the splits are shared by the batches)
  rng = np.random.RandomState(4242)
  splits, counts = [], []
  for c in range(cols):
    lens = rng.poisson(8, size=batch).clip(0, 32)
    sp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    splits.append(torch.from_numpy(sp).to(device))
    counts.append(int(sp[-1]))
  ragged = []
  for b in range(min(pool, 4)):
    g = torch.Generator(device=device)
    g.manual_seed(9000 + b)
    ragged.append([torch.randint(0, 1 << 40, (counts[c],), device=device, dtype=torch.int64,
                                 generator=g) for c in range(cols)])
  run('bwd_ragged', ragged, splits, one, 'mean', 0.0, True)
  # the reproducible mode (option bwd_deterministic = 1: sums in id order, bit-equal to the oracle's
  # in-order fp32 sum, rows ascending) on the same two shapes: what exactness costs
  run('bwd_emit_det', flat, None, one, 'sum', 0.0, True, deterministic=1)
  run('bwd_ragged_det', ragged, splits, one, 'mean', 0.0, True, deterministic=1)
  out['steps'] = steps
  out['warmup'] = warmup
  out['id_batches'] = pool
  out['bytes_formula'] = ('n*8 + S*4D + [emit] u*(4D+8) + [step] u*2*4D (+ (S+1)*4 per ragged '
                          'column); SURVEY 8(d) bytes_bwd, u = n_unique of the call')
  return out


def load_traffic(config_key):
  """HBM bytes per launch from the committed PMC summary (profiles/hbm_traffic.json),
  if one exists for this exact workload; else None."""
  path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
  if not os.path.exists(path):
    return None, None
  try:
    data = json.load(open(path))
  except (OSError, ValueError):
    return None, None
  entry = data.get(config_key, {})
  return entry.get('hbm_bytes_per_launch'), entry.get('measured_in')


def rccl_versions():
  """RCCL the library was built against / the one the process runs (torch may load its own)."""
  import ctypes
  from hybridbackend_amd import _lib
  built, runtime = ctypes.c_int32(), ctypes.c_int32()
  _lib.check(_lib.lib().hbk_comm_rccl_versions(ctypes.byref(built), ctypes.byref(runtime)))
  return {'built': built.value, 'runtime': runtime.value}


def main():
  args = parse_args()
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(launch_ranks(args))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
  use_dist = world > 1 or ('RANK' in os.environ and args.sharded)
  dist = None
  watchdog = None
  headline = {'done': False, 'line': None}   # (read by the watchdog)
  if use_dist:
    # a rank that dies leaves its peers waiting inside a collective: never hang the node
    def _abort():
      sys.stderr.write(f'bench.py: rank {rank} gave up after {args.watchdog} s (a peer is gone '
                       'or a collective hangs)\n')
      sys.stderr.flush()
      if headline['done']:
        # the K timed steps are measured; what hangs is one of the reference measurements behind
        # them: the line goes out without those
        if headline['line'] is not None:
          headline['line']['config']['secondary_error'] = (
              f'watchdog: no answer within {args.watchdog} s during the measurements behind the '
              'headline steps')
          print(json.dumps(headline['line']), flush=True)
        os._exit(0)
      os._exit(3)
    watchdog = threading.Timer(args.watchdog, _abort)
    watchdog.daemon = True
    watchdog.start()
    import torch.distributed as dist   # pylint: disable=redefined-outer-name
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # host-side rendezvous only (barrier, max over ranks, RCCL-id broadcast): gloo, so that the
    # library's RCCL communicator is the only one in the process
    dist.init_process_group('gloo', rank=rank, world_size=world)
  if args.dry_run:
    # launcher / rendezvous check without a GPU: the same barrier-bracketed timing skeleton
    if use_dist:
      obj = [os.urandom(128) if rank == 0 else None]
      dist.broadcast_object_list(obj, src=0)
      assert len(obj[0]) == 128
      dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    el = time.perf_counter() - t0
    if use_dist:
      dist.barrier()
      t = torch.tensor([el], dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      el = float(t.item())
    if rank == 0:
      line = json.loads(error_line(args, 'dry run: launcher and rendezvous only, nothing measured'))
      line['dry_run'] = True
      line['ranks'] = world
      # the keys a measured N > 1 line carries next to the headline (nothing measured here)
      line['config'], line['xgmi'] = multi_gpu_skeleton()
      line['max_rank_sleep_ms'] = round(el * 1e3, 2)
      print(json.dumps(line), flush=True)
    if use_dist:
      watchdog.cancel()
      dist.destroy_process_group()
    return
  if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
    if rank == 0:
      print(error_line(args, 'no GPU visible for this rank: the HIP path is the only path'),
            flush=True)
    sys.exit(2)
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)

  import hybridbackend_amd as hb
  from hybridbackend_amd import _lib
  _lib.lib()

  n_batches = args.id_batches or min(args.steps + args.warmup, 64)
  n_batches = max(1, n_batches)
  tables = make_tables(args, device, rank, world)
  batches = make_id_batches(args, device, rank, n_batches)
  lookups_per_step_per_rank = args.columns * args.batch

  if world == 1 and not args.sharded:
    # one pre-bound descriptor set per id batch: a step is a single C-ABI call
    outs = [torch.empty(args.batch, args.dim, device=device) for _ in range(args.columns)]
    plans = []
    for b in range(n_batches):
      gl = hb.embedding.GroupLookup(tables, buckets=[args.rows] * args.columns,
                                    combiners='sum')
      gl.bind(batches[b], None, outs)
      plans.append(gl)

    def step(i):
      plans[i % n_batches].launch()
    parallelism = 'single-gpu'
  else:
    coll = hb.distribute.Collective(world_size=world, rank=rank, local_size=world)
    hb.distribute.Collective.set_default(coll)
    # what RCCL itself says it connected (ncclCommCount): proof in the line that the exchanges of a
    # multi-GPU number ran through ONE communicator spanning all ranks
    from hybridbackend_amd import _lib as _hbk_lib
    rccl_ranks_seen = int(_hbk_lib.lib().hbk_comm_rccl_ranks(coll._handle))
    sharded = hb.embedding.ShardedGroupLookup(
      tables, coll, buckets=[args.rows] * args.columns, combiners='sum',
      wire_dtype=torch.float16 if args.wire == 'fp16' else None)

    sh_outs = [torch.empty(args.batch, args.dim, device=device) for _ in range(args.columns)]
    # the id batches are resident: marshal every step's arguments once, a step is one C-ABI call
    bound = [sharded.bind(batches[b], None, sh_outs) for b in range(n_batches)]

    def step(i):
      sharded.launch(bound[i % n_batches])
      if not args.no_prefetch:
        # the next batch is resident: partition it while this step's exchanges are on the wire
        sharded.prefetch(bound[(i + 1) % n_batches])
    parallelism = f'row-sharded id-mod-{world} (alltoallv ids + rows over RCCL/xGMI)'

  def barrier():
    if use_dist:
      dist.barrier()

  # The step forms of the sharded step (a hardware question: a cross-stream hop costs ~11 us on this
  # chip, an owner gather ~33 us, profiles/r02_hop_probe.txt): two column groups with the exchanges
  # on the communicator's stream beside the gathers (the default until round 4), one group on the
  # communicator's stream, the exchanges enqueued inline on the compute stream (no event hops; the
  # shipped default), and the p2p form (owners store rows straight into the requester's outputs).
  forms = {'pipelined_2_groups': (2, 0, False), 'one_group': (1, 0, False), 'inline': (0, 1, False)}
  if args.wire == 'fp32' and (args.p2p == 'on' or (args.p2p == 'auto' and world == 1)):
    forms['p2p'] = (0, 1, True)

  def set_form(name):
    """(Re)creates the plan in form `name`; False when the p2p form cannot be bound (a collective
    answer: the same on every rank)."""
    from hybridbackend_amd import _lib as _hbk
    g, inline, p2p = forms[name]
    _hbk.set_option('sharded_groups', g)
    _hbk.set_option('sharded_inline', inline)
    sharded.close()                 # the options are read when the plan is (re)created
    return sharded.p2p_bind(sh_outs) if p2p else True

  def probe_forms():
    """A few untimed steps + args.tune_steps timed ones in every form, MAX over the ranks."""
    out = {}
    for name in forms:
      if not set_form(name):
        out[name] = None
        continue
      for i in range(3):
        step(i)
      torch.cuda.synchronize()
      barrier()
      t_probe = time.perf_counter()
      for i in range(args.tune_steps):
        step(3 + i)
      torch.cuda.synchronize()
      barrier()
      dt = time.perf_counter() - t_probe
      if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
      out[name] = round(dt / args.tune_steps * 1e3, 5)
    return out

  def probe_pipelined_steps_3():
    """Reference form: three plans over the one communicator, begin(step i + 1) before end(step i)
    (hb.embedding.PipelinedLookup): ids(i + 1) travel ahead of rows(i).  Outputs arrive one step
    late (forward-only use), so the timed steps never run in this form -- and it is measured BEHIND
    them: a first contact with a multi-GPU box that goes wrong here costs a reference figure, not
    the headline (the watchdog then prints the line without it)."""
    from hybridbackend_amd import _lib as _hbk
    old_groups = _hbk.get_option('sharded_groups')
    try:
      sharded.close()
      _hbk.set_option('sharded_groups', 1)
      plans3 = [hb.embedding.ShardedGroupLookup(
          tables, coll, buckets=[args.rows] * args.columns, combiners='sum',
          wire_dtype=torch.float16 if args.wire == 'fp16' else None) for _ in range(3)]
      pipe = hb.embedding.PipelinedLookup(plans3)
      outs3 = [sh_outs] + [[torch.empty_like(o) for o in sh_outs] for _ in range(2)]
      nb3 = min(n_batches, 8)
      bounds3 = [[pipe.bind(k, batches[b], None, outs3[k]) for b in range(nb3)] for k in range(3)]

      def step3(i):
        k = pipe.next_plan()
        pipe.step(bounds3[k][i % nb3], prefetch=None if args.no_prefetch else bounds3[k][(i + 3) % nb3])
      for i in range(6):
        step3(i)
      pipe.flush()
      torch.cuda.synchronize()
      barrier()
      t_probe = time.perf_counter()
      n3 = max(args.tune_steps, 6)
      for i in range(n3):
        step3(6 + i)
      pipe.flush()
      torch.cuda.synchronize()
      barrier()
      dt = time.perf_counter() - t_probe
      if use_dist:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
      pipe.close()
      return round(dt / n3 * 1e3, 5), None
    except Exception as e:  # pylint: disable=broad-except
      # (the same code on every rank: the same error on every rank; a reference only)
      return None, f'{type(e).__name__}: {e}'[:200]
    finally:
      _hbk.set_option('sharded_groups', old_groups)

  def timed_steps(step_fn, steps, warmup):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides;
    returns (wall seconds, MAX over ranks; HIP-event milliseconds on the launch stream).

    At the driver's K = 20 the bracket itself is visible (tools/scratch/bench_tail.py,
    profiles/r04_bench_tail.txt): the 20 kernels run back to back in 1127-1136 us (rocprofv3:
    one 6 us gap) and the first one starts 15-17 us after the clock; but the
    torch.cuda.synchronize() behind them took 15-20 us in some processes, 52-85 in others and
    140-147 in a few -- with nothing left to wait for, and only when a timing event had been
    recorded behind the last launch (without the events: 57.0-57.7 us per step in 10 of 10
    processes; with them 58, 61 or 64).  Waiting on the launch stream first
    (hipStreamSynchronize) leaves the device-wide synchronize the contract asks for with 5 us of
    work in 10 of 10 processes: 57.5-57.9 us per step on the wall against 56.6 by events.  Also
    kept out of the bracket, because it is free: torch creates a HIP event at its first record(),
    so both events are recorded once beforehand; the garbage collector is held off; the clock
    stops when every rank's work is complete (synchronize + barrier) and the last synchronize
    follows it."""
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    for i in range(warmup):
      step_fn(i)
    torch.cuda.synchronize()
    gc_was_on = gc.isenabled()
    gc.disable()
    barrier()
    torch.cuda.synchronize()
    stamps = [] if os.environ.get('HBK_BENCH_STAMPS') else None   # (diagnostics, stderr only)
    # HBK_BENCH_PROBE (diagnostics of the bracket only, profiles/r04_bench_tail.txt): 'poll' /
    # 'evsync' await the closing event by query / synchronize instead, 'noevents' records none
    probe_mode = os.environ.get('HBK_BENCH_PROBE', '')
    if probe_mode != 'noevents':
      ev0.record()
    t0 = time.perf_counter()
    for i in range(steps):
      step_fn(warmup + i)
      if stamps is not None:
        stamps.append(time.perf_counter())
    if probe_mode != 'noevents':
      ev1.record()
    if probe_mode == 'evsync':
      ev1.synchronize()
    elif probe_mode == 'poll':
      while not ev1.query():
        pass
    elif probe_mode != 'noevents':
      torch.cuda.current_stream().synchronize()
    t_ready = time.perf_counter()
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    if stamps:
      sys.stderr.write('bench stamps (us after the clock): launches returned %s | launch stream '
                       'done %.1f | synchronize + barrier back %.1f\n' % (
                           ' '.join('%.0f' % ((t - t0) * 1e6) for t in stamps),
                           (t_ready - t0) * 1e6, el * 1e6))
    torch.cuda.synchronize()
    if gc_was_on:
      gc.enable()
    ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream (torch's current stream)
    if use_dist:
      t = torch.tensor([el], dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      el = float(t.item())
    return el, ms

  # The K timed steps FIRST, in the library's shipped default form (round 6): at N > 1 the link probe,
  # the form probe and every reference measurement run BEHIND them -- the first contact of this code
  # with a multi-GPU box may go wrong in any of those, and then costs a reference figure (the
  # watchdog prints the line that is already measured), not the line.  If the form probe finds a
  # form more than 3 % faster on the machine at hand, the K steps are timed again in that form and
  # the faster measurement becomes `value` (config.sharded_form names it; the shipped default's
  # figure stays in config.value_at_shipped_default_M_lookups_per_s).
  if world > 1 or args.sharded:
    set_form('inline')
  elapsed, gpu_ms = timed_steps(step, args.steps, args.warmup)
  probe, groups_probe = None, None
  best_form = 'inline' if (world > 1 or args.sharded) else None

  def headline_fields(elapsed, gpu_ms):
    total_lookups = lookups_per_step_per_rank * world * args.steps
    launch_s = gpu_ms / 1e3 / args.steps                 # avg duration of the dominant kernel
    achieved = lookups_per_step_per_rank * (8 + 4 * args.dim + 4 * args.dim) / launch_s / 1e9
    return (round(total_lookups / elapsed / 1e6, 3), round(elapsed / args.steps * 1e3, 5), launch_s,
            achieved)

  total_lookups = lookups_per_step_per_rank * world * args.steps
  value = total_lookups / elapsed / 1e6
  ms_per_step = elapsed / args.steps * 1e3
  result = None

  if rank == 0:
    bytes_per_lookup = 8 + 4 * args.dim + 4 * args.dim  # id + row read + output write
    launch_s = gpu_ms / 1e3 / args.steps                 # avg duration of the dominant kernel
    achieved = lookups_per_step_per_rank * bytes_per_lookup / launch_s / 1e9
    workload = (f'{args.columns} cols x {args.rows} rows x dim{args.dim} fp32, batch '
                f'{args.batch}/GPU, 1 id/sample, fused bucketize+gather+combiner')
    key = f'c{args.columns}_r{args.rows}_d{args.dim}_b{args.batch}_n{world}'
    # (the committed PMC figure is the unsharded forward kernel's: a sharded step has none)
    traffic, traffic_round = (load_traffic(key) if world == 1 and not args.sharded
                              else (None, None))
    result = {
      'metric': 'M-lookups/sec, 26-col Criteo-shape dim16, 1/2/4/8 GPUs; % HBM roofline',
      'value': round(value, 3), 'unit': 'M-lookups/sec', 'n_gpus': world,
      'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 5),
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': workload, 'global_batch': args.batch * world,
                 'parallelism': parallelism,
                 'wire': args.wire if (world > 1 or args.sharded) else None,
                 'prefetch_next_partition': (world > 1 or args.sharded) and not args.no_prefetch,
                 'id_batches_resident': n_batches,
                 # the form the timed steps ran in, picked on this machine by the probe below;
                 # 'inline' is the library's shipped default since round 5
                 # (profiles/r05_overlap_model.txt): its probe time is the figure comparable with
                 # runs that do not tune (--tune-steps 0)
                 'rccl_ranks_seen': (rccl_ranks_seen if (world > 1 or args.sharded) else None),
                 'sharded_form': best_form,
                 'sharded_form_probe_ms_per_step': None,      # (filled behind the headline steps)
                 'value_at_shipped_default_M_lookups_per_s': (
                     round(value, 3) if (world > 1 or args.sharded) else None),
                 **{key: None for key in SECONDARY_KEYS}},
      'roofline': {
        'bound': 'hbm',
        'kernel': ('group_lookup_fwd_kernel' if world == 1 and not args.sharded
                   else 'sharded step (all kernels + exchanges)'),
        'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 4),
        'traffic': traffic,
        # the round whose --pmc passes produced the figure (tools/hbm_traffic.sh regenerates it)
        'traffic_measured_in': traffic_round,
        # what the memory system moved per second (PMC bytes / launch time) and how that stands
        # against the measured-achievable 6.29 TB/s of the guide: the dim-16 gather is bound by
        # the rate of random 64-byte row requests, not by bytes
        'memory_side_GBps': (round(traffic / launch_s / 1e9, 1) if traffic else None),
        'achievable_frac': (round(traffic / launch_s / 1e9 / HBM_ACHIEVABLE_GBS, 4)
                            if traffic else None),
        'algorithmic_bytes_per_lookup': bytes_per_lookup,
        'avg_launch_us': round(launch_s * 1e6, 3)},
    }
    if world > 1:
      # what each rank puts on its xGMI links per step (uniform ids: (W-1)/W of everything
      # leaves the GPU): int32 ids out, fp32|fp16 rows back; one link per peer
      id_b = 4 if args.rows <= 0x7fffffff else 8
      row_b = args.dim * (2 if args.wire == 'fp16' else 4)
      off = lookups_per_step_per_rank * (world - 1) / world
      link_bytes = off * (id_b + row_b)
      result['xgmi'] = {
        'bytes_out_per_rank_per_step': int(link_bytes), 'links_per_rank': world - 1,
        'achieved_GBps_per_rank_each_way': round(link_bytes / (elapsed / args.steps) / 1e9, 2),
        'link_probe': None, 'link_bound': None,   # (measured behind the headline steps)
        'rccl': rccl_versions(),
        'note': 'the sharded step is link-bound (DESIGN.md 5): one xGMI link per peer pair'}
    if world == 1 and not args.sharded and not args.no_secondary:
      # the backward family through the same bracket (after the headline: the SGD cases step the tables)
      try:
        result['config']['secondary_steps'] = backward_secondary(args, hb, tables, batches, device,
                                                                 timed_steps)
      except Exception as e:  # pylint: disable=broad-except
        result['config']['secondary_error'] = f'{type(e).__name__}: {e}'[:300]
    if world == 1 and not args.sharded and args.cpu_seconds > 0:
      result['cpu_baseline'] = cpu_baseline(args, tables, batches[0], args.cpu_seconds)
    else:
      result['cpu_baseline'] = None
    headline['line'] = result
  headline['done'] = True
  if use_dist:
    # from here on a hang costs the probes and reference measurements only, and not 15 minutes
    watchdog.cancel()
    watchdog = threading.Timer(min(args.watchdog, 300.0), _abort)
    watchdog.daemon = True
    watchdog.start()


  # ---- behind the headline steps (a hang from here on costs what follows, not the line) ----------
  if world > 1 and args.link_probe_mb > 0:
    try:
      probe = link_probe(coll, device, world, args.link_probe_mb, dist)
    except Exception as e:  # pylint: disable=broad-except
      # (the same error on every rank)
      probe = {'error': f'{type(e).__name__}: {e}'[:300]}
    if rank == 0:
      result['xgmi']['link_probe'] = probe
      # what those links allow at best, to read the value against
      result['xgmi']['link_bound'] = link_bound(args, world, probe, lookups_per_step_per_rank)
  if (world > 1 or args.sharded) and args.tune_steps > 0:
    try:
      groups_probe = probe_forms()
      fastest = min((k for k, v in groups_probe.items() if v is not None), key=groups_probe.get)
      groups_probe['pipelined_steps_3'] = None   # (measured last, see probe_pipelined_steps_3)
      if rank == 0:
        result['config']['sharded_form_probe_ms_per_step'] = groups_probe
      # (the probe's times are MAX over the ranks: every rank takes the same decision)
      if fastest != 'inline' and groups_probe[fastest] < 0.97 * groups_probe['inline'] and set_form(fastest):
        el2, ms2 = timed_steps(step, args.steps, args.warmup)
        if el2 < elapsed:
          best_form = fastest
          if rank == 0:
            v2, ms_step2, launch2, achieved2 = headline_fields(el2, ms2)
            result['value'], result['ms_per_step'] = v2, ms_step2
            result['config']['sharded_form'] = fastest
            result['roofline']['achieved'] = round(achieved2, 2)
            result['roofline']['frac'] = round(achieved2 / HBM_PEAK_GBS, 4)
            result['roofline']['avg_launch_us'] = round(launch2 * 1e6, 3)
            if world > 1:
              result['xgmi']['achieved_GBps_per_rank_each_way'] = round(
                  result['xgmi']['bytes_out_per_rank_per_step'] / (el2 / args.steps) / 1e9, 2)
        else:
          set_form('inline')
      else:
        set_form('inline')
    except Exception as e:  # pylint: disable=broad-except
      # (the same code on every rank: the same error on every rank; the line is measured)
      if rank == 0:
        result['config']['sharded_form_probe_error'] = f'{type(e).__name__}: {e}'[:300]

  # Reference measurements next to the sharded headline, same run, same batches:
  #  * the OTHER wire format of the embedding exchange (fp16 when the headline is fp32: the
  #    link-bound lever; the reference's comm_wire_dtype, collective.py:291-296);
  #  * REPLICATED: every rank holds all tables and looks its own batch up, no exchange -- what
  #    the reference does for tables that fit (variables.py:93-98) and the ceiling a sharded step
  #    can be compared with.
  if (world > 1 or args.sharded) and not args.no_secondary:
    try:
      sec_steps, sec_warm = max(1, min(args.steps, 20)), min(args.warmup, 5)
      other = 'fp16' if args.wire == 'fp32' else 'fp32'
      sharded.close()                 # the wire format is fixed when the plan is created
      sharded.wire_dtype = torch.float16 if other == 'fp16' else None
      el2, _ = timed_steps(step, sec_steps, sec_warm)
      sharded.close()
      sharded.wire_dtype = torch.float16 if args.wire == 'fp16' else None
      full = tables if world == 1 else make_tables(args, device, 0, 1)
      r_plans = []
      for b in range(n_batches):
        gl = hb.embedding.GroupLookup(full, buckets=[args.rows] * args.columns, combiners='sum')
        gl.bind(batches[b], None, sh_outs)
        r_plans.append(gl)
      el3, _ = timed_steps(lambda i: r_plans[i % n_batches].launch(), sec_steps, sec_warm)
      per_step = lookups_per_step_per_rank * world
      secondary = {
        'replicated_M_lookups_per_s': round(per_step * sec_steps / el3 / 1e6, 3),
        'replicated_ms_per_step': round(el3 / sec_steps * 1e3, 5),
        'other_wire': other,
        'other_wire_M_lookups_per_s': round(per_step * sec_steps / el2 / 1e6, 3),
        'other_wire_ms_per_step': round(el2 / sec_steps * 1e3, 5),
        'secondary_steps': {'steps': sec_steps, 'warmup': sec_warm}}
      del r_plans, full
      if groups_probe is not None:
        ms3, err3 = probe_pipelined_steps_3()
        groups_probe['pipelined_steps_3'] = ms3
        if err3:
          groups_probe['pipelined_steps_3_error'] = err3
    except Exception as e:  # pylint: disable=broad-except
      # (every rank runs the same code on the same shapes: an error here is the same error on
      # every rank; the headline steps are measured and their line goes out regardless)
      secondary = {key: None for key in SECONDARY_KEYS}
      secondary['secondary_error'] = f'{type(e).__name__}: {e}'[:300]
    if rank == 0:
      result['config'].update(secondary)

  if rank == 0:
    print(json.dumps(result), flush=True)
    headline['line'] = None        # (printed: a watchdog firing during the teardown adds nothing)

  if world > 1 or args.sharded:
    sharded.close()
    coll.close()
  if use_dist:
    watchdog.cancel()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
