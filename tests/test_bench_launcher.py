"""bench.py's launch paths on CPU: `python bench.py --gpus N` must start its own ranks (the driver
runs exactly that form), the torch.distributed.run form must work too, and a node without enough
GPUs must answer with one JSON line carrying "error" and a non-zero exit code -- quickly."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _last_json(out):
  lines = [ln for ln in out.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, out
  return json.loads(lines[0])


def test_self_launch_two_ranks_dry_run():
  r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1',
                      '--dry-run'], capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert r.returncode == 0, r.stderr
  line = _last_json(r.stdout)
  assert line['dry_run'] is True and line['ranks'] == 2 and line['n_gpus'] == 2
  assert line['max_rank_sleep_ms'] >= 19.0          # the MAX over ranks (rank 1 sleeps 20 ms)
  assert line['value'] is None and 'error' in line   # nothing measured, and it says so
  # the reference measurements a real N > 1 line carries next to the sharded headline: every
  # rank holding all tables (replicated), and the other wire format -- named here, measured there
  for key in ('replicated_M_lookups_per_s', 'replicated_ms_per_step', 'other_wire',
              'other_wire_M_lookups_per_s', 'other_wire_ms_per_step', 'secondary_steps'):
    assert key in line['config'] and line['config'][key] is None


def test_dry_run_names_every_key_of_a_multi_gpu_line():
  """N = 2, 4, 8: the one line carries, by name, everything the first
  multi-GPU box has to report next to the headline -- per-form ms (inline / p2p / pipelined column
  groups / three plans pipelined across steps), rccl_ranks_seen, the link probe and the ceiling
  computed from it, the other wire format and the replicated reference."""
  for n in (2, 4, 8):
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(n), '--steps', '2', '--warmup', '1',
                        '--dry-run'], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    line = _last_json(r.stdout)
    assert line['ranks'] == n and line['n_gpus'] == n
    cfg = line['config']
    for key in ('rccl_ranks_seen', 'sharded_form', 'value_at_shipped_default_M_lookups_per_s',
                'replicated_M_lookups_per_s', 'replicated_ms_per_step', 'other_wire',
                'other_wire_M_lookups_per_s', 'other_wire_ms_per_step', 'secondary_steps', 'wire'):
      assert key in cfg, (n, key)
    assert set(cfg['sharded_form_probe_ms_per_step']) == {
        'pipelined_2_groups', 'one_group', 'inline', 'p2p', 'pipelined_steps_3'}
    for key in ('link_probe', 'link_bound', 'rccl', 'bytes_out_per_rank_per_step',
                'achieved_GBps_per_rank_each_way'):
      assert key in line['xgmi'], (n, key)


def test_the_measured_line_is_built_from_the_same_key_lists():
  """The real N > 1 path fills the keys the dry run names: both come from bench.py's FORM_KEYS /
  XGMI_KEYS / SECONDARY_KEYS, and the source mentions every one of them where it is measured."""
  sys.path.insert(0, ROOT)
  import bench
  src = open(BENCH).read()
  for key in bench.FORM_KEYS + bench.XGMI_KEYS + bench.SECONDARY_KEYS:
    assert src.count("'" + key + "'") >= 2, key    # once in the list, once where it is filled
  probe = {'GBps_per_link_each_way': 50.0}
  args = bench.parse_args.__globals__['argparse'].Namespace(rows=1000000, dim=16, wire='fp32')
  lb = bench.link_bound(args, 8, probe, 26 * 65536)
  # 1/8 of 1.7 M lookups x (4 + 64) B over one 50 GB/s link = 289.7 us
  assert abs(lb['us_per_step'] - 26 * 65536 / 8 * 68 / 50e9 * 1e6) < 0.1
  assert bench.link_bound(args, 8, None, 1) is None


def test_torchrun_form_two_ranks_dry_run():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                      '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
                      str(port), BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                     capture_output=True, text=True, timeout=300, cwd=ROOT)
  assert r.returncode == 0, r.stderr
  line = _last_json(r.stdout)
  assert line['dry_run'] is True and line['ranks'] == 2


def test_not_enough_gpus_is_a_fast_json_error():
  import torch
  if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
    import pytest
    pytest.skip('this node has 8 GPUs')
  r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '3', '--warmup', '1'],
                     capture_output=True, text=True, timeout=120, cwd=ROOT)
  assert r.returncode != 0
  line = _last_json(r.stdout)
  assert line['value'] is None and 'only' in line['error'] and line['n_gpus'] == 8
