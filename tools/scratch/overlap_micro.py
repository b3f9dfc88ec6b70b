"""Does a stream held back by the modelled wire (a host function in stream order) let OTHER streams
run their kernels?  s_wire: wait 300 us.  s_k: a ~100 us kernel.  Together: ~300 us if they overlap."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hybridbackend_amd import _lib
t = _lib.testing_lib()
t.hbk_testing_wire_wait.restype = C.c_int
t.hbk_testing_wire_wait.argtypes = [C.c_void_p, C.c_double]
dev = torch.device('cuda:0')
x = torch.empty(1 << 28, device=dev)          # 1 GB: a fill takes ~200 us
s_wire, s_k, s_k2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e6

def wire(us=300.0):
  t.hbk_testing_wire_wait(C.c_void_p(s_wire.cuda_stream), us)
def kern():
  with torch.cuda.stream(s_k):
    x.fill_(1.0)
def both():
  wire(); kern()
def chain():
  # wire on s_wire, then an event; s_k: kernel, wait event, kernel
  wire()
  ev = torch.cuda.Event(); ev.record(s_wire)
  with torch.cuda.stream(s_k):
    x.fill_(1.0)
    s_k.wait_event(ev)
    x.fill_(2.0)
print('wire alone      %.1f us' % timed(wire))
print('kernel alone    %.1f us' % timed(kern))
print('wire + kernel   %.1f us   (overlap: ~max, serial: ~sum)' % timed(both))
print('wire -> event; kernel, wait, kernel  %.1f us   (overlap: wire + one kernel)' % timed(chain))
# does enqueueing behind a running host function block the HOST?
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
  wire(300.0)
t1 = time.perf_counter()
with torch.cuda.stream(s_k):
  x.fill_(3.0)
t2 = time.perf_counter()
ev = torch.cuda.Event(); ev.record(s_k)
t3 = time.perf_counter()
torch.cuda.synchronize()
t4 = time.perf_counter()
print('host: 5 x wire enqueue %.1f us, kernel launch on another stream %.1f us, event record %.1f us, drain %.1f us'
      % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6))
