// libhbk_testing.so -- TEST SUPPORT, not part of the product library.
//
// An in-process "world": `world_size` ranks living in ONE process (one host thread and one stream
// each, all on the current GPU) exchange through device copies.  It plugs into libhbk_core.so
// through the public custom-transport hook (hbk_comm_create_custom, include/hbk.h), so the
// multi-rank driver (hbk_sharded_lookup_fwd/_bwd) and every collective entry point run their
// production code -- chunk / offset arithmetic included -- on a single-GPU machine; only the
// wire is different.  Built by tests/support/Makefile into hybridbackend_amd/lib/.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "hbk.h"

namespace {

struct LocalWorld {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<const void*> ptr;          // [world] published send pointer
  std::vector<std::vector<int64_t>> off; // [world][n] element offset of the chunk for peer k
  std::vector<std::vector<int64_t>> len; // [world][n] elements for peer k
  std::vector<hipEvent_t> ready, done;   // [world]
  // artificial wire (hbk_testing_set_wire): every exchange is held, on its stream, by a host
  // function that waits  latency + (largest message this rank receives from a PEER) x scale / rate  -- one
  // link per peer pair, all links in parallel, as over xGMI.  With `count_self` the rank's own
  // chunk counts as a peer message (a world of ONE rank then models what the same step would put
  // on a link).  0 = off.
  double wire_bytes_per_us = 0.0, wire_latency_us = 0.0, wire_scale = 1.0;
  bool wire_count_self = false;
  // the wait runs BESIDE the copies of the exchange (a link transfer IS the copy, it does not follow
  // it): per rank a side stream that carries only the host functions, and the two events that fork
  // it off the exchange's stream and join it again.  exchange = max(copies, modelled time)
  std::vector<hipStream_t> wire_stream;  // [world], created by hbk_testing_set_wire
  std::vector<hipEvent_t> wire_fork, wire_join;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct RankCtx {
  LocalWorld* w;
  int rank;
};

// The wire: a HOST function in stream order (hipLaunchHostFunc) that waits for the modelled time:
// the stream the exchange runs on is held back, every other stream -- and the whole GPU -- goes on,
// which is what a link transfer looks like to the chip.  (A first version spun ONE WAVE on the
// device clock instead: with it no form of the step overlapped anything and more hardware queues
// made every form erratic -- a spinning kernel is not a good stand-in for an idle link.)
void wire_wait(void* us_bits) {
  const double us = (double)(uintptr_t)us_bits / 16.0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {
  }
}

#define LW_HIP(expr)                      \
  do {                                    \
    if ((expr) != hipSuccess) return 13;  \
  } while (0)

// Every rank publishes (pointer, per-peer offsets and lengths, indexed by position in `ranks`;
// every member of a group has the same list), then copies its chunk out of every peer's buffer
// on its own stream.  All world ranks take part in the barriers (a collective is called by every
// rank, whatever its group).
int lw_exchange(void* ctx_, int32_t rank, const int32_t* ranks, int32_t n_ranks, const void* sendbuf,
                const int64_t* send_off, const int64_t* send_len, void* recvbuf,
                const int64_t* recv_off, size_t esize, int32_t skip_self, hbk_stream_t stream_) {
  RankCtx* ctx = reinterpret_cast<RankCtx*>(ctx_);
  LocalWorld* w = ctx->w;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  const int me = rank;
  int k_me = -1;
  for (int k = 0; k < n_ranks; ++k) {
    if (ranks[k] == me) k_me = k;
  }
  if (k_me < 0) return 13;
  w->ptr[me] = sendbuf;
  w->off[me].assign(send_off, send_off + n_ranks);
  w->len[me].assign(send_len, send_len + n_ranks);
  LW_HIP(hipEventRecord(w->ready[me], stream));
  w->barrier();
  int64_t largest = 0;   // elements of the largest message this rank receives over a "link"
  for (int k = 0; k < n_ranks; ++k) {
    const int64_t n = w->len[ranks[k]][k_me];
    if ((ranks[k] != me || w->wire_count_self) && n > largest) largest = n;
  }
  // (control messages -- the size exchange, tokens: under 64 KB -- are not modelled: HIP runs the host
  // functions of ALL streams on one thread, so a 3 us wait for the sizes would queue behind the
  // rows' 270 us and hold the host, which waits for the sizes, for a whole wire time)
  const bool wired = w->wire_bytes_per_us > 0.0 && !w->wire_stream.empty() &&
                     (double)largest * (double)esize * w->wire_scale >= 65536.0;
  for (int k = 0; k < n_ranks; ++k) LW_HIP(hipStreamWaitEvent(stream, w->ready[ranks[k]], 0));
  if (wired) {
    const double us = w->wire_latency_us +
                      (double)largest * (double)esize * w->wire_scale / w->wire_bytes_per_us;
    LW_HIP(hipEventRecord(w->wire_fork[me], stream));
    LW_HIP(hipStreamWaitEvent(w->wire_stream[me], w->wire_fork[me], 0));
    LW_HIP(hipLaunchHostFunc(w->wire_stream[me], wire_wait, (void*)(uintptr_t)(us * 16.0)));
    LW_HIP(hipEventRecord(w->wire_join[me], w->wire_stream[me]));
  }
  for (int k = 0; k < n_ranks; ++k) {
    const int peer = ranks[k];
    const int64_t n = w->len[peer][k_me];
    if (n > 0 && !(skip_self && peer == me)) {
      LW_HIP(hipMemcpyAsync(reinterpret_cast<char*>(recvbuf) + (size_t)recv_off[k] * esize,
                            reinterpret_cast<const char*>(w->ptr[peer]) +
                                (size_t)w->off[peer][k_me] * esize,
                            (size_t)n * esize, hipMemcpyDeviceToDevice, stream));
    }
  }
  if (wired) LW_HIP(hipStreamWaitEvent(stream, w->wire_join[me], 0));
  LW_HIP(hipEventRecord(w->done[me], stream));
  w->barrier();
  // nobody may reuse its send buffer before every peer has copied out of it
  for (int k = 0; k < n_ranks; ++k) LW_HIP(hipStreamWaitEvent(stream, w->done[ranks[k]], 0));
  w->barrier();
  return 0;
}

struct PeerPtrs {
  const void* p[64];
  int32_t world;
  int32_t op;
};

template <typename T>
__device__ inline T red_op(T a, T b, int op) {
  switch (op) {
    case 1: return a * b;
    case 2: return a > b ? a : b;
    case 3: return a < b ? a : b;
    default: return a + b;
  }
}

// out[i] = op over ranks (rank order) of the published buffers
template <typename T>
__global__ __launch_bounds__(256) void local_reduce_kernel(const PeerPtrs pp, int64_t n, T* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T acc = reinterpret_cast<const T*>(pp.p[0])[i];
  for (int r = 1; r < pp.world; ++r) acc = red_op<T>(acc, reinterpret_cast<const T*>(pp.p[r])[i], pp.op);
  out[i] = acc;
}

int lw_allreduce(void* ctx_, int32_t rank, int32_t world_size, int32_t dtype, int32_t reduce_op,
                 const void* in, void* out, int64_t count, hbk_stream_t stream_) {
  RankCtx* ctx = reinterpret_cast<RankCtx*>(ctx_);
  LocalWorld* w = ctx->w;
  hipStream_t rs = reinterpret_cast<hipStream_t>(stream_);
  const int me = rank;
  if (world_size != w->world || w->world > 64) return 3;
  size_t esize = 0;
  switch (dtype) {
    case HBK_FLOAT: case HBK_INT32: esize = 4; break;
    case HBK_INT64: case HBK_DOUBLE: esize = 8; break;
    default: return 3;   // the in-process world reduces float, double, int32, int64
  }
  // partial results go to a private buffer so that peers still read this rank's INPUT
  void* tmp = nullptr;
  LW_HIP(hipMalloc(&tmp, (size_t)count * esize + 16));
  w->ptr[me] = in;
  LW_HIP(hipEventRecord(w->ready[me], rs));
  w->barrier();
  PeerPtrs pp;
  pp.world = w->world;
  pp.op = reduce_op;
  for (int i = 0; i < w->world; ++i) {
    LW_HIP(hipStreamWaitEvent(rs, w->ready[i], 0));
    pp.p[i] = w->ptr[i];
  }
  const unsigned blocks = (unsigned)((count + 255) / 256);
  switch (dtype) {
    case HBK_FLOAT:
      hipLaunchKernelGGL(local_reduce_kernel<float>, dim3(blocks), dim3(256), 0, rs, pp, count,
                         reinterpret_cast<float*>(tmp));
      break;
    case HBK_INT32:
      hipLaunchKernelGGL(local_reduce_kernel<int32_t>, dim3(blocks), dim3(256), 0, rs, pp, count,
                         reinterpret_cast<int32_t*>(tmp));
      break;
    case HBK_INT64:
      hipLaunchKernelGGL(local_reduce_kernel<int64_t>, dim3(blocks), dim3(256), 0, rs, pp, count,
                         reinterpret_cast<int64_t*>(tmp));
      break;
    default:
      hipLaunchKernelGGL(local_reduce_kernel<double>, dim3(blocks), dim3(256), 0, rs, pp, count,
                         reinterpret_cast<double*>(tmp));
      break;
  }
  LW_HIP(hipEventRecord(w->done[me], rs));
  w->barrier();
  for (int i = 0; i < w->world; ++i) LW_HIP(hipStreamWaitEvent(rs, w->done[i], 0));
  LW_HIP(hipMemcpyAsync(out, tmp, (size_t)count * esize, hipMemcpyDeviceToDevice, rs));
  LW_HIP(hipStreamSynchronize(rs));   // test transport: tmp is freed right away
  (void)hipFree(tmp);
  w->barrier();
  return 0;
}

void lw_destroy(void* ctx_) { delete reinterpret_cast<RankCtx*>(ctx_); }

}  // namespace

extern "C" int hbk_testing_local_world_create(void** world, int32_t world_size) {
  if (world == nullptr || world_size < 1) return 3;
  LocalWorld* w = new LocalWorld();
  w->world = world_size;
  w->ptr.resize(world_size);
  w->off.resize(world_size);
  w->len.resize(world_size);
  w->ready.resize(world_size);
  w->done.resize(world_size);
  for (int i = 0; i < world_size; ++i) {
    LW_HIP(hipEventCreateWithFlags(&w->ready[i], hipEventDisableTiming));
    LW_HIP(hipEventCreateWithFlags(&w->done[i], hipEventDisableTiming));
  }
  *world = w;
  return 0;
}

// Artificial wire for every exchange of the world: `gb_per_s` per link and direction (0 = off),
// `latency_us` per exchange, message sizes multiplied by `scale` (a scaled-down batch stands for the
// full one), `count_self` != 0: the rank's own chunk counts as a peer message.
extern "C" int hbk_testing_set_wire(void* world, double gb_per_s, double latency_us, double scale,
                                    int32_t count_self) {
  LocalWorld* w = reinterpret_cast<LocalWorld*>(world);
  if (w == nullptr || gb_per_s < 0.0 || scale <= 0.0) return 3;
  w->wire_bytes_per_us = gb_per_s * 1e3;   // GB/s = 1e3 bytes per us
  w->wire_latency_us = latency_us;
  w->wire_scale = scale;
  w->wire_count_self = count_self != 0;
  if (gb_per_s > 0.0 && w->wire_stream.empty()) {
    w->wire_stream.resize(w->world);
    w->wire_fork.resize(w->world);
    w->wire_join.resize(w->world);
    for (int i = 0; i < w->world; ++i) {
      LW_HIP(hipStreamCreateWithFlags(&w->wire_stream[i], hipStreamNonBlocking));
      LW_HIP(hipEventCreateWithFlags(&w->wire_fork[i], hipEventDisableTiming));
      LW_HIP(hipEventCreateWithFlags(&w->wire_join[i], hipEventDisableTiming));
    }
  }
  return 0;
}

// the wire's wait alone, on `stream` (tools/scratch/overlap_micro.py: do other streams go on?)
extern "C" int hbk_testing_wire_wait(hbk_stream_t stream, double us) {
  LW_HIP(hipLaunchHostFunc(reinterpret_cast<hipStream_t>(stream), wire_wait,
                           (void*)(uintptr_t)(us * 16.0)));
  return 0;
}

extern "C" int hbk_testing_local_world_destroy(void* world) {
  LocalWorld* w = reinterpret_cast<LocalWorld*>(world);
  if (w == nullptr) return 0;
  for (int i = 0; i < w->world; ++i) {
    (void)hipEventDestroy(w->ready[i]);
    (void)hipEventDestroy(w->done[i]);
  }
  for (size_t i = 0; i < w->wire_stream.size(); ++i) {
    (void)hipStreamSynchronize(w->wire_stream[i]);
    (void)hipStreamDestroy(w->wire_stream[i]);
    (void)hipEventDestroy(w->wire_fork[i]);
    (void)hipEventDestroy(w->wire_join[i]);
  }
  delete w;
  return 0;
}

// rank `rank` of the in-process world as an hbk communicator (local_size ranks per "node", so
// that INTRA_NODE / INTER_NODE exchanges can be exercised)
extern "C" int hbk_testing_comm_create(hbk_comm_t* comm, void* world, int32_t rank,
                                       int32_t local_size) {
  LocalWorld* w = reinterpret_cast<LocalWorld*>(world);
  if (comm == nullptr || w == nullptr) return 3;
  hbk_transport_t t;
  t.ctx = new RankCtx{w, rank};
  t.exchange = lw_exchange;
  t.allreduce = lw_allreduce;
  t.destroy = lw_destroy;
  const int rc = hbk_comm_create_custom(comm, &t, w->world, local_size, rank);
  if (rc != 0) lw_destroy(t.ctx);
  return rc;
}
