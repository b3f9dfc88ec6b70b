// C-ABI wall time of the ops around the gather, without any Python: stable partition (R2),
// unique (R7), backward duplicate-row reduction (R10).  HIP events around `iters` back-to-back
// calls on the null stream; inputs regenerated per call from a pool of resident id batches.
//   build: make -C tools        run: tools/bin/bench_ops
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <functional>
#include <random>
#include <vector>

#include "../include/hbk.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)
#define HB(x)                                                                      \
  do {                                                                             \
    int rc = (x);                                                                  \
    if (rc != HBK_OK) {                                                            \
      fprintf(stderr, "%s: %d %s\n", #x, rc, hbk_last_error());                    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

template <typename T>
static T* dev_random(size_t n, uint64_t mod) {
  std::vector<T> h(n);
  for (auto& v : h) v = (T)(rnd() % mod);
  T* d;
  CK(hipMalloc(&d, n * sizeof(T) + 16));
  CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

template <typename T>
static T* dev_alloc(size_t n) {
  T* d;
  CK(hipMalloc(&d, n * sizeof(T) + 16));
  return d;
}

// host time per call of the last time_us (the calls return before the device is done: what the
// entry point costs its caller -- a step is host-bound when this exceeds the device time)
static double g_host_us = 0.0;
static float time_us(int iters, const std::function<void(int)>& f) {
  if (const char* it = getenv("HBK_BENCH_ITERS")) {   // (counter passes: a few calls are enough)
    if (atoi(it) > 0) iters = atoi(it);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  const auto h0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) f(i + 3);
  g_host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / iters;
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

template <typename T>
static void bench_partition(const char* what, int n_cols, int64_t len, int P, int dtype) {
  const int kPool = 4;
  std::vector<T*> pool(kPool);
  for (auto& p : pool) p = dev_random<T>((size_t)n_cols * len, (uint64_t)1 << 30);
  T* out = dev_alloc<T>((size_t)n_cols * len);
  int32_t* idx = dev_alloc<int32_t>((size_t)n_cols * len);
  int32_t* sizes = dev_alloc<int32_t>((size_t)n_cols * P);
  std::vector<int64_t> lens(n_cols, len);
  const size_t ws_bytes = hbk_partition_workspace_bytes(n_cols, lens.data(), P);
  char* ws = dev_alloc<char>(ws_bytes);
  std::vector<const void*> in(n_cols);
  std::vector<void*> o(n_cols);
  std::vector<int32_t*> s(n_cols), ix(n_cols);
  float us = time_us(50, [&](int i) {
    for (int c = 0; c < n_cols; ++c) {
      in[c] = pool[i % kPool] + (size_t)c * len;
      o[c] = out + (size_t)c * len;
      s[c] = sizes + (size_t)c * P;
      ix[c] = idx + (size_t)c * len;
    }
    HB(hbk_partition_by_modulo_n(n_cols, dtype, P, in.data(), lens.data(), o.data(), s.data(),
                                 ix.data(), ws, ws_bytes, nullptr));
  });
  const double ids = (double)n_cols * len;
  const double bytes = ids * (3.0 * sizeof(T) + 4);
  printf("%-66s %9.2f us  %8.1f M ids/s  %7.1f GB/s (%.3f of 8 TB/s)\n", what, us, ids / us,
         bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
  {  // probe build of the library (-DHBK_PART_STAMPS): constant-clock stamps of the one-pass waves
    typedef int (*trace_fn)(unsigned long long*, int);
    trace_fn fn = (trace_fn)dlsym(RTLD_DEFAULT, "hbk_debug_part_trace");
    if (fn != nullptr) {
      std::vector<unsigned long long> tr(8192 * 8);
      fn(nullptr, 1);
      HB(hbk_partition_by_modulo_n(n_cols, dtype, P, in.data(), lens.data(), o.data(), s.data(),
                                   ix.data(), ws, ws_bytes, nullptr));
      fn(tr.data(), 0);
      static const char* names[7] = {"descriptor", "ids arrive", "ranks", "publish + wait",
                                     "bases + stores issued", "stores land", "-"};
      double sum[7] = {0}, life = 0;
      unsigned long long t_min = ~0ull, t_max = 0, first_pub = ~0ull, last_pub = 0, last_start = 0;
      int nb = 0;
      for (int b = 0; b < 8192; ++b) {
        const unsigned long long* t = &tr[(size_t)b * 8];
        if (t[0] == 0 || t[7] == 0) continue;
        ++nb;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        life += (double)(t[7] - t[0]);
        t_min = t[0] < t_min ? t[0] : t_min;
        t_max = t[7] > t_max ? t[7] : t_max;
        last_start = t[0] > last_start ? t[0] : last_start;
        first_pub = t[3] < first_pub ? t[3] : first_pub;
        last_pub = t[3] > last_pub ? t[3] : last_pub;
      }
      if (nb > 0) {
        printf("   one-pass kernel: %d traced waves over %.2f us, mean life %.2f us; last wave starts at "
               "%.2f us, publishes from %.2f to %.2f us; per phase (us):", nb, (t_max - t_min) * 0.01,
               life / nb * 0.01, (last_start - t_min) * 0.01, (first_pub - t_min) * 0.01,
               (last_pub - t_min) * 0.01);
        for (int i = 0; i < 6; ++i) printf("  %s %.2f", names[i], sum[i] / nb * 0.01);
        printf("\n");
      }
    }
  }
}

static void bench_unique(int n_cols, int64_t len, uint64_t mod) {
  const int kPool = 4;
  std::vector<int64_t*> pool(kPool);
  for (auto& p : pool) p = dev_random<int64_t>((size_t)n_cols * len, mod);
  int64_t* uniq = dev_alloc<int64_t>((size_t)n_cols * len);
  int32_t* idx = dev_alloc<int32_t>((size_t)n_cols * len);
  int32_t* nu = dev_alloc<int32_t>(n_cols);
  std::vector<int64_t> lens(n_cols, len);
  const size_t ws_bytes = hbk_unique_workspace_bytes(n_cols, lens.data());
  char* ws = dev_alloc<char>(ws_bytes);
  std::vector<const int64_t*> in(n_cols);
  std::vector<int64_t*> u(n_cols);
  std::vector<int32_t*> ix(n_cols), n(n_cols);
  float us = time_us(50, [&](int i) {
    for (int c = 0; c < n_cols; ++c) {
      in[c] = pool[i % kPool] + (size_t)c * len;
      u[c] = uniq + (size_t)c * len;
      ix[c] = idx + (size_t)c * len;
      n[c] = nu + c;
    }
    HB(hbk_unique_n(n_cols, in.data(), lens.data(), u.data(), ix.data(), n.data(), ws, ws_bytes,
                    nullptr));
  });
  {  // probe build: per-workgroup stamps of the group / first kernels
    typedef int (*trace_fn)(unsigned long long*, int);
    trace_fn fn = (trace_fn)dlsym(RTLD_DEFAULT, "hbk_debug_uni_trace");
    for (int which = 0; fn != nullptr && which < 3; ++which) {
      std::vector<unsigned long long> tr(8192 * 8);
      fn(nullptr, which);
      HB(hbk_unique_n(n_cols, in.data(), lens.data(), u.data(), ix.data(), n.data(), ws, ws_bytes, nullptr));
      fn(tr.data(), which);
      double sum[7] = {0}, life = 0;
      unsigned long long t_min = ~0ull, t_max = 0, last_start = 0;
      int nb = 0;
      for (int b = 0; b < 8192; ++b) {
        const unsigned long long* t = &tr[(size_t)b * 8];
        if (t[0] == 0 || t[7] == 0) continue;
        ++nb;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        life += (double)(t[7] - t[0]);
        t_min = t[0] < t_min ? t[0] : t_min;
        t_max = t[7] > t_max ? t[7] : t_max;
        last_start = t[0] > last_start ? t[0] : last_start;
      }
      if (nb > 0) {
        printf("   unique %s kernel: %d traced workgroups over %.2f us, mean life %.2f us, last starts at %.2f us; phases (us):",
               which == 0 ? "group" : which == 1 ? "first" : "order", nb, (t_max - t_min) * 0.01, life / nb * 0.01, (last_start - t_min) * 0.01);
        for (int i = 0; i < 7; ++i) printf(" %.2f", sum[i] / nb * 0.01);
        printf("\n");
      }
    }
  }
  const double ids = (double)n_cols * len;
  char what[128];
  snprintf(what, sizeof(what), "unique_n %d x %lld int64 (ids uniform in [0, %llu))", n_cols,
           (long long)len, (unsigned long long)mod);
  printf("%-66s %9.2f us  %8.1f M ids/s\n", what, us, ids / us);
}

// H > 0: every sample holds H ids (row_splits 0, H, 2H, ...), combiner mean; n_ids = B
// H < 0: B / |H| samples of Poisson(|H|) ids clipped to [0, 4 |H|] (SURVEY 8d's multi-hot variant; the
//        case bench.py times as config.secondary_steps.bwd_ragged), one split array for all columns
static void bench_backward(int n_cols, int64_t B, int dim, int64_t rows, float lr,
                           bool step_only = false, int H = 0) {
  const int kPool = 4;
  const int aH = H < 0 ? -H : H;
  const int64_t n_seg = aH > 0 ? B / aH : B;
  int32_t* splits = nullptr;
  if (aH > 0) {
    std::vector<int32_t> hs((size_t)n_seg + 1);
    if (H > 0) {
      for (int64_t i = 0; i <= n_seg; ++i) hs[(size_t)i] = (int32_t)(i * H);
    } else {
      std::mt19937 gen(4242);
      std::poisson_distribution<int> pd((double)aH);
      hs[0] = 0;
      for (int64_t i = 0; i < n_seg; ++i) {
        int len = pd(gen);
        len = len > 4 * aH ? 4 * aH : len;
        hs[(size_t)i + 1] = hs[(size_t)i] + len;
      }
      B = hs[(size_t)n_seg];     // ids per column
    }
    splits = dev_alloc<int32_t>((size_t)n_seg + 1);
    CK(hipMemcpy(splits, hs.data(), ((size_t)n_seg + 1) * 4, hipMemcpyHostToDevice));
  }
  std::vector<int64_t*> pool(kPool);
  for (auto& p : pool) p = dev_random<int64_t>((size_t)n_cols * B, (uint64_t)1 << 40);
  std::vector<float*> tables(n_cols);
  for (auto& t : tables) {
    t = dev_alloc<float>((size_t)rows * dim);
    CK(hipMemset(t, 0, (size_t)rows * dim * 4));
  }
  float* gout = dev_alloc<float>((size_t)n_cols * B * dim);
  CK(hipMemset(gout, 0x3c, (size_t)n_cols * B * dim * 4));
  int64_t* urows = dev_alloc<int64_t>((size_t)n_cols * B);
  float* grows = dev_alloc<float>((size_t)n_cols * B * dim);
  int32_t* nu = dev_alloc<int32_t>(n_cols);
  std::vector<hbk_lookup_grad_column_t> cols(n_cols);
  auto fill = [&](int i) {
    for (int c = 0; c < n_cols; ++c) {
      hbk_lookup_grad_column_t& h = cols[c];
      h = hbk_lookup_grad_column_t();
      h.table = tables[c];
      h.rows = rows;
      h.dim = dim;
      h.ids_dtype = HBK_INT64;
      h.ids = pool[i % kPool] + (size_t)c * B;
      h.n_ids = B;
      h.n_segments = n_seg;
      h.row_splits = splits;
      h.bucket = rows;
      h.divisor = 1;
      h.combiner = aH > 0 ? HBK_COMBINER_MEAN : HBK_COMBINER_SUM;
      h.grad_out = gout + (size_t)c * n_seg * dim;
      h.unique_rows = step_only ? nullptr : urows + (size_t)c * B;
      h.grad_rows = step_only ? nullptr : grows + (size_t)c * B * dim;
      h.n_unique = nu + c;
    }
  };
  fill(0);
  const size_t ws_bytes = hbk_group_lookup_bwd_workspace_bytes(n_cols, cols.data());
  char* ws = dev_alloc<char>(ws_bytes);
  float us = time_us(30, [&](int i) {
    fill(i);
    HB(hbk_group_lookup_bwd(n_cols, cols.data(), lr, ws, ws_bytes, nullptr));
  });
  const double n = (double)n_cols * B;
  char what[128];
  snprintf(what, sizeof(what), "group_lookup_bwd %d x %lld ids%s, dim %d, %lld rows%s", n_cols,
           (long long)B, H > 0 ? " (ragged mean)" : H < 0 ? " (ragged Poisson lengths, mean)" : "", dim, (long long)rows,
           step_only ? ", SGD step only" : lr != 0.f ? " + SGD apply" : "");
  printf("%-66s %9.2f us  %8.1f M lookups/s   (host %.1f us per call)\n", what, us, n / us, g_host_us);
  {  // probe build: stamps of the grouping kernel's workgroups
    typedef int (*trace_fn)(unsigned long long*, int);
    trace_fn fn = (trace_fn)dlsym(RTLD_DEFAULT, "hbk_debug_grp_trace");
    if (fn != nullptr) {
      std::vector<unsigned long long> tr(8192 * 8);
      fn(nullptr, 1);
      fill(0);
      HB(hbk_group_lookup_bwd(n_cols, cols.data(), lr, ws, ws_bytes, nullptr));
      fn(tr.data(), 0);
      static const char* names[7] = {"column + clear + issue loads", "ids arrive + ranks", "publish",
                                     "wait + sums", "scan + offsets", "issue stores", "stores land"};
      double sum[7] = {0}, life = 0;
      unsigned long long t_min = ~0ull, t_max = 0, last_start = 0, last_pub = 0;
      int nb = 0;
      for (int b = 0; b < 8192; ++b) {
        const unsigned long long* t = &tr[(size_t)b * 8];
        if (t[0] == 0 || t[7] == 0) continue;
        ++nb;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        life += (double)(t[7] - t[0]);
        t_min = t[0] < t_min ? t[0] : t_min;
        t_max = t[7] > t_max ? t[7] : t_max;
        last_start = t[0] > last_start ? t[0] : last_start;
        last_pub = t[3] > last_pub ? t[3] : last_pub;
      }
      if (nb > 0) {
        printf("   group kernel: %d traced workgroups over %.2f us, mean life %.2f us; last starts at %.2f us, "
               "last publish at %.2f us; per phase (us):", nb, (t_max - t_min) * 0.01, life / nb * 0.01,
               (last_start - t_min) * 0.01, (last_pub - t_min) * 0.01);
        for (int i = 0; i < 7; ++i) printf("  %s %.2f", names[i], sum[i] / nb * 0.01);
        printf("\n");
      }
    }
  }
  {  // probe build of the library (-DHBK_BWD_STAMPS): shader-clock stamps of the reduce workgroups
    typedef int (*trace_fn)(unsigned long long*, int);
    trace_fn fn = (trace_fn)dlsym(RTLD_DEFAULT, "hbk_debug_bwd_trace");
    if (fn != nullptr) {
      std::vector<unsigned long long> tr(8192 * 8);
      fn(nullptr, 1);
      fill(0);
      HB(hbk_group_lookup_bwd(n_cols, cols.data(), lr, ws, ws_bytes, nullptr));
      fn(tr.data(), 0);
      static const char* names_h[7] = {"setup (column, bstart)", "table init", "(a) pairs + insert",
                                       "(b) scan", "(c) gradient round", "(d,e) + chunk end", "exit"};
      // row-sorted buckets (HBK_STAMP_NAMES=rowsort): the phases of lookup_bwd_rowsort.h
      static const char* names_r[7] = {"job descriptor", "pairs + clear", "A bitmap + B ranks + claim",
                                       "C tickets", "D runs + E sorted", "F walk + G hot rows",
                                       "row numbers + exit"};
      const char* which = getenv("HBK_STAMP_NAMES");
      const char** names = which != nullptr && which[0] == 'r' ? names_r : names_h;
      // stamps are ticks of the 100 MHz constant clock (10 ns), the same clock on every CU
      double sum[7] = {0}, life = 0;
      unsigned long long t_min = ~0ull, t_max = 0;
      int n_blocks = 0;
      for (int b = 0; b < 8192; ++b) {
        const unsigned long long* t = &tr[(size_t)b * 8];
        if (t[0] == 0 || t[7] == 0) continue;
        bool ok = true;
        for (int i = 1; i < 8; ++i) ok = ok && t[i] >= t[i - 1];
        if (!ok) continue;
        ++n_blocks;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(t[i + 1] - t[i]);
        life += (double)(t[7] - t[0]);
        t_min = t[0] < t_min ? t[0] : t_min;
        t_max = t[7] > t_max ? t[7] : t_max;
      }
      {
        typedef int (*sub_fn)(unsigned long long*);
        sub_fn sf = (sub_fn)dlsym(RTLD_DEFAULT, "hbk_debug_bwd_sub");
        if (sf != nullptr) {
          std::vector<unsigned long long> sb(8192 * 4);
          sf(sb.data());
          double d[3] = {0, 0, 0};
          int m = 0;
          for (int b = 0; b < 8192; ++b) {
            const unsigned long long* q = &sb[(size_t)b * 4];
            if (q[0] == 0 || q[3] < q[0]) continue;
            ++m;
            for (int i = 0; i < 3; ++i) d[i] += (double)(q[i + 1] - q[i]);
          }
          if (m > 0) {
            printf("   inside (a), first chunk: pairs arrive %.2f us, search + insert (both pairs) %.2f, barrier %.2f\n",
                   d[0] / m * 0.01, d[1] / m * 0.01, d[2] / m * 0.01);
          }
        }
      }
      const double nb = n_blocks ? n_blocks : 1;
      printf("   reduce kernel: %d traced workgroups over %.1f us, mean life %.2f us (=> %.0f alive on "
             "average); per phase (us):", n_blocks, (t_max - t_min) * 0.01, life / nb * 0.01,
             life * 0.01 / ((t_max - t_min) * 0.01 + 1e-9));
      for (int i = 0; i < 7; ++i) printf("  %s %.2f", names[i], sum[i] / nb * 0.01);
      printf("\n");
      // how many workgroups are alive / past their setup at a few instants
      for (int q = 1; q <= 9; q += 2) {
        const unsigned long long at = t_min + (t_max - t_min) * q / 10;
        int alive = 0, working = 0, started = 0;
        for (int b = 0; b < 8192; ++b) {
          const unsigned long long* t = &tr[(size_t)b * 8];
          if (t[0] == 0 || t[7] == 0) continue;
          started += t[0] <= at;
          alive += t[0] <= at && at < t[7];
          working += t[1] <= at && at < t[7];
        }
        printf("     at %d0%% of the span: %d started, %d alive, %d past setup\n", q, started, alive, working);
      }
    }
  }
  for (auto& t : tables) CK(hipFree(t));
}

static void bench_streaming(int n_cols, int64_t len) {
  // R1 bucketize and R6 wire casts: pure streams, N tensors per launch
  int64_t* ids = dev_random<int64_t>((size_t)n_cols * len, (uint64_t)1 << 40);
  int64_t* out = dev_alloc<int64_t>((size_t)n_cols * len);
  std::vector<const void*> in(n_cols);
  std::vector<void*> o(n_cols);
  std::vector<int64_t> lens(n_cols, len), buckets(n_cols, 1000000);
  for (int c = 0; c < n_cols; ++c) {
    in[c] = ids + (size_t)c * len;
    o[c] = out + (size_t)c * len;
  }
  float us = time_us(50, [&](int) {
    HB(hbk_floormod_n(n_cols, HBK_INT64, in.data(), lens.data(), buckets.data(), o.data(), nullptr));
  });
  char what[128];
  const double n = (double)n_cols * len;
  snprintf(what, sizeof(what), "floormod_n %d x %lld int64 %% 1000000", n_cols, (long long)len);
  printf("%-66s %9.2f us  %8.1f M ids/s  %7.1f GB/s (%.3f of 8 TB/s)\n", what, us, n / us,
         n * 16 / us / 1e3, n * 16 / us / 1e3 / 8000.0);
  const int64_t fl = len * 16;  // one column's rows of a dim-16 exchange
  float* f32 = dev_alloc<float>((size_t)n_cols * fl);
  uint16_t* f16 = dev_alloc<uint16_t>((size_t)n_cols * fl);
  CK(hipMemset(f32, 0x3c, (size_t)n_cols * fl * 4));
  std::vector<int64_t> flens(n_cols, fl);
  for (int c = 0; c < n_cols; ++c) {
    in[c] = f32 + (size_t)c * fl;
    o[c] = f16 + (size_t)c * fl;
  }
  us = time_us(50, [&](int) {
    HB(hbk_cast_n(n_cols, HBK_FLOAT, HBK_HALF, in.data(), flens.data(), o.data(), nullptr));
  });
  const double nf = (double)n_cols * fl;
  snprintf(what, sizeof(what), "cast_n fp32 -> fp16, %d x %lld floats", n_cols, (long long)fl);
  printf("%-66s %9.2f us  %7.1f GB/s (%.3f of 8 TB/s)\n", what, us, nf * 6 / us / 1e3,
         nf * 6 / us / 1e3 / 8000.0);
}

static void bench_probe(int64_t n_keys, int64_t slab_count, int slab_size) {
  // every slab half full (odd slots EMPTY, so a probe ends in its first slab); half of the looked
  // up keys are present
  std::vector<int64_t> hc_(slab_count * slab_size), hk_(n_keys);
  for (size_t i = 0; i < hc_.size(); ++i) {
    hc_[i] = (i & 1) ? INT64_MIN : (int64_t)(rnd() % ((uint64_t)1 << 40));
  }
  for (int64_t i = 0; i < n_keys; ++i) {
    hk_[i] = (i & 1) ? (int64_t)(rnd() % ((uint64_t)1 << 40)) : hc_[(rnd() % (hc_.size() / 2)) * 2];
  }
  int64_t* cache = dev_alloc<int64_t>(hc_.size());
  int64_t* keys = dev_alloc<int64_t>(n_keys);
  CK(hipMemcpy(cache, hc_.data(), hc_.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(keys, hk_.data(), n_keys * 8, hipMemcpyHostToDevice));
  int32_t* hi = dev_alloc<int32_t>(n_keys);
  int64_t* hc = dev_alloc<int64_t>(n_keys);
  int32_t* mi = dev_alloc<int32_t>(n_keys);
  int64_t* mk = dev_alloc<int64_t>(n_keys);
  int32_t* counts = dev_alloc<int32_t>(2);
  const size_t ws_bytes = hbk_cache_lookup_workspace_bytes(n_keys);
  char* ws = dev_alloc<char>(ws_bytes);
  float us = time_us(50, [&](int) {
    HB(hbk_cache_lookup(cache, slab_count, slab_size, keys, n_keys, hi, hc, mi, mk, counts, ws,
                        ws_bytes, nullptr));
  });
  char what[128];
  snprintf(what, sizeof(what), "cache_lookup %lld keys, %lld slabs x %d (probe + 4 compacted lists)",
           (long long)n_keys, (long long)slab_count, slab_size);
  printf("%-66s %9.2f us  %8.1f M keys/s\n", what, us, (double)n_keys / us);
}

int main(int argc, char** argv) {
  printf("hbk %s -- C-ABI wall time per call (HIP events, no Python)\n", hbk_version());
  if (argc > 1 && argv[1][0] == 'b') {  // "bwd": only the config-2 backward (for --pmc passes)
    bench_backward(26, 65536, 16, 1000000, 0.f);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'p') {  // "part": the single-node partition only
    bench_partition<int64_t>("partition_by_modulo_n 26 x 65536 int64, P = 8", 26, 65536, 8, HBK_INT64);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'P') {  // the 64-way partition only
    bench_partition<int64_t>("partition_by_modulo_n 26 x 65536 int64, P = 64", 26, 65536, 64, HBK_INT64);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'c') {  // "cache": the slab-cache lookup only
    bench_probe(26 * 65536, 1 << 16, 32);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'R') {  // the sweep's ragged case at fixed lengths: 26 x 65536 segments of 8 ids, 1M rows
    // (bench_ops R 64: 64 ids per segment -- the same pairs and output rows over a gradient block of
    // 512 KB per column instead of 4 MB: what the reduce stage costs when its gradient reads hit L2)
    bench_backward(26, 524288, 16, 1000000, 0.f, false, argc > 2 ? atoi(argv[2]) : 8);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'Q') {  // the ragged case with Poisson(8) lengths clipped to [0, 32]: bench.py's bwd_ragged
    bench_backward(26, 524288, 16, 1000000, 0.f, false, -8);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'r') {  // "ragged": config-5-like columns (8 ids per sample, mean)
    bench_backward(16, 524288, 16, 100000, 0.01f, false, 8);
    bench_backward(16, 524288, 64, 100000, 0.01f, false, 8);
    bench_backward(16, 524288, 16, 10000000, 0.01f, false, 8);
    bench_backward(16, 524288, 4, 3000, 0.01f, false, 8);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'd') {  // "dup": duplicated ids (3, 33 and 330 pairs per row)
    bench_backward(26, 65536, 16, 20000, 0.f);
    bench_backward(26, 65536, 16, 2000, 0.f);
    bench_backward(26, 65536, 16, 200, 0.f);
    bench_backward(26, 65536, 64, 20000, 0.f);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'w') {  // "wide": dim-128 backward (config 4's 1M-row tables, uniform ids)
    bench_backward(26, 65536, 128, 1000000, 0.f);
    bench_backward(26, 65536, 128, 1000000, 0.01f);
    bench_backward(26, 65536, 128, 1000000, 0.01f, true);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'u') {  // "unique": the owner-side unique only
    bench_unique(26, 65536, 1000000);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'q') {  // the reference benchmark's shape only
    bench_partition<int32_t>("partition_by_modulo_n 100 x 100000 int32, P = 8 (reference benchmark)", 100,
                             100000, 8, HBK_INT32);
    return 0;
  }
  if (argc > 1 && argv[1][0] == 's') {  // "step": config-2 backward with the step, both modes
    bench_backward(26, 65536, 16, 1000000, 0.01f);
    bench_backward(26, 65536, 16, 1000000, 0.01f, true);
    return 0;
  }
  bench_partition<int64_t>("partition_by_modulo_n 26 x 65536 int64, P = 8", 26, 65536, 8, HBK_INT64);
  bench_partition<int64_t>("partition_by_modulo_n 26 x 65536 int64, P = 2", 26, 65536, 2, HBK_INT64);
  bench_partition<int64_t>("partition_by_modulo_n 26 x 65536 int64, P = 64", 26, 65536, 64, HBK_INT64);
  bench_partition<int32_t>("partition_by_modulo_n 100 x 100000 int32, P = 8 (reference benchmark)", 100,
                           100000, 8, HBK_INT32);
  bench_partition<int64_t>("partition_by_modulo_n 26 x 1048576 int64, P = 8", 26, 1048576, 8, HBK_INT64);
  bench_unique(26, 65536, 1000000);
  bench_unique(26, 65536, 4096);
  bench_backward(26, 65536, 16, 1000000, 0.f);
  bench_backward(26, 65536, 16, 1000000, 0.01f);
  bench_backward(26, 65536, 16, 1000000, 0.01f, true);
  bench_backward(26, 65536, 128, 1000000, 0.f);
  bench_backward(26, 65536, 128, 1000000, 0.01f);
  bench_backward(26, 65536, 128, 1000000, 0.01f, true);
  bench_streaming(26, 65536);
  bench_streaming(26, 1048576);
  bench_probe(26 * 65536, 1 << 16, 32);
  return 0;
}
