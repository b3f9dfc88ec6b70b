// Round-5 probe: "the same build differs by 5-15 % between processes with where the tables land".
// ONE process allocates the tables of config 4 (25 x 1M + 1 x 100M rows, dim 128 = 64 GB) and of
// config 2 (26 x 1M rows, dim 16) under different allocation policies, and times the SAME
// hbk_group_lookup_fwd launch (uniform ids) on each.  Under rocprofv3 --pmc the per-dispatch rows
// of the counter file are cut into the policies by tools/prof_summary.py `chunks` (the probe
// prints the launch count per policy).
//
//   malloc      one hipMalloc per table, columns in order
//   malloc_rev  one hipMalloc per table, the 51 GB table first
//   slab        one hipMalloc for everything, tables carved at 2 MB-aligned offsets
//   frag        the free list is chopped first (4096 x 16 MB taken, every second one returned),
//               then `malloc`
//   vmm_1g      virtual range aligned to 1 GB, physical handles of 1 GB (hipMemCreate / hipMemMap)
//   vmm_2m      the same range backed by 2 MB handles (no fragment can be larger than 2 MB)
//
//   build: make -C tools bin/placement_probe     run: tools/bin/placement_probe [--quick] [cfg4|cfg2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/hbk.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);   \
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

__global__ void fill_ids(int64_t* ids, int64_t n, uint64_t rows, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i * 0x9e3779b97f4a7c15ull + seed;
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ull;
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ull;
  x ^= x >> 32;
  ids[i] = (int64_t)(x % rows);
}
// first touch of every 4 KB page (the timing must not contain page faults or zero-fill)
__global__ void touch(float* p, size_t n_floats) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 1024;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 1024; i < n_floats; i += stride) {
    p[i] = 1e-3f;
  }
}

struct Tables {
  std::vector<float*> tab;
  std::vector<void*> to_free;                       // hipMalloc'ed blocks
  std::vector<hipMemGenericAllocationHandle_t> handles;
  void* va = nullptr;
  size_t va_bytes = 0;
};

static const size_t kAlign = (size_t)2 << 20;
static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static bool alloc_vmm(Tables* t, const std::vector<size_t>& bytes, size_t chunk) {
  int dev = 0;
  CK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) !=
      hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  chunk = round_up(chunk, gran);
  size_t total = 0;
  std::vector<size_t> off;
  for (size_t b : bytes) {
    off.push_back(total);
    total += round_up(b, chunk);
  }
  if (hipMemAddressReserve(&t->va, total, (size_t)1 << 30, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  t->va_bytes = total;
  for (size_t o = 0; o < total; o += chunk) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    t->handles.push_back(h);
    CK(hipMemMap(static_cast<char*>(t->va) + o, chunk, 0, h, 0));
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(t->va, total, &acc, 1));
  for (size_t c = 0; c < bytes.size(); ++c) {
    t->tab.push_back(reinterpret_cast<float*>(static_cast<char*>(t->va) + off[c]));
  }
  return true;
}

static bool alloc_tables(const std::string& policy, const std::vector<size_t>& bytes, Tables* t) {
  const int n = (int)bytes.size();
  t->tab.assign(0, nullptr);
  if (policy == "malloc" || policy == "frag") {
    std::vector<void*> chop;
    if (policy == "frag") {
      for (int i = 0; i < 4096; ++i) {
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)16 << 20) != hipSuccess) {
          (void)hipGetLastError();
          break;
        }
        chop.push_back(p);
      }
      for (size_t i = 0; i < chop.size(); i += 2) CK(hipFree(chop[i]));   // holes of 16 MB
    }
    for (int c = 0; c < n; ++c) {
      void* p;
      CK(hipMalloc(&p, bytes[c]));
      t->tab.push_back(static_cast<float*>(p));
      t->to_free.push_back(p);
    }
    for (size_t i = 1; i < chop.size(); i += 2) t->to_free.push_back(chop[i]);
    return true;
  }
  if (policy == "malloc_rev") {
    t->tab.assign(n, nullptr);
    for (int c = n - 1; c >= 0; --c) {
      void* p;
      CK(hipMalloc(&p, bytes[c]));
      t->tab[c] = static_cast<float*>(p);
      t->to_free.push_back(p);
    }
    return true;
  }
  if (policy == "slab") {
    size_t total = 0;
    std::vector<size_t> off;
    for (size_t b : bytes) {
      off.push_back(total);
      total += round_up(b, kAlign);
    }
    void* p;
    CK(hipMalloc(&p, total));
    t->to_free.push_back(p);
    for (int c = 0; c < n; ++c) t->tab.push_back(reinterpret_cast<float*>(static_cast<char*>(p) + off[c]));
    return true;
  }
  if (policy == "vmm_1g") return alloc_vmm(t, bytes, (size_t)1 << 30);
  if (policy == "vmm_2m") return alloc_vmm(t, bytes, (size_t)2 << 20);
  return false;
}

static void free_tables(Tables* t) {
  for (void* p : t->to_free) CK(hipFree(p));
  if (t->va != nullptr) {
    CK(hipMemUnmap(t->va, t->va_bytes));
    for (auto h : t->handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(t->va, t->va_bytes));
  }
  *t = Tables();
}

int main(int argc, char** argv) {
  bool quick = false;
  std::string which = "both";
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--quick")) quick = true;
    else which = argv[i];
  }
  const int iters = quick ? 3 : 20, warm = quick ? 1 : 3;
  const int64_t B = 65536;
  const int n_cols = 26;
  const int kPool = 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Cfg {
    const char* name;
    int dim;
    std::vector<int64_t> rows;
  };
  std::vector<Cfg> cfgs;
  if (which == "both" || which == "cfg4") {
    Cfg c{"config 4 (dim 128, 25 x 1M + 100M rows)", 128, std::vector<int64_t>(26, 1000000)};
    c.rows[25] = 100000000;
    cfgs.push_back(c);
  }
  if (which == "both" || which == "cfg2") {
    cfgs.push_back(Cfg{"config 2 (dim 16, 26 x 1M rows)", 16, std::vector<int64_t>(26, 1000000)});
  }
  const char* policies[] = {"malloc", "slab", "vmm_1g", "malloc_rev", "frag", "vmm_2m",
                            "malloc", "slab", "vmm_1g"};
  for (const Cfg& cfg : cfgs) {
    printf("---- %s, batch %lld, uniform ids, %d timed launches after %d warm-ups per policy\n",
           cfg.name, (long long)B, iters, warm);
    std::vector<size_t> bytes;
    for (int64_t r : cfg.rows) bytes.push_back((size_t)r * cfg.dim * 4);
    // ids and outputs live through all policies (allocated first, never moved)
    std::vector<int64_t*> ids(kPool);
    for (int b = 0; b < kPool; ++b) {
      CK(hipMalloc(&ids[b], (size_t)n_cols * B * 8));
      for (int c = 0; c < n_cols; ++c) {
        hipLaunchKernelGGL(fill_ids, dim3((B + 255) / 256), dim3(256), 0, 0, ids[b] + (size_t)c * B,
                           B, (uint64_t)cfg.rows[c], (uint64_t)(b * 131 + c) * 0x1234567ull + 99);
      }
    }
    float* out;
    CK(hipMalloc(&out, (size_t)n_cols * B * cfg.dim * 4));
    CK(hipDeviceSynchronize());
    for (const char* pol : policies) {
      Tables t;
      if (!alloc_tables(pol, bytes, &t)) {
        printf("%-12s not available here\n", pol);
        free_tables(&t);
        continue;
      }
      for (int c = 0; c < n_cols; ++c) {
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, t.tab[c], bytes[c] / 4);
      }
      CK(hipDeviceSynchronize());
      std::vector<hbk_lookup_column_t> cols(n_cols);
      auto launch = [&](int i) {
        for (int c = 0; c < n_cols; ++c) {
          hbk_lookup_column_t& k = cols[c];
          memset(&k, 0, sizeof k);
          k.table = t.tab[c];
          k.rows = cfg.rows[c];
          k.dim = cfg.dim;
          k.ids_dtype = HBK_INT64;
          k.ids = ids[i % kPool] + (size_t)c * B;
          k.n_ids = B;
          k.n_segments = B;
          k.divisor = 1;
          k.combiner = HBK_COMBINER_SUM;
          k.out = out + (size_t)c * B * cfg.dim;
        }
        HB(hbk_group_lookup_fwd(n_cols, cols.data(), nullptr));
      };
      for (int i = 0; i < warm; ++i) launch(i);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) launch(i + warm);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-12s %8.2f us per launch   (big table at %p)\n", pol, ms * 1000.f / iters,
             (void*)t.tab[n_cols - 1]);
      fflush(stdout);
      free_tables(&t);
    }
    for (auto p : ids) CK(hipFree(p));
    CK(hipFree(out));
  }
  return 0;
}
