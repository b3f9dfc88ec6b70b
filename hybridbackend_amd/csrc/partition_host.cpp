// Host-memory twins of the partition entries: the CPU kernels of HbPartitionByModulo and
// HbPartitionByDualModuloStage{One,Two} (the reference registers DEVICE_CPU kernels for the non-N
// ops, hbtf/distribute/partition/partition_by_modulo_ops.cc:62-101 and
// partition_by_dual_modulo_ops.cc:62-130, so a graph may place them on the host).  Same arguments
// as the device entries minus workspace and stream; the buffers are host memory.  A stable
// counting sort per column: count per shard, exclusive prefix, place in input order.
#include <vector>

#include "common.h"

namespace hbk {
namespace {

// shard of one id: non-negative remainder for signed ids (C's % truncates towards zero)
template <typename T>
inline int64_t nonneg_mod(T v, int64_t m) {
  const int64_t r = (int64_t)(v % (T)m);
  return r < 0 ? r + m : r;
}
template <>
inline int64_t nonneg_mod<uint64_t>(uint64_t v, int64_t m) { return (int64_t)(v % (uint64_t)m); }
template <>
inline int64_t nonneg_mod<uint32_t>(uint32_t v, int64_t m) {
  return (int64_t)((uint64_t)v % (uint64_t)m);
}

template <typename T>
void partition_column(const T* in, int64_t len, int32_t P, int64_t modulus, int32_t stage, T* out,
                      int32_t* sizes, int32_t* indices) {
  std::vector<int32_t> shard((size_t)len);
  std::vector<int64_t> next((size_t)P, 0);
  for (int64_t i = 0; i < len; ++i) {
    int64_t s;
    if (stage == 0) {
      s = nonneg_mod<T>(in[i], P);
    } else {
      const int64_t pre = nonneg_mod<T>(in[i], (int64_t)P * modulus);
      s = stage == 1 ? pre % P : pre / modulus;
    }
    shard[(size_t)i] = (int32_t)s;
    ++next[(size_t)s];
  }
  int64_t run = 0;
  for (int32_t p = 0; p < P; ++p) {
    sizes[p] = (int32_t)next[(size_t)p];
    const int64_t n = next[(size_t)p];
    next[(size_t)p] = run;
    run += n;
  }
  for (int64_t i = 0; i < len; ++i) {
    const int64_t at = next[(size_t)shard[(size_t)i]]++;
    out[at] = in[i];
    indices[i] = (int32_t)at;
  }
}

int partition_host(const char* what, int32_t n_cols, int32_t dtype, int32_t P, int64_t modulus,
                   int32_t stage, const void* const* inputs, const int64_t* lens,
                   void* const* outputs, int32_t* const* sizes, int32_t* const* indices) {
  HBK_REQUIRE(n_cols >= 0, "%s: n_cols must be >= 0, got %d", what, n_cols);
  if (n_cols == 0) return HBK_OK;
  HBK_REQUIRE(inputs && lens && outputs && sizes && indices, "%s: NULL argument array", what);
  HBK_REQUIRE(dtype == HBK_INT32 || dtype == HBK_INT64 || dtype == HBK_UINT32 ||
                  dtype == HBK_UINT64,
              "%s: dtype must be int32, int64, uint32 or uint64", what);
  HBK_REQUIRE(P >= 1 && P <= 16384, "%s: num_partitions must be in [1, 16384], got %d", what, P);
  HBK_REQUIRE(stage == 0 || (modulus >= 1 && (int64_t)P * modulus < (1ll << 31)),
              "%s: modulus must be >= 1 and num_partitions * modulus < 2^31", what);
  for (int32_t c = 0; c < n_cols; ++c) {
    HBK_REQUIRE(lens[c] >= 0 && lens[c] < (1ll << 31),
                "%s: input %d must have fewer than 2^31 elements (rank-1 int32 indices), got %lld",
                what, c, (long long)lens[c]);
    HBK_REQUIRE(sizes[c] != nullptr, "%s: sizes[%d] is NULL", what, c);
    HBK_REQUIRE(lens[c] == 0 || (inputs[c] && outputs[c] && indices[c]),
                "%s: NULL buffer for input %d", what, c);
  }
  // (everything is checked before anything is written: an error leaves no column half done)
  for (int32_t c = 0; c < n_cols; ++c) {
    switch (dtype) {
      case HBK_INT32:
        partition_column<int32_t>((const int32_t*)inputs[c], lens[c], P, modulus, stage,
                                  (int32_t*)outputs[c], sizes[c], indices[c]);
        break;
      case HBK_UINT32:
        partition_column<uint32_t>((const uint32_t*)inputs[c], lens[c], P, modulus, stage,
                                   (uint32_t*)outputs[c], sizes[c], indices[c]);
        break;
      case HBK_INT64:
        partition_column<int64_t>((const int64_t*)inputs[c], lens[c], P, modulus, stage,
                                  (int64_t*)outputs[c], sizes[c], indices[c]);
        break;
      default:
        partition_column<uint64_t>((const uint64_t*)inputs[c], lens[c], P, modulus, stage,
                                   (uint64_t*)outputs[c], sizes[c], indices[c]);
        break;
    }
  }
  return HBK_OK;
}

}  // namespace
}  // namespace hbk

extern "C" int hbk_partition_by_modulo_host(int32_t n_cols, int32_t dtype, int32_t num_partitions,
                                            const void* const* inputs, const int64_t* lens,
                                            void* const* outputs, int32_t* const* sizes,
                                            int32_t* const* indices) {
  return hbk::partition_host("partition_by_modulo_host", n_cols, dtype, num_partitions, 1, 0,
                             inputs, lens, outputs, sizes, indices);
}

extern "C" int hbk_partition_by_dual_modulo_host(int32_t n_cols, int32_t dtype,
                                                 int32_t num_partitions, int32_t modulus,
                                                 int32_t stage, const void* const* inputs,
                                                 const int64_t* lens, void* const* outputs,
                                                 int32_t* const* sizes, int32_t* const* indices) {
  HBK_REQUIRE(stage == 1 || stage == 2, "partition_by_dual_modulo_host: stage must be 1 or 2, got %d",
              stage);
  return hbk::partition_host("partition_by_dual_modulo_host", n_cols, dtype, num_partitions,
                             modulus, stage, inputs, lens, outputs, sizes, indices);
}
