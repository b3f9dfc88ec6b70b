// Error channel, version string and options of libhbk_core.so.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace hbk {
namespace {
thread_local char g_last_error[1024] = "";
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace hbk

namespace hbk {
namespace {
struct OptionEntry {
  const char* name;
  const char* env;
  int Options::*field;
};
const OptionEntry kOptions[] = {
    {"bwd_buckets_log2", "HBK_BWD_LOG2P", &Options::bwd_buckets_log2},
    {"bwd_bucket_pairs", "HBK_BWD_TARGET", &Options::bwd_bucket_pairs},
    {"bwd_split_pairs", "HBK_BWD_SPLIT", &Options::bwd_split_pairs},
    {"bwd_onepass", "HBK_BWD_ONEPASS", &Options::bwd_onepass},
    {"bwd_group_cols", "HBK_BWD_GROUP_COLS", &Options::bwd_group_cols},
    {"bwd_dense", "HBK_BWD_DENSE", &Options::bwd_dense},
    {"bwd_scatter_staged", "HBK_BWD_SCATTER_STAGED", &Options::bwd_scatter_staged},
    {"bwd_rowsort_pos", "HBK_BWD_ROWSORT_POS", &Options::bwd_rowsort_pos},
    {"bwd_rowsort_ratio", "HBK_BWD_ROWSORT_RATIO", &Options::bwd_rowsort_ratio},
    {"bwd_deterministic", "HBK_BWD_DETERMINISTIC", &Options::bwd_deterministic},
    {"bwd_streams", "HBK_BWD_STREAMS", &Options::bwd_streams},
    {"bwd_trace", "HBK_BWD_TRACE", &Options::bwd_trace},
    {"bwd_lds_pad", "HBK_BWD_LDS_PAD", &Options::bwd_lds_pad},
    {"bwd_large_first", "HBK_BWD_LARGE_FIRST", &Options::bwd_large_first},
    {"bwd_pairs_packed", "HBK_BWD_PAIRS_PACKED", &Options::bwd_pairs_packed},
    {"bwd_seg_inline", "HBK_BWD_SEG_INLINE", &Options::bwd_seg_inline},
    {"bwd_scale_fused", "HBK_BWD_SCALE_FUSED", &Options::bwd_scale_fused},
    {"bwd_simple", "HBK_BWD_SIMPLE", &Options::bwd_simple},
    {"fwd_d16", "HBK_FWD_D16", &Options::fwd_d16},
    {"bwd_wide", "HBK_BWD_WIDE", &Options::bwd_wide},
    {"bwd_xcd", "HBK_BWD_XCD", &Options::bwd_xcd},
    {"fwd_xcd", "HBK_FWD_XCD", &Options::fwd_xcd},
    {"fwd_hot_rows", "HBK_FWD_HOT", &Options::fwd_hot_rows},
    {"fwd_interleave", "HBK_FWD_INTERLEAVE", &Options::fwd_interleave},
    {"unique_buckets_log2", "HBK_UNIQUE_LOG2P", &Options::unique_buckets_log2},
    {"partition_sub_tiles", "HBK_PART_SUB", &Options::partition_sub_tiles},
    {"partition_fixed_max", "HBK_PART_FIXED", &Options::partition_fixed_max},
    {"partition_onepass", "HBK_PART_ONEPASS", &Options::partition_onepass},
    {"unique_onepass", "HBK_UNIQUE_ONEPASS", &Options::unique_onepass},
    {"sharded_groups", "HBK_SHARDED_GROUPS", &Options::sharded_groups},
    {"sharded_id64", "HBK_SHARDED_ID64", &Options::sharded_id64},
    {"sharded_copy_self", "HBK_SHARDED_COPY_SELF", &Options::sharded_copy_self},
    {"sharded_trace", "HBK_SHARDED_TRACE", &Options::sharded_trace},
    {"sharded_inline", "HBK_SHARDED_INLINE", &Options::sharded_inline},
    {"sharded_p2p", "HBK_SHARDED_P2P", &Options::sharded_p2p},
    {"sharded_p2p_test_refuse", "HBK_SHARDED_P2P_TEST_REFUSE", &Options::sharded_p2p_test_refuse},
    {"sharded_wire_fused", "HBK_SHARDED_WIRE_FUSED", &Options::sharded_wire_fused},
    {"sharded_pack_early", "HBK_SHARDED_PACK_EARLY", &Options::sharded_pack_early},
    {"sync_wait_ms", "HBK_SYNC_WAIT_MS", &Options::sync_wait_ms},
    {"sync_onepass_off", "HBK_SYNC_ONEPASS_OFF", &Options::sync_onepass_off},
    {"sync_test_withhold", "HBK_SYNC_TEST_WITHHOLD", &Options::sync_test_withhold},
};

Options from_environment() {
  Options o;
  for (const OptionEntry& e : kOptions) {
    const char* v = getenv(e.env);
    if (v != nullptr && *v != 0) o.*(e.field) = atoi(v);
  }
  return o;
}
}  // namespace

Options& options() {
  static Options o = from_environment();   // once, at first use
  return o;
}
}  // namespace hbk

extern "C" const char* hbk_last_error(void) { return hbk::g_last_error; }

extern "C" int hbk_set_option(const char* name, int32_t value) {
  using namespace hbk;
  HBK_REQUIRE(name != nullptr, "set_option: name is NULL");
  for (const OptionEntry& e : kOptions) {
    if (strcmp(e.name, name) == 0) {
      options().*(e.field) = value;
      return HBK_OK;
    }
  }
  return fail(HBK_INVALID_ARGUMENT, "set_option: unknown option '%s'", name);
}

extern "C" int hbk_get_option(const char* name, int32_t* value) {
  using namespace hbk;
  HBK_REQUIRE(name != nullptr && value != nullptr, "get_option: NULL argument");
  for (const OptionEntry& e : kOptions) {
    if (strcmp(e.name, name) == 0) {
      *value = options().*(e.field);
      return HBK_OK;
    }
  }
  return fail(HBK_INVALID_ARGUMENT, "get_option: unknown option '%s'", name);
}

extern "C" int hbk_sync_check(void) { return hbk::sync_check_any("sync_check"); }
extern "C" int hbk_sync_check_stream(hbk_stream_t stream) {
  return hbk::sync_check("sync_check_stream", reinterpret_cast<hipStream_t>(stream));
}

extern "C" const char* hbk_version(void) { return "hbk 0.1.0 gfx950"; }

// One slab for N tables, each at a 2 MB-aligned offset: the allocation policy that was fastest in
// every run of tools/placement_probe (profiles/r05_placement.txt: -2 % on config 4's forward, 2.1 x
// fewer UTCL1 translation misses than one hipMalloc per table).  Offsets by hbk_tables_layout, so
// a framework that owns its memory can carve its own slab the same way.
extern "C" size_t hbk_tables_layout(int32_t n, const size_t* bytes, size_t* offsets) {
  const size_t kAlign = (size_t)2 << 20;
  size_t total = 0;
  if (n > 0 && bytes == nullptr) {   // (the size is the only return value: 0 = nothing laid out)
    (void)hbk::fail(HBK_INVALID_ARGUMENT, "tables_layout: bytes is NULL");
    return 0;
  }
  for (int32_t i = 0; i < n; ++i) {
    if (offsets != nullptr) offsets[i] = total;
    total += (bytes[i] + kAlign - 1) / kAlign * kAlign;
  }
  return total;
}
extern "C" int hbk_tables_alloc(int32_t n, const size_t* bytes, void** tables, void** slab) {
  using namespace hbk;
  HBK_REQUIRE(n >= 0 && (n == 0 || (bytes && tables)) && slab != nullptr,
              "tables_alloc: bad arguments");
  *slab = nullptr;
  if (n == 0) return HBK_OK;
  std::vector<size_t> off((size_t)n);
  const size_t total = hbk_tables_layout(n, bytes, off.data());
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, total > 0 ? total : 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HBK_INTERNAL, "tables_alloc: hipMalloc of %zu bytes failed: %s", total,
                hipGetErrorString(e));
  }
  for (int32_t i = 0; i < n; ++i) tables[i] = static_cast<char*>(p) + off[(size_t)i];
  *slab = p;
  return HBK_OK;
}
extern "C" int hbk_tables_free(void* slab) {
  using namespace hbk;
  if (slab == nullptr) return HBK_OK;
  const hipError_t e = hipFree(slab);
  if (e != hipSuccess) return fail(HBK_INTERNAL, "tables_free: %s", hipGetErrorString(e));
  return HBK_OK;
}

// Host-side evaluation of the device's divide-free floor-mod / floor-div (common.h), so the
// magic-number arithmetic can be checked exhaustively without a GPU.
extern "C" int64_t hbk_host_floormod_i64(int64_t v, int64_t d) {
  if (d <= 0) return -1;
  hbk::FastDiv f = hbk::make_fastdiv((uint64_t)d);
  f.d = (uint64_t)d;
  return (int64_t)hbk::floormod_i64(v, f);
}

// the block -> work item mapping of the XCD-aware launches (xcd_contiguous), for a bijection check
extern "C" int32_t hbk_host_xcd_contiguous(int32_t block, int32_t n_blocks) {
  return hbk::xcd_contiguous(block, n_blocks, 1);
}

extern "C" uint64_t hbk_host_fastdiv_u64(uint64_t n, uint64_t d) {
  if (d == 0) return 0;
  hbk::FastDiv f = hbk::make_fastdiv(d);
  return hbk::fastdiv(n, f);
}

// CRC-32C (Castagnoli, reflected polynomial 0x82f63b78) of host bytes, continuing from `crc`
// (0 for a new message): the checksum TensorFlow's tensor bundle stores per tensor and per index
// block (training/tf_bundle.py reads and writes that format on the host).  Slicing by 8.
extern "C" uint32_t hbk_host_crc32c(uint32_t crc, const void* data, int64_t n) {
  struct Tables {
    uint32_t t[8][256];
    Tables() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
        t[0][i] = c;
      }
      for (uint32_t i = 0; i < 256; ++i) {
        for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
      }
    }
  };
  static const Tables tab;
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t l = ~crc;
  if (p == nullptr || n <= 0) return crc;
  while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7u) != 0) {
    l = tab.t[0][(l ^ *p++) & 0xffu] ^ (l >> 8);
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    const uint32_t lo = (uint32_t)w ^ l, hi = (uint32_t)(w >> 32);
    l = tab.t[7][lo & 0xffu] ^ tab.t[6][(lo >> 8) & 0xffu] ^ tab.t[5][(lo >> 16) & 0xffu] ^
        tab.t[4][lo >> 24] ^ tab.t[3][hi & 0xffu] ^ tab.t[2][(hi >> 8) & 0xffu] ^
        tab.t[1][(hi >> 16) & 0xffu] ^ tab.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n > 0) {
    l = tab.t[0][(l ^ *p++) & 0xffu] ^ (l >> 8);
    --n;
  }
  return ~l;
}
