/*
 * hbk_oracle.c -- CPU restatement of the HybridBackend sharded-embedding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hybridbackend_amd/ may import, link or
 * call this file.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py -- always as the checker / reported baseline,
 * never as the thing shipped.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * restates.  Abbreviation: hbtf/ = hybridbackend/tensorflow/.
 *
 * Pinning status
 *   - R2/R3 partition, R4 active ranks, R5 alltoallv offsets, R11 murmur3 + probe:
 *     pinned by the reference's own known-answer vectors and by vectors derived
 *     from its CPU functor (tests/golden/).  murmur3 additionally pinned against
 *     the reference header compiled as-is (oracle/_ref, see oracle/Makefile).
 *   - R1 FloorMod, R7 unique, R8 gather, R9 combiner, R10 scatter-add live in
 *     TensorFlow 1.15 (third-party; not under /root/reference, not installable
 *     here).  Their published semantics are restated; the reference's tests never
 *     assert embedding values (hbtf/embedding/tests/deeprecev_test.py:73-79 only
 *     prints).  These rows are pinned to what exists outside this restatement:
 *     the worked examples printed in the TF 1.15 API documentation and vectors
 *     derived from the rules it states (tests/golden/tf115_semantics.json, each
 *     entry marked "published" or "derived"), numpy float64 (config-1 fixture)
 *     and a second implementation (torch embedding_bag + autograd).  Agreement
 *     is with TF 1.15's documentation, not with its binary: nothing runs it here.
 *     One documented difference: TF 1.15's CPU SparseSegmentReduction adds rows
 *     in blocks of 8 (Eigen), this restatement strictly in order of j -- the two
 *     agree within the 1e-5 the path is held to, not bit for bit.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* R1  bucketize: `feature % embedding_size` -- docs/tutorial/ranking/data.py:179,186
 * (TF FloorMod; TF1.15 tensorflow/core/kernels/cwise_ops.h google_floor_mod:
 *  trunc = x % y; (trunc != 0 && ((y < 0) != (trunc < 0))) ? trunc + y : trunc).   */
void orc_floormod_i64(const int64_t* in, int64_t n, int64_t m, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t t = in[i] % m;
    out[i] = (t != 0 && ((m < 0) != (t < 0))) ? t + m : t;
  }
}

void orc_floormod_i32(const int32_t* in, int64_t n, int32_t m, int32_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    int32_t t = in[i] % m;
    out[i] = (t != 0 && ((m < 0) != (t < 0))) ? t + m : t;
  }
}

/* ------------------------------------------------------------------------- */
/* R2  PartitionByModulo<CPUDevice,T>::operator()
 *     hbtf/distribute/partition/partition_by_modulo_functors.cc:39-70
 * Same statement order, same integer types (int32 num_partitions, T shard,
 * int32 offsets) so C's usual arithmetic conversions reproduce the C++ ones.    */
#define ORC_PARTITION_BY_MODULO(NAME, T)                                          \
  void NAME(int32_t num_partitions, const T* h_input, int32_t input_size,         \
            T* h_output, int32_t* h_sizes, int32_t* h_indices) {                  \
    int32_t* local_offsets =                                                      \
        (int32_t*)calloc((size_t)(input_size > 0 ? input_size : 1), 4);           \
    int32_t* shard_offsets = (int32_t*)calloc((size_t)num_partitions, 4);         \
    for (int64_t i = 0; i < input_size; ++i) {                                    \
      const T shard =                                                             \
          (h_input[i] % num_partitions + num_partitions) % num_partitions;        \
      local_offsets[i] = shard_offsets[shard];                                    \
      shard_offsets[shard]++;                                                     \
    }                                                                             \
    memcpy(h_sizes, shard_offsets, (size_t)num_partitions * sizeof(int32_t));     \
    for (int32_t i = 1; i < num_partitions; ++i) {                                \
      shard_offsets[i] += shard_offsets[i - 1];                                   \
    }                                                                             \
    for (int64_t i = 0; i < input_size; ++i) {                                    \
      const T v = h_input[i];                                                     \
      const T shard = (v % num_partitions + num_partitions) % num_partitions;     \
      int32_t offset = local_offsets[i];                                          \
      if (shard > 0) {                                                            \
        offset += shard_offsets[shard - 1];                                       \
      }                                                                           \
      h_output[offset] = v;                                                       \
      h_indices[i] = offset;                                                      \
    }                                                                             \
    free(local_offsets);                                                          \
    free(shard_offsets);                                                          \
  }

ORC_PARTITION_BY_MODULO(orc_partition_by_modulo_i32, int32_t)
ORC_PARTITION_BY_MODULO(orc_partition_by_modulo_i64, int64_t)
ORC_PARTITION_BY_MODULO(orc_partition_by_modulo_u32, uint32_t)
ORC_PARTITION_BY_MODULO(orc_partition_by_modulo_u64, uint64_t)

/* ------------------------------------------------------------------------- */
/* R3  PartitionByDualModulo<CPUDevice,T,ComputeShard>::operator()
 *     hbtf/distribute/partition/partition_by_dual_modulo_functors.cc:37-91
 *     stage 1: ComputeShardAtStageOne (:37-42)  (pre % P + P) % P
 *     stage 2: ComputeShardAtStageTwo (:44-49)  pre / modulus                    */
#define ORC_PARTITION_BY_DUAL_MODULO(NAME, T)                                     \
  void NAME(int32_t num_partitions, int32_t modulus, int32_t stage,               \
            const T* h_input, int32_t input_size, T* h_output, int32_t* h_sizes,  \
            int32_t* h_indices) {                                                 \
    size_t cap = (size_t)(input_size > 0 ? input_size : 1);                       \
    int32_t* local_offsets = (int32_t*)calloc(cap, 4);                            \
    int32_t* shard_offsets = (int32_t*)calloc((size_t)num_partitions, 4);         \
    T* shard_idx = (T*)calloc(cap, sizeof(T));                                    \
    const int32_t pre_mod_size = num_partitions * modulus;                        \
    for (int64_t i = 0; i < input_size; ++i) {                                    \
      const T pre_mod_res =                                                       \
          (h_input[i] % pre_mod_size + pre_mod_size) % pre_mod_size;              \
      if (stage == 1) {                                                           \
        shard_idx[i] =                                                            \
            (pre_mod_res % num_partitions + num_partitions) % num_partitions;     \
      } else {                                                                    \
        shard_idx[i] = pre_mod_res / modulus;                                     \
      }                                                                           \
    }                                                                             \
    for (int64_t i = 0; i < input_size; ++i) {                                    \
      local_offsets[i] = shard_offsets[shard_idx[i]];                             \
      shard_offsets[shard_idx[i]]++;                                              \
    }                                                                             \
    memcpy(h_sizes, shard_offsets, (size_t)num_partitions * sizeof(int32_t));     \
    for (int32_t i = 1; i < num_partitions; ++i) {                                \
      shard_offsets[i] += shard_offsets[i - 1];                                   \
    }                                                                             \
    for (int64_t i = 0; i < input_size; ++i) {                                    \
      const T v = h_input[i];                                                     \
      const T shard = shard_idx[i];                                               \
      int32_t offset = local_offsets[i];                                          \
      if (shard > 0) {                                                            \
        offset += shard_offsets[shard - 1];                                       \
      }                                                                           \
      h_output[offset] = v;                                                       \
      h_indices[i] = offset;                                                      \
    }                                                                             \
    free(local_offsets);                                                          \
    free(shard_offsets);                                                          \
    free(shard_idx);                                                              \
  }

ORC_PARTITION_BY_DUAL_MODULO(orc_partition_by_dual_modulo_i32, int32_t)
ORC_PARTITION_BY_DUAL_MODULO(orc_partition_by_dual_modulo_i64, int64_t)
ORC_PARTITION_BY_DUAL_MODULO(orc_partition_by_dual_modulo_u32, uint32_t)
ORC_PARTITION_BY_DUAL_MODULO(orc_partition_by_dual_modulo_u64, uint64_t)

/* ------------------------------------------------------------------------- */
/* R4  Collective::compute_active_ranks / compute_active_size
 *     hbtf/distribute/collective.h:80-112.  topology: 0 ALL, 1 INTRA, 2 INTER.
 * Returns the number of ranks written to `out` (capacity world_size).           */
int32_t orc_compute_active_ranks(int32_t topology, int32_t world_size,
                                 int32_t local_size, int32_t rank_, int32_t* out) {
  int32_t k = 0;
  if (topology == 1) {
    int32_t node_idx = rank_ / local_size;
    for (int32_t rank = node_idx * local_size; rank < (node_idx + 1) * local_size;
         ++rank) {
      out[k++] = rank;
    }
  } else if (topology == 2) {
    for (int32_t rank = 0; rank < world_size; ++rank) {
      if (local_size == 1 || (rank % local_size) == (rank_ % local_size)) {
        out[k++] = rank;
      }
    }
  } else {
    for (int32_t rank = 0; rank < world_size; ++rank) {
      out[k++] = rank;
    }
  }
  return k;
}

int32_t orc_compute_active_size(int32_t topology, int32_t world_size,
                                int32_t local_size) {
  if (topology == 1) return local_size;
  if (topology == 2) return world_size / local_size;
  return world_size;
}

/* ------------------------------------------------------------------------- */
/* R5  NcclCollective::Alltoallv offset arithmetic
 *     hbtf/distribute/nccl/nccl_collective.cc:250-288: chunk i of the send buffer
 *     starts at sum_{j<i} send_sizes[j]*common_size elements; the chunk received
 *     from peer i lands at sum_{j<i} recv_sizes[j]*common_size.
 * Simulates all `world` ranks in one process (topology ALL).
 *   send_sizes : [world][world]  send_sizes[r][i] = rows rank r sends to rank i
 *   recv_sizes : [world][world]  out; recv_sizes[r][i] = send_sizes[i][r]
 *                (this is what Alltoall(sizes) delivers, nccl_alltoallv.cc:306-308)
 *   send_bufs[r], recv_bufs[r]: byte buffers, elem_bytes*common_size per row.
 *   recv_bufs[r] must hold sum_i recv_sizes[r][i] rows.                         */
void orc_alltoallv_sim(int32_t world, int64_t row_bytes,
                       const uint8_t* const* send_bufs, const int32_t* send_sizes,
                       uint8_t* const* recv_bufs, int32_t* recv_sizes) {
  for (int32_t r = 0; r < world; ++r) {
    for (int32_t i = 0; i < world; ++i) {
      recv_sizes[r * world + i] = send_sizes[i * world + r];
    }
  }
  for (int32_t r = 0; r < world; ++r) {
    int64_t recvoffset = 0;
    for (int32_t i = 0; i < world; ++i) {
      /* where does peer i keep the chunk addressed to r? */
      int64_t sendoffset = 0;
      for (int32_t j = 0; j < r; ++j) sendoffset += send_sizes[i * world + j];
      int64_t rows = send_sizes[i * world + r];
      if (rows > 0 && recv_bufs[r] != NULL) {
        memcpy(recv_bufs[r] + recvoffset * row_bytes,
               send_bufs[i] + sendoffset * row_bytes, (size_t)(rows * row_bytes));
      }
      recvoffset += rows;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* R6  fp32 <-> fp16 wire casts: hbtf/common/cast.cu.cc:37-42 (__float2half,
 *     round-to-nearest-even) and :60-65 (__half2float).  Bit-level software
 *     conversion; handles subnormals, inf, nan.                                 */
static uint16_t f32_to_f16_rne(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 0xffu) { /* inf / nan */
    if (man == 0) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | 0x7c00u | 0x200u | (man >> 13));
  }
  int32_t e = (int32_t)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
  if (e <= 0) {                                   /* subnormal or zero */
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
    return (uint16_t)(sign | half_man);
  }
  uint32_t half = (uint32_t)(e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
  return (uint16_t)(sign | half);
}

static float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (man == 0) {
      x = sign;
    } else {
      int32_t e = -1;
      do {
        man <<= 1;
        e++;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      x = sign | (uint32_t)((127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    x = sign | 0x7f800000u | (man << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

void orc_cast_f32_to_f16(const float* in, int64_t n, uint16_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = f32_to_f16_rne(in[i]);
}

void orc_cast_f16_to_f32(const uint16_t* in, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = f16_to_f32(in[i]);
}

/* ------------------------------------------------------------------------- */
/* R7  owner-side `array_ops.unique` -- hbtf/embedding/sharding.py:186 (TF Unique:
 *     output in FIRST-OCCURRENCE order, idx[i] = position of in[i] in output).
 *     Pinned to the TF 1.15 docstring example (tests/golden: unique, published).
 * Open-addressing hash table; returns the number of unique values.              */
static inline uint64_t orc_mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

/* slots of the open-addressing table orc_unique_i64 uses for n ids */
static uint64_t orc_unique_slots(int64_t n) {
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2) cap <<= 1;
  return cap;
}

/* the same with the table provided by the caller (orc_unique_slots(n) int64): the timed baseline
 * keeps one per thread instead of a malloc / free per column and pass */
static int64_t orc_unique_i64_with(const int64_t* in, int64_t n, int64_t* uniq, int32_t* idx,
                                   int64_t* slot_pos) {
  if (n <= 0) return 0;
  const uint64_t cap = orc_unique_slots(n);
  for (uint64_t i = 0; i < cap; ++i) slot_pos[i] = -1;
  int64_t u = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t h = orc_mix64((uint64_t)in[i]) & (cap - 1);
    for (;;) {
      int64_t p = slot_pos[h];
      if (p < 0) {
        slot_pos[h] = u;
        uniq[u] = in[i];
        idx[i] = (int32_t)u;
        ++u;
        break;
      }
      if (uniq[p] == in[i]) {
        idx[i] = (int32_t)p;
        break;
      }
      h = (h + 1) & (cap - 1);
    }
  }
  return u;
}

int64_t orc_unique_i64(const int64_t* in, int64_t n, int64_t* uniq, int32_t* idx) {
  if (n <= 0) return 0;
  uint64_t cap = orc_unique_slots(n);
  int64_t* slot_pos = (int64_t*)malloc(cap * sizeof(int64_t));
  for (uint64_t i = 0; i < cap; ++i) slot_pos[i] = -1;
  int64_t u = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t h = orc_mix64((uint64_t)in[i]) & (cap - 1);
    for (;;) {
      int64_t p = slot_pos[h];
      if (p < 0) {
        slot_pos[h] = u;
        uniq[u] = in[i];
        idx[i] = (int32_t)u;
        ++u;
        break;
      }
      if (uniq[p] == in[i]) {
        idx[i] = (int32_t)p;
        break;
      }
      h = (h + 1) & (cap - 1);
    }
  }
  free(slot_pos);
  return u;
}

/* ------------------------------------------------------------------------- */
/* R8  local gather `fn(params, shard_ids)` -- hbtf/embedding/sharding.py:191,193,200
 *     (TF GatherV2 axis 0): out[k,:] = table[row[k],:].  Out-of-range rows give
 *     zeros (TF GPU GatherV2 behaviour).  A copy: exact by construction.          */
void orc_gather_f32(const float* table, int64_t rows, int32_t dim,
                    const int64_t* row_ids, int64_t n, float* out) {
  for (int64_t k = 0; k < n; ++k) {
    int64_t r = row_ids[k];
    if (r < 0 || r >= rows) {
      memset(out + k * dim, 0, (size_t)dim * sizeof(float));
    } else {
      memcpy(out + k * dim, table + r * dim, (size_t)dim * sizeof(float));
    }
  }
}

void orc_gather_f32_i32(const float* table, int64_t rows, int32_t dim,
                        const int32_t* row_ids, int64_t n, float* out) {
  for (int64_t k = 0; k < n; ++k) {
    int64_t r = row_ids[k];
    if (r < 0 || r >= rows) {
      memset(out + k * dim, 0, (size_t)dim * sizeof(float));
    } else {
      memcpy(out + k * dim, table + r * dim, (size_t)dim * sizeof(float));
    }
  }
}

/* ------------------------------------------------------------------------- */
/* R9  combiner -- TF1.15 embedding_lookup_sparse -> sparse_segment_{sum,mean,sqrt_n}
 *     call sites docs/tutorial/ranking/data.py:189-193,
 *     hbtf/benchmarks/embedding_benchmark_tier1.py:65-66.
 *     out[s,:] = sum_{j in [splits[s],splits[s+1])} emb[idx[j],:]
 *     mean: / count, sqrtn: / sqrt(count); empty segment -> 0.
 *     fp32, fixed IN-ORDER accumulation (the documented association order of this
 *     build; TF's own chunked order differs in the last ulp -> tolerance 1e-5).
 *     Pinned to the tf.sparse.segment_sum / segment_mean docstring examples
 *     (published) and sum / sqrt(N) (derived).  combiner: 0 sum, 1 mean, 2 sqrtn.
 *     idx may be NULL (identity: emb row j).                                     */
void orc_segment_combine_f32(const float* emb, int32_t dim, const int32_t* idx,
                             const int32_t* splits, int64_t n_segments,
                             int32_t combiner, float* out) {
  for (int64_t s = 0; s < n_segments; ++s) {
    float* o = out + s * dim;
    for (int32_t d = 0; d < dim; ++d) o[d] = 0.0f;
    int32_t beg = splits[s], end = splits[s + 1];
    for (int32_t j = beg; j < end; ++j) {
      const float* e = emb + (int64_t)(idx ? idx[j] : j) * dim;
      for (int32_t d = 0; d < dim; ++d) o[d] = o[d] + e[d];
    }
    int32_t cnt = end - beg;
    if (cnt > 0 && combiner == 1) {
      float c = (float)cnt;
      for (int32_t d = 0; d < dim; ++d) o[d] = o[d] / c;
    } else if (cnt > 0 && combiner == 2) {
      float c = sqrtf((float)cnt);
      for (int32_t d = 0; d < dim; ++d) o[d] = o[d] / c;
    }
  }
}

/* float64 accumulation of the same thing: the numerical reference the 1e-5
 * relative tolerance is measured against.                                       */
void orc_segment_combine_f64acc(const float* emb, int32_t dim, const int32_t* idx,
                                const int32_t* splits, int64_t n_segments,
                                int32_t combiner, double* out) {
  for (int64_t s = 0; s < n_segments; ++s) {
    double* o = out + s * dim;
    for (int32_t d = 0; d < dim; ++d) o[d] = 0.0;
    int32_t beg = splits[s], end = splits[s + 1];
    for (int32_t j = beg; j < end; ++j) {
      const float* e = emb + (int64_t)(idx ? idx[j] : j) * dim;
      for (int32_t d = 0; d < dim; ++d) o[d] += (double)e[d];
    }
    int32_t cnt = end - beg;
    if (cnt > 0 && combiner == 1) {
      for (int32_t d = 0; d < dim; ++d) o[d] /= (double)cnt;
    } else if (cnt > 0 && combiner == 2) {
      for (int32_t d = 0; d < dim; ++d) o[d] /= sqrt((double)cnt);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* R10 backward.  TF autodiff of R9 + R8 + R7 (hbtf/embedding/sharding.py:186-200
 *     in reverse; hbtf/distribute/collective.py:334-347 for the exchange):
 *       d(combiner): g_id[j,:] = g_out[seg(j),:] * scale(seg)   (sum 1, mean 1/cnt,
 *                    sqrtn 1/sqrt(cnt))  -- SparseSegment*Grad
 *       d(gather by idx) = UnsortedSegmentSum(g_id, idx, u): g_u[idx[j],:] += g_id[j,:]
 *     in-order accumulation over j (TF 1.15's CPU kernel adds in blocks of 8:
 *     same value within 1e-5).  Pinned as R9 + the embedding_lookup_sparse example. */
void orc_segment_combine_grad_f32(const float* g_out, int32_t dim,
                                  const int32_t* splits, int64_t n_segments,
                                  int32_t combiner, float* g_id) {
  for (int64_t s = 0; s < n_segments; ++s) {
    int32_t beg = splits[s], end = splits[s + 1];
    int32_t cnt = end - beg;
    const float* g = g_out + s * dim;
    for (int32_t j = beg; j < end; ++j) {
      float* o = g_id + (int64_t)j * dim;
      if (combiner == 1) {
        float c = (float)cnt;
        for (int32_t d = 0; d < dim; ++d) o[d] = g[d] / c;
      } else if (combiner == 2) {
        float c = sqrtf((float)cnt);
        for (int32_t d = 0; d < dim; ++d) o[d] = g[d] / c;
      } else {
        for (int32_t d = 0; d < dim; ++d) o[d] = g[d];
      }
    }
  }
}

void orc_unsorted_segment_sum_f32(const float* g, int32_t dim, const int32_t* idx,
                                  int64_t n, int64_t num_segments, float* out) {
  memset(out, 0, (size_t)(num_segments * dim) * sizeof(float));
  for (int64_t j = 0; j < n; ++j) {
    int64_t u = idx[j];
    if (u < 0 || u >= num_segments) continue;
    float* o = out + u * dim;
    const float* e = g + j * dim;
    for (int32_t d = 0; d < dim; ++d) o[d] = o[d] + e[d];
  }
}

void orc_unsorted_segment_sum_f64acc(const float* g, int32_t dim,
                                     const int32_t* idx, int64_t n,
                                     int64_t num_segments, double* out) {
  memset(out, 0, (size_t)(num_segments * dim) * sizeof(double));
  for (int64_t j = 0; j < n; ++j) {
    int64_t u = idx[j];
    if (u < 0 || u >= num_segments) continue;
    double* o = out + u * dim;
    const float* e = g + j * dim;
    for (int32_t d = 0; d < dim; ++d) o[d] += (double)e[d];
  }
}

/* sparse SGD apply on the shard: table[row[u],:] -= lr * g_u[u,:]
 * (TF ScatterSub / SparseApplyGradientDescent on IndexedSlices;
 *  sharded variables skip cross-rank aggregation, hbtf/training/gradient.py:193-217) */
void orc_sparse_sgd_apply_f32(float* table, int64_t rows, int32_t dim,
                              const int64_t* row_ids, const float* g_u, int64_t u,
                              float lr) {
  for (int64_t k = 0; k < u; ++k) {
    int64_t r = row_ids[k];
    if (r < 0 || r >= rows) continue;
    float* t = table + r * dim;
    const float* g = g_u + k * dim;
    for (int32_t d = 0; d < dim; ++d) t[d] = t[d] - lr * g[d];
  }
}

/* ------------------------------------------------------------------------- */
/* R11 murmur3_hash32<T=int64, seed=0> -- hybridbackend/common/murmur3.cu.h:32-77
 *     (MurmurHash3_x86_32 over the 8 key bytes: two 4-byte blocks, no tail).    */
static inline uint32_t orc_rotl32(uint32_t x, int8_t r) {
  return (x << r) | (x >> (32 - r));
}

uint32_t orc_murmur3_hash32_i64(int64_t key, uint32_t seed) {
  const int len = 8;
  uint32_t blocks[2];
  memcpy(blocks, &key, 8);
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51u;
  const uint32_t c2 = 0x1b873593u;
  for (int i = 0; i < 2; ++i) {
    uint32_t k1 = blocks[i];
    k1 *= c1;
    k1 = orc_rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = orc_rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

void orc_murmur3_hash32_i64_n(const int64_t* keys, int64_t n, uint32_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = orc_murmur3_hash32_i64(keys[i], 0);
}

/* R11 cache probe -- LookupKernel hbtf/embedding/lookup_functors.cu.cc:54-149,
 *     one key at a time (the per-key outcome does not depend on the warp it ran
 *     in): slab = murmur3(key) % slab_count; scan the slab's slots: first slot
 *     whose key matches -> hit (cache index = slab*slab_size + slot); else if the
 *     slab holds an EMPTY (INT64_MIN, service.py:87) slot -> miss; else next slab
 *     (linear, wrapping); all slabs probed -> miss.
 *     hit_slot[i] = cache index or -1 for a miss.                                */
void orc_cache_probe_i64(const int64_t* keys_cache, int64_t slab_count,
                         int32_t slab_size, const int64_t* keys, int64_t n,
                         int64_t* hit_slot) {
  const int64_t kEmpty = INT64_MIN;
  for (int64_t i = 0; i < n; ++i) {
    int64_t key = keys[i];
    int64_t slab = (int64_t)(orc_murmur3_hash32_i64(key, 0) % (uint64_t)slab_count);
    int64_t result = -1;
    for (int64_t probed = 0; probed < slab_count; ++probed) {
      const int64_t* s = keys_cache + slab * slab_size;
      int good = -1, has_empty = 0;
      for (int32_t k = 0; k < slab_size; ++k) {
        if (good < 0 && s[k] == key) good = k;
        if (s[k] == kEmpty) has_empty = 1;
      }
      if (good >= 0) {
        result = slab * slab_size + good;
        break;
      }
      if (has_empty) break;
      slab = (slab + 1) % slab_count;
    }
    hit_slot[i] = result;
  }
}

/* ------------------------------------------------------------------------- */
/* R13 shard sizing -- hbtf/embedding/variables.py:93-123.
 *     returns 0 when the table stays replicated (bucket<=W or bucket<=batch),
 *     else writes rows_local and the (contiguous) save-slice offset.            */
int32_t orc_shard_rows(int64_t bucket_size, int32_t num_shards, int32_t shard,
                       int64_t batch_size, int64_t* rows_local,
                       int64_t* bucket_offset) {
  if (bucket_size <= num_shards || bucket_size <= batch_size) {
    *rows_local = bucket_size;
    *bucket_offset = 0;
    return 0;
  }
  int64_t sharded = bucket_size / num_shards;
  if (shard < bucket_size % num_shards) sharded += 1;
  int64_t off = (bucket_size / num_shards) * shard;
  int64_t remained = bucket_size % num_shards;
  off += (shard < remained) ? shard : remained;
  *rows_local = sharded;
  *bucket_offset = off;
  return 1;
}

/* ------------------------------------------------------------------------- */
/* R12 (W = 1) composition used as the host-CPU baseline in bench.py:
 *     bucketize (R1) -> partition_by_modulo P=1 (R2) -> unique (R7) -> gather (R8)
 *     -> restore by unique index + stitch by shard index (R8) -> combiner (R9),
 *     per column, columns spread over a pthread pool (the reference's CPU analogue
 *     is TF's inter-op pool over the per-column ops).                            */
typedef struct {
  const float* table;
  int64_t rows;
  int32_t dim;
  const int64_t* ids;
  int64_t n_ids;
  const int32_t* splits; /* NULL => one id per segment */
  int64_t n_segments;
  int64_t bucket;
  int32_t combiner;
  float* out;
} orc_lookup_column_t;

/* Per-thread scratch of the pipeline, grown on demand and kept across columns and passes (round 5:
 * the timed baseline used to malloc / free seven arrays per column and pass and hand its tasks out
 * under a mutex -- 234 threads scaled 5.3 x over one). */
typedef struct {
  size_t ids_cap, emb_cap, slots_cap;
  int64_t *bucketized, *shuffled, *uniq, *slot_pos;
  int32_t *shard_index, *uniq_index, *comp;
  float* emb_u;
} orc_scratch_t;

static void orc_scratch_need(orc_scratch_t* s, size_t ids, size_t emb_floats, size_t slots) {
  if (ids > s->ids_cap) {
    free(s->bucketized); free(s->shuffled); free(s->uniq);
    free(s->shard_index); free(s->uniq_index); free(s->comp);
    s->bucketized = (int64_t*)malloc(ids * 8);
    s->shuffled = (int64_t*)malloc(ids * 8);
    s->uniq = (int64_t*)malloc(ids * 8);
    s->shard_index = (int32_t*)malloc(ids * 4);
    s->uniq_index = (int32_t*)malloc(ids * 4);
    s->comp = (int32_t*)malloc(ids * 4);
    s->ids_cap = ids;
  }
  if (emb_floats > s->emb_cap) {
    free(s->emb_u);
    s->emb_u = (float*)malloc(emb_floats * 4);
    s->emb_cap = emb_floats;
  }
  if (slots > s->slots_cap) {
    free(s->slot_pos);
    s->slot_pos = (int64_t*)malloc(slots * 8);
    s->slots_cap = slots;
  }
}

static void orc_scratch_free(orc_scratch_t* s) {
  free(s->bucketized); free(s->shuffled); free(s->uniq); free(s->slot_pos);
  free(s->shard_index); free(s->uniq_index); free(s->comp); free(s->emb_u);
  memset(s, 0, sizeof(*s));
}

static void orc_lookup_one_column_with(const orc_lookup_column_t* c, orc_scratch_t* s) {
  int64_t n = c->n_ids;
  size_t cap = (size_t)(n > 0 ? n : 1);
  orc_scratch_need(s, cap, cap * (size_t)c->dim, (size_t)orc_unique_slots(n > 0 ? n : 1));
  int32_t sizes[1];
  if (c->bucket > 0) {
    orc_floormod_i64(c->ids, n, c->bucket, s->bucketized);
  } else {
    memcpy(s->bucketized, c->ids, (size_t)n * 8);
  }
  orc_partition_by_modulo_i64(1, s->bucketized, (int32_t)n, s->shuffled, sizes, s->shard_index);
  int64_t u = orc_unique_i64_with(s->shuffled, n, s->uniq, s->uniq_index, s->slot_pos);
  orc_gather_f32(c->table, c->rows, c->dim, s->uniq, u, s->emb_u);
  for (int64_t j = 0; j < n; ++j) s->comp[j] = s->uniq_index[s->shard_index[j]];
  if (c->splits) {
    orc_segment_combine_f32(s->emb_u, c->dim, s->comp, c->splits, c->n_segments, c->combiner,
                            c->out);
  } else {
    orc_gather_f32_i32(s->emb_u, u, c->dim, s->comp, n, c->out);
  }
}

static void orc_lookup_one_column(const orc_lookup_column_t* c) {
  int64_t n = c->n_ids;
  size_t cap = (size_t)(n > 0 ? n : 1);
  int64_t* bucketized = (int64_t*)malloc(cap * 8);
  int64_t* shuffled = (int64_t*)malloc(cap * 8);
  int32_t* shard_index = (int32_t*)malloc(cap * 4);
  int64_t* uniq = (int64_t*)malloc(cap * 8);
  int32_t* uniq_index = (int32_t*)malloc(cap * 4);
  int32_t sizes[1];
  if (c->bucket > 0) {
    orc_floormod_i64(c->ids, n, c->bucket, bucketized);
  } else {
    memcpy(bucketized, c->ids, (size_t)n * 8);
  }
  orc_partition_by_modulo_i64(1, bucketized, (int32_t)n, shuffled, sizes,
                              shard_index);
  int64_t u = orc_unique_i64(shuffled, n, uniq, uniq_index);
  float* emb_u = (float*)malloc((size_t)(u > 0 ? u : 1) * c->dim * 4);
  orc_gather_f32(c->table, c->rows, c->dim, uniq, u, emb_u);
  /* restore duplicates then stitch back to requester order: two chained gathers
   * (sharding.py:193 and :200) == one gather by uniq_index[shard_index[j]] */
  int32_t* comp = (int32_t*)malloc(cap * 4);
  for (int64_t j = 0; j < n; ++j) comp[j] = uniq_index[shard_index[j]];
  if (c->splits) {
    orc_segment_combine_f32(emb_u, c->dim, comp, c->splits, c->n_segments,
                            c->combiner, c->out);
  } else {
    orc_gather_f32_i32(emb_u, u, c->dim, comp, n, c->out);
  }
  free(comp);
  free(emb_u);
  free(uniq_index);
  free(uniq);
  free(shard_index);
  free(shuffled);
  free(bucketized);
}

typedef struct {
  const orc_lookup_column_t* cols;
  int32_t n_cols;
  int32_t n_tasks; /* n_cols * repeat */
  int32_t next;
  pthread_mutex_t mu;
} orc_pool_t;

static void* orc_pool_worker(void* arg) {
  orc_pool_t* p = (orc_pool_t*)arg;
  orc_scratch_t scratch;
  memset(&scratch, 0, sizeof(scratch));
  for (;;) {
    const int32_t i = __atomic_fetch_add(&p->next, 1, __ATOMIC_RELAXED);   /* (no lock: one add) */
    if (i >= p->n_tasks) break;
    orc_lookup_one_column_with(&p->cols[i % p->n_cols], &scratch);
  }
  orc_scratch_free(&scratch);
  return NULL;
}

/* `repeat` passes over the same columns inside one pool (the timed baseline: thread start-up is
 * paid once per call, not once per pass) */
void orc_group_lookup_fwd_repeat(const orc_lookup_column_t* cols, int32_t n_cols,
                                 int32_t n_threads, int32_t repeat) {
  if (repeat < 1) repeat = 1;
  if (n_threads <= 1) {
    orc_scratch_t scratch;
    memset(&scratch, 0, sizeof(scratch));
    for (int32_t r = 0; r < repeat; ++r) {
      for (int32_t i = 0; i < n_cols; ++i) orc_lookup_one_column_with(&cols[i], &scratch);
    }
    orc_scratch_free(&scratch);
    return;
  }
  orc_pool_t pool;
  pool.cols = cols;
  pool.n_cols = n_cols;
  pool.n_tasks = n_cols * repeat;
  pool.next = 0;
  pthread_mutex_init(&pool.mu, NULL);
  pthread_t* th = (pthread_t*)malloc((size_t)n_threads * sizeof(pthread_t));
  for (int32_t t = 0; t < n_threads; ++t) {
    pthread_create(&th[t], NULL, orc_pool_worker, &pool);
  }
  for (int32_t t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  free(th);
  pthread_mutex_destroy(&pool.mu);
}

void orc_group_lookup_fwd(const orc_lookup_column_t* cols, int32_t n_cols,
                          int32_t n_threads) {
  orc_group_lookup_fwd_repeat(cols, n_cols, n_threads, 1);
}
