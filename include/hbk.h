/*
 * hbk.h -- C ABI of libhbk_core.so: the MI355X (gfx950) sharded-embedding engine that
 * sits behind HybridBackend's TensorFlow custom-op surface.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary").  Every entry point replaces the
 * device side of one reference op family; the citation after each declaration is the
 * reference interface it stands in for (paths relative to the reference tree,
 * hbtf/ = hybridbackend/tensorflow/).  INTEGRATION.md shows the REGISTER_OP /
 * REGISTER_KERNEL_BUILDER shim a maintainer adds on the TensorFlow side.
 *
 * Conventions (mirroring the reference's):
 *   - extern "C", plain pointers and sizes; no TF / torch / HIP types in signatures.
 *     `hbk_stream_t` is a hipStream_t passed as void* (NULL = the default stream).
 *   - The caller owns every buffer (TF: ctx->allocate_output / allocate_temp); the
 *     library never frees caller memory.  Scratch is a caller-provided workspace whose
 *     size comes from the matching hbk_*_workspace_bytes() query.
 *   - All device work is enqueued on the given stream and returns without a host
 *     sync unless the entry point's comment says otherwise.
 *   - Return value: 0 (HBK_OK) or a TensorFlow error code (the reference maps
 *     CUDA/NCCL failures to errors::Internal and shape violations to
 *     errors::InvalidArgument: hbtf/common/host_functions.h:37-42,
 *     hbtf/distribute/partition/partition_by_modulo_ops.cc:81-83).
 *     hbk_last_error() returns the calling thread's last message.  Never throws.
 *   - Re-entrant: compute entry points keep no global mutable state.  One communicator
 *     is one ordered queue; the caller issues collective calls in the same order on
 *     every rank (the reference guarantees this with graph linearisation,
 *     hbtf/graph/common/linearization.cc:42-82).
 */
#ifndef HBK_H_
#define HBK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hbk_stream_t;

/* status = TensorFlow error codes */
#define HBK_OK 0
#define HBK_INVALID_ARGUMENT 3
#define HBK_UNIMPLEMENTED 12
#define HBK_INTERNAL 13

/* dtypes: the list HbNcclAlltoallv accepts (hbtf/distribute/nccl/types.h:59-67) */
#define HBK_INT8 0
#define HBK_UINT8 1
#define HBK_INT32 2
#define HBK_UINT32 3
#define HBK_INT64 4
#define HBK_UINT64 5
#define HBK_HALF 6
#define HBK_FLOAT 7
#define HBK_DOUBLE 8

/* combiner of tf.nn.embedding_lookup_sparse (None => mean) */
#define HBK_COMBINER_SUM 0
#define HBK_COMBINER_MEAN 1
#define HBK_COMBINER_SQRTN 2

/* hbtf/distribute/ops.py:34-39, hbtf/distribute/collective.h:52-56 */
#define HBK_TOPOLOGY_ALL 0
#define HBK_TOPOLOGY_INTRA_NODE 1
#define HBK_TOPOLOGY_INTER_NODE 2

const char* hbk_last_error(void);
/* "hbk <version> gfx950" */
const char* hbk_version(void);

/* Table memory (optional; tables stay caller-owned): one slab for N tables, each at a 2 MB-aligned
 * offset -- the policy that was fastest in every run of tools/placement_probe
 * (profiles/r05_placement.txt).  hbk_tables_layout only computes the offsets (returns the slab size)
 * for callers that carve their own memory (a TF allocator, torch); hbk_tables_alloc does it with
 * hipMalloc; hbk_tables_free releases the slab. */
size_t hbk_tables_layout(int32_t n, const size_t* bytes, size_t* offsets);
int hbk_tables_alloc(int32_t n, const size_t* bytes, void** tables, void** slab);
int hbk_tables_free(void* slab);
/* Tuning / diagnostic options of the library, process wide.  Defaults come from the environment
 * once, when the library is first used (HBK_BWD_LOG2P, HBK_BWD_TARGET, HBK_BWD_SPLIT,
 * HBK_BWD_ONEPASS, HBK_BWD_GROUP_COLS, HBK_UNIQUE_LOG2P, HBK_UNIQUE_ONEPASS, HBK_PART_SUB, HBK_PART_FIXED,
 * HBK_PART_ONEPASS, HBK_SHARDED_GROUPS, HBK_SHARDED_ID64, HBK_SHARDED_COPY_SELF,
 * HBK_SHARDED_TRACE); no entry point reads the environment per call.
 * Names: bwd_buckets_log2, bwd_bucket_pairs, bwd_split_pairs, bwd_onepass, bwd_group_cols, bwd_dense, bwd_wide, bwd_xcd, fwd_xcd, fwd_interleave, fwd_hot_rows,
 * bwd_deterministic (1 or 2: the duplicate-row reduction sums every row's terms in ID ORDER by one lane group, so IndexedSlices and stepped
 * tables are bit-identical from run to run and equal to the sequential fp32 sum (TF's CPU UnsortedSegmentSum), and rows leave ascending;
 * 1: columns whose row range fits the row-sorted jobs take those jobs' in-order form -- a row's pairs ordered by gradient row inside the
 * job, output ranges in bucket order, no bucket split over workgroups -- and the other columns the sort of 2; 2: a stable sort of the
 * batch's (row, gradient row) pairs and a sequential walk for every column; the reproducibility mode, TF_DETERMINISTIC_OPS' analogue),
 * bwd_pairs_packed, bwd_seg_inline, bwd_scale_fused, bwd_simple (0: the general instantiation of the grouping kernels for every launch group), fwd_d16 (0: the general gather for columns of 16 floats too), bwd_scatter_staged, bwd_rowsort_pos, bwd_rowsort_ratio (round 5: x 4 for dim <= 32),
 * bwd_streams (launch groups of > 64 columns rotate over this many library streams; 0: the caller's), bwd_large_first,
 * bwd_trace (the launch groups of every backward call on stderr), bwd_lds_pad (a probe), sharded_p2p,
 * sharded_p2p_test_refuse (test hook: the rank whose hbk_sharded_p2p_bind behaves as if a peer's memory could not be mapped),
 * unique_buckets_log2,
 * unique_onepass, partition_sub_tiles, partition_fixed_max, partition_onepass, sharded_groups,
 * sharded_id64, sharded_copy_self, sharded_trace, sharded_inline, sharded_wire_fused, sharded_pack_early (the sharded_* ones are taken by
 * hbk_sharded_create), sync_wait_ms, sync_onepass_off, sync_test_withhold.
 * *_onepass (default 1): small calls of partition / unique / the backward group their ids in ONE
 * launch whose tiles wait for each other; 0 keeps the multi-launch forms.  The
 * waits are bounded (sync_wait_ms, default 2000).  A launch that gave up (never seen outside the
 * test hook sync_test_withhold) poisons its call: the call's later kernels leave without touching
 * anything, its outputs are not valid, and the failure is reported ONCE as HBK_INTERNAL -- by the
 * sharded step in the call that suffered it (it synchronises anyway), else by the next call of
 * these entries ON THE SAME STREAM or by hbk_sync_check[_stream]() -- after which the library takes the multi-launch forms
 * (sync_onepass_off = 1; writable).  The one-launch forms are also not taken when the device
 * cannot hold a whole column's workgroups at once (partitioned modes, small devices; a CU mask set
 * on one stream is not seen by the occupancy query -- there the bounded wait applies).  The words the
 * tiles poll live in buffers the library keeps per (device, stream): calls that share a stream
 * are ordered, which is all these entries ask of the caller. */
int hbk_set_option(const char* name, int32_t value);
int hbk_get_option(const char* name, int32_t* value);
/* HBK_OK, or once per timed-out one-launch wait HBK_INTERNAL (see above).  Callers that read
 * the outputs of hbk_partition_* / hbk_unique_n / hbk_group_lookup_bwd* after their own stream
 * synchronisation call this behind it to learn of a failed call before using its outputs. */
int hbk_sync_check(void);
/* The same for ONE stream of the current device (round 5): the status word of a timed-out wait is
 * kept per (device, stream) -- the stream the failed call was made on -- so a failure on stream A
 * is reported to the next entry call on A (or to this function with A) and is neither seen nor
 * consumed by calls on stream B.  hbk_sync_check() above reports -- and clears -- whatever any
 * stream has raised.
 * Key semantics: a status word (one 64-byte line of pinned memory) and the poll buffers belong to
 * the PAIR (device, hipStream_t value) and live as long as the process: the library cannot see a
 * stream being destroyed.  A new stream that the runtime gives the handle value of a destroyed one
 * inherits that key -- including a failure the destroyed stream never had reported -- so a caller
 * that destroys streams calls hbk_sync_check_stream(stream) (or hbk_sync_check()) before
 * hipStreamDestroy; frameworks with a fixed set of compute streams (TensorFlow, torch) never get
 * there.  The stream to name is the one the failed ENTRY CALL was made on: work the library put on
 * its own helper streams (the backward's launch groups, the sharded plan's prefetch) reports to
 * that caller stream, and asking with any other stream reports nothing. */
int hbk_sync_check_stream(hbk_stream_t stream);
/* the kernels' divide-free floor-mod / floor-div (multiply-high by a
 * host-computed magic) evaluated on the host, so the integer arithmetic can be checked
 * against Python's % and // without a GPU.  d > 0. */
int64_t hbk_host_floormod_i64(int64_t v, int64_t d);
uint64_t hbk_host_fastdiv_u64(uint64_t n, uint64_t d);
/* the block -> work item mapping of the XCD-aware launches (every XCD takes one contiguous range
 * of a launch's work items; block b is observed to run on XCD b % 8), evaluated on the host: a
 * bijection of [0, n_blocks) for every n_blocks. */
int32_t hbk_host_xcd_contiguous(int32_t block, int32_t n_blocks);
/* CRC-32C (Castagnoli) of n host bytes, continuing from crc (0 starts a message): the checksum
 * of TensorFlow's tensor-bundle checkpoints (per tensor and per index block), for the host-side
 * reader / writer of that format (hybridbackend_amd/training/tf_bundle.py; the reference saves
 * through TF's bundle writer, hbtf/training/saver.py:97-185).  "123456789" -> 0xe3069283. */
uint32_t hbk_host_crc32c(uint32_t crc, const void* data, int64_t n);

/* ------------------------------------------------------------------------------------
 * R1  bucketize `feature % embedding_size` (TF FloorMod), N columns in one launch.
 *     docs/tutorial/ranking/data.py:179,186.  dtype HBK_INT32 | HBK_INT64.
 *     out may alias in.  buckets[c] > 0. */
int hbk_floormod_n(int32_t n_cols, int32_t dtype, const void* const* inputs,
                   const int64_t* lens, const int64_t* buckets, void* const* outputs,
                   hbk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * R2  HbPartitionByModulo / HbPartitionByModuloN
 *     ops: hbtf/distribute/partition/partition_by_modulo_ops.cc:46-60, :124-143
 *     semantics = the CPU functor (STABLE counting sort), partition_by_modulo_functors.cc:39-70:
 *       shard = ((v % P) + P) % P;  outputs[c] = ids grouped by shard, input order kept
 *       inside a shard;  sizes[c][p] = count;  indices[c][i] = position of inputs[c][i]
 *       in outputs[c]  (so outputs[c][indices[c]] == inputs[c]).
 *     dtype: HBK_INT32 | HBK_INT64 | HBK_UINT32 | HBK_UINT64.  n_cols = 1 covers the
 *     non-N op.  lens[c] < 2^31 (indices are int32, as in the reference).
 *     1 <= num_partitions <= 16384. */
size_t hbk_partition_workspace_bytes(int32_t n_cols, const int64_t* lens,
                                     int32_t num_partitions);
int hbk_partition_by_modulo_n(int32_t n_cols, int32_t dtype, int32_t num_partitions,
                              const void* const* inputs, const int64_t* lens,
                              void* const* outputs, int32_t* const* sizes,
                              int32_t* const* indices, void* workspace,
                              size_t workspace_bytes, hbk_stream_t stream);

/* R3  HbPartitionByDualModuloStage{One,Two}[N]
 *     ops: hbtf/distribute/partition/partition_by_dual_modulo_ops.cc:46-61,132-147,184-204,278-298
 *     functor: partition_by_dual_modulo_functors.cc:37-91
 *       pre = ((v % (P*M)) + P*M) % (P*M);  stage 1: shard = pre % P;  stage 2: shard = pre / M
 *     stage is 1 or 2; modulus >= 1. */
int hbk_partition_by_dual_modulo_n(int32_t n_cols, int32_t dtype, int32_t num_partitions,
                                   int32_t modulus, int32_t stage,
                                   const void* const* inputs, const int64_t* lens,
                                   void* const* outputs, int32_t* const* sizes,
                                   int32_t* const* indices, void* workspace,
                                   size_t workspace_bytes, hbk_stream_t stream);
/* Host-memory twins (all buffers in host memory, no workspace, no stream): the CPU kernels of
 * the non-N ops, which the reference registers for DEVICE_CPU
 * (partition_by_modulo_ops.cc:62-101, partition_by_dual_modulo_ops.cc:62-130; the N-ary CPU form
 * is Unimplemented there, partition_by_modulo_functors.cc:84-85 -- here it simply loops).  Same
 * results as the device entries, bit for bit. */
int hbk_partition_by_modulo_host(int32_t n_cols, int32_t dtype, int32_t num_partitions,
                                 const void* const* inputs, const int64_t* lens,
                                 void* const* outputs, int32_t* const* sizes,
                                 int32_t* const* indices);
int hbk_partition_by_dual_modulo_host(int32_t n_cols, int32_t dtype, int32_t num_partitions,
                                      int32_t modulus, int32_t stage,
                                      const void* const* inputs, const int64_t* lens,
                                      void* const* outputs, int32_t* const* sizes,
                                      int32_t* const* indices);

/* ------------------------------------------------------------------------------------
 * R6  fp32 <-> fp16 wire casts, N tensors in one launch.
 *     hbtf/common/cast.h:40-54, cast.cu.cc:37-42,60-65,84-95,287 (functor::Cast / CastN).
 *     (src,dst) = (HBK_FLOAT,HBK_HALF) round-to-nearest-even, or (HBK_HALF,HBK_FLOAT). */
int hbk_cast_n(int32_t n, int32_t src_dtype, int32_t dst_dtype, const void* const* inputs,
               const int64_t* lens, void* const* outputs, hbk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * R7  owner-side unique (TF `array_ops.unique`, hbtf/embedding/sharding.py:186):
 *     unique_out[c] = distinct ids in FIRST-OCCURRENCE order, index_out[c][i] = position of
 *     inputs[c][i] in unique_out[c], n_unique[c] (device int32) = number of distinct ids.
 *     ids int64.  unique_out[c] has capacity lens[c]. */
size_t hbk_unique_workspace_bytes(int32_t n_cols, const int64_t* lens);
int hbk_unique_n(int32_t n_cols, const int64_t* const* inputs, const int64_t* lens,
                 int64_t* const* unique_out, int32_t* const* index_out,
                 int32_t* const* n_unique, void* workspace, size_t workspace_bytes,
                 hbk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * R1+R7+R8+R9 fused:  HbGroupLookup (new, additive op; N-ary conventions of the
 * reference's Hb...N ops so Pack-style grouping can target it).  Per column c
 *     row(j)   = ids[j]                       (bucket == 0)
 *              = floormod(ids[j], bucket)     (bucket  > 0)            -- R1
 *     row(j)   = row(j) / divisor             (owner-side `// W`, sharding.py:189)
 *     out[s,:] = combine_{j in [row_splits[s], row_splits[s+1])} table[row(j), :]
 *   row_splits == NULL means one id per segment (Criteo scalar columns; the combiner is
 *   then the identity and out = table[row(ids)]).  Rows outside [0, rows) contribute
 *   zeros (TF GPU GatherV2 behaviour).  The unique/restore pair of the reference
 *   (sharding.py:186,193) is value-transparent in the forward and is not materialised.
 *   replaces: tf.nn.embedding_lookup_sparse as patched by
 *   hbtf/embedding/sharding.py:171-205 (local part) and docs/tutorial/ranking/data.py:179-193.
 *   fp32 tables, fixed in-order accumulation over j.                                   */
typedef struct {
  const float* table;        /* device [rows, dim] row-major */
  int64_t rows;
  int32_t dim;
  int32_t ids_dtype;         /* HBK_INT32 | HBK_INT64 */
  const void* ids;           /* device [n_ids] */
  int64_t n_ids;
  const int32_t* row_splits; /* device [n_segments + 1] or NULL */
  int64_t n_segments;        /* == n_ids when row_splits == NULL */
  int64_t bucket;            /* 0 = ids are row numbers already */
  int32_t divisor;           /* >= 1 */
  int32_t combiner;          /* HBK_COMBINER_* */
  float* out;                /* device [n_segments, dim] */
  /* optional segmented table (n_runs > 0): logical rows [run_start[k], run_start[k+1]) live
   * at table + run_base[k] floats (run_start[0] = 0).  Lets the stitch of the sharded
   * pipeline read the peer-major exchange buffer in place.  Device arrays [n_runs]. */
  const int64_t* run_start;
  const int64_t* run_base;
  int32_t n_runs;
  /* row stride of `out` in floats; 0 = dim.  Lets N columns write their blocks of one
   * concatenated [segments, sum of dims] tensor (hb.feature_column.DenseFeatures) in place. */
  int32_t out_stride;
  /* 1: skewed ids expected (Zipf heads): wide one-id-per-segment columns (dim >= 64, 16-byte
   * chunks, plain table of < 2^32 rows) go through 256-segment tiles that fetch every row
   * repeated inside a tile once and serve the repeats from LDS (config 4: 239 -> 215 us; on
   * uniform ids the tiles cost 15 %, hence a hint and not the default).  Option fwd_hot_rows = 1
   * sets it for every eligible column.  Other columns ignore it. */
  int32_t hot_rows;
  /* fp16 rows on one side (the embedding exchange's wire format fused into the kernels around
   * it; replaces the reference's separate cast passes, hbtf/common/cast.cu.cc:84-285):
   * HBK_LOOKUP_OUT_HALF: `out` holds fp16 rows (fp32 -> fp16 round to nearest even); one id per
   * segment, plain table.  HBK_LOOKUP_TABLE_HALF: `table` holds fp16 rows (sums stay fp32);
   * segmented tables only (n_runs > 0).  Offsets, strides and run bases count elements. */
  int32_t half_io;
  /* optional output permutation (round 5): segment s is written to row out_slots[s] of `out`
   * instead of row s (device int32 [n_segments]; one id per segment, plain table, fp32 rows).
   * The owner gather of the sharded step's p2p form stores every row straight into its place in
   * the requester's output this way.  NULL: rows in segment order. */
  const int32_t* out_slots;
} hbk_lookup_column_t;
#define HBK_LOOKUP_OUT_HALF 1
#define HBK_LOOKUP_TABLE_HALF 2

int hbk_group_lookup_fwd(int32_t n_cols, const hbk_lookup_column_t* cols,
                         hbk_stream_t stream);

/* R10  HbGroupLookupGrad: backward of the above up to the IndexedSlices the optimizer
 *   gets (SparseSegment*Grad -> UnsortedSegmentSum dup-reduction):
 *     unique_rows[c][u]  = the distinct row(j) of the column, in unspecified order   (int64)
 *     grad_rows[c][u,:]  = sum_{j: row(j) == unique_rows[u]} scale(seg(j)) * grad_out[seg(j),:]
 *     n_unique[c]        = u   (device int32; stays on the device, no host sync)
 *   scale = 1 (sum), 1/count (mean), 1/sqrt(count) (sqrtn).  unique_rows / grad_rows have
 *   capacity n_ids; rows >= n_unique are untouched.  Ids that map outside [0, rows) contribute
 *   nothing.  No global atomics on the data path: ids are grouped by a hash of their row and
 *   each group is reduced by one workgroup (LDS hash table of the distinct rows, sums in
 *   registers; a group holding a hot row is split over several workgroups and merged), so the
 *   summation order is not fixed: tolerance 1e-5 relative (option bwd_deterministic fixes it: every
 *   row's terms in id order, bit-equal to the sequential fp32 sum, unique_rows ascending -- see
 *   hbk_set_option).  unique_rows[c][0..u) are DISTINCT
 *   whatever the column holds: a group with more distinct rows than the workgroup's LDS table
 *   takes further passes over its pairs (rows handled in one pass are struck out), so every row
 *   is emitted -- and stepped by the fused optimizer -- exactly once.
 *   apply_lr != 0 additionally performs the sparse SGD update on the shard in the same
 *   pass: table[unique_rows[u],:] -= apply_lr * grad_rows[u,:] (sharded variables skip
 *   cross-rank aggregation, hbtf/training/gradient.py:193-217).  With apply_lr != 0 a column
 *   may pass unique_rows = grad_rows = NULL ("step only"): the rows are stepped where their sums
 *   sit in registers and no IndexedSlices are written (a third of the traffic, and no output range
 *   to claim); n_unique still receives the number of distinct rows.  The workspace query must be
 *   made with the same NULL / non-NULL pointers as the call.                            */
typedef struct {
  float* table;              /* device [rows, dim]; only touched when apply_lr != 0 */
  int64_t rows;
  int32_t dim;
  int32_t ids_dtype;
  const void* ids;
  int64_t n_ids;
  const int32_t* row_splits;
  int64_t n_segments;
  int64_t bucket;
  int32_t divisor;
  int32_t combiner;
  const float* grad_out;     /* device [n_segments, dim] */
  int64_t* unique_rows;      /* device [n_ids] */
  float* grad_rows;          /* device [n_ids, dim] */
  int32_t* n_unique;         /* device [1] */
  /* optional segmented inputs (n_runs > 0; row_splits must be NULL): ids j in
   * [run_start[k], run_start[k+1]) live at ids + run_ids[k] (elements) and their gradient rows
   * at grad_out + run_grads[k] (floats, 16-byte aligned).  Lets the owner side of the sharded
   * backward read the peer-major exchange buffers in place.  Device arrays [n_runs]. */
  const int64_t* run_start;
  const int64_t* run_ids;
  const int64_t* run_grads;
  int32_t n_runs;
  int32_t grad_stride;       /* row stride of grad_out in floats; 0 = dim (see out_stride) */
  float* accum;              /* Adagrad accumulator [rows, dim] (HBK_APPLY_ADAGRAD), else NULL */
  /* floats between consecutive rows of `table` AND of `accum` (round 6); 0 = dim.  Lets a caller
   * keep weights and accumulator interleaved row by row (table = buf, accum = buf + dim,
   * table_pitch = 2 dim): a dim-16 row's weights and accumulator then share ONE 128-byte line, the
   * Adagrad step fetches it with one request instead of two.  Only the optimizer step reads it
   * (the forward takes contiguous rows): measured on the config-5 shape in DESIGN.md 4.4. */
  int32_t table_pitch;
  /* HBK_GRAD_DETERMINISTIC: this column's rows are summed in id order -- what option
   * bwd_deterministic = 1 does for every column of every call (and the option, when set, still does):
   * the reproducible mode chosen per call instead of per process.  Other bits: 0. */
  int32_t flags;
} hbk_lookup_grad_column_t;
#define HBK_GRAD_DETERMINISTIC 1

size_t hbk_group_lookup_bwd_workspace_bytes(int32_t n_cols,
                                            const hbk_lookup_grad_column_t* cols);
int hbk_group_lookup_bwd(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                         float apply_lr, void* workspace, size_t workspace_bytes,
                         hbk_stream_t stream);
/* The same with the optimizer named: HBK_APPLY_SGD (above), or HBK_APPLY_ADAGRAD =
 *   accum[row,:] += g * g;  table[row,:] -= apply_lr * g * (1 / sqrt(accum[row,:]))
 * on the deduplicated gradient g of every touched row -- tf.train.AdagradOptimizer's sparse
 * apply, the optimizer of the reference's Taobao tutorials (docs/tutorial/ranking/taobao/
 * train.py:115); cols[c].accum holds the accumulator (initial_accumulator_value filled in by
 * the caller). */
#define HBK_APPLY_SGD 0
#define HBK_APPLY_ADAGRAD 2
int hbk_group_lookup_bwd_apply(int32_t n_cols, const hbk_lookup_grad_column_t* cols,
                               int32_t apply, float apply_lr, void* workspace,
                               size_t workspace_bytes, hbk_stream_t stream);

/* R10 (sharded form)  d(stitch + combiner): the transpose of the requester-side
 *   `gather(embeddings, shard_index)` + combiner (hbtf/embedding/sharding.py:200; TF emits
 *   SparseSegment*Grad followed by an UnsortedSegmentSum over a permutation):
 *     grad_rows[index[j], :] = scale(seg(j)) * grad_out[seg(j), :]
 *   `index` = the `indices` output of the forward partition (a permutation of 0..n_ids-1), so
 *   each row of grad_rows [n_ids, dim] is written exactly once.  The result is what travels
 *   back through the reverse alltoallv (hbtf/distribute/collective.py:334-347).            */
typedef struct {
  int32_t dim;
  int32_t combiner;
  int64_t n_ids;
  const int32_t* index;       /* device [n_ids] */
  const int32_t* row_splits;  /* device [n_segments + 1] or NULL */
  int64_t n_segments;
  const float* grad_out;      /* device [n_segments, dim] */
  float* grad_rows;           /* device [n_ids, dim] */
  /* optional segmented destination (n_runs > 0): rows [run_start[k], run_start[k+1]) of
   * grad_rows live at grad_rows + run_base[k] floats (the peer-major send buffer of the
   * reverse exchange; same tables as the forward's hbk_lookup_column_t runs).  Device arrays. */
  const int64_t* run_start;
  const int64_t* run_base;
  int32_t n_runs;
  int32_t grad_stride;        /* row stride of grad_out in floats; 0 = dim */
} hbk_stitch_grad_column_t;

int hbk_group_stitch_bwd(int32_t n_cols, const hbk_stitch_grad_column_t* cols,
                         hbk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * R11 HbLookup: cache probe.  op hbtf/embedding/lookup_ops.cc:38-58; kernel
 *     hbtf/embedding/lookup_functors.cu.cc:54-149; hash hybridbackend/common/murmur3.cu.h:32-77.
 *     slab = murmur3_hash32(key) % slab_count; a slab is `slab_size` consecutive int64 keys
 *     (the reference fixes 32 = its warp; here 1..64, one wave64 probes a slab with one
 *     ballot); first matching slot -> hit; an EMPTY (INT64_MIN) slot in the slab -> miss;
 *     else next slab (linear, wrapping).
 *     hit_slot[i] = cache index (slab*slab_size+slot) or -1 for a miss -- a per-key result
 *     in key order (the reference's compacted hit/miss lists follow from it; its own
 *     slicing is inconsistent).  n_miss: device int32, may be NULL.          */
int hbk_cache_probe(const int64_t* keys_cache, int64_t slab_count, int32_t slab_size,
                    const int64_t* keys, int64_t n_keys, int64_t* hit_slot,
                    int32_t* n_miss, hbk_stream_t stream);
/* The op's four outputs (lookup_ops.cc:38-58), key order kept inside each list (the reference
 * fills them through atomic counters, i.e. in no particular order):
 *   hit_keys_indices[h]  = i of the h-th key found        hit_cache_indices[h] = its cache index
 *   miss_keys_indices[m] = i of the m-th key not found    miss_keys[m]         = keys[i]
 * All four have capacity n_keys; counts (device int32[2]) = {n_hit, n_miss}.  The op sizes its
 * outputs from counts after one host sync (lookup_ops.cc:118-121); the C ABI leaves that to the
 * caller. */
size_t hbk_cache_lookup_workspace_bytes(int64_t n_keys);
int hbk_cache_lookup(const int64_t* keys_cache, int64_t slab_count, int32_t slab_size,
                     const int64_t* keys, int64_t n_keys, int32_t* hit_keys_indices,
                     int64_t* hit_cache_indices, int32_t* miss_keys_indices, int64_t* miss_keys,
                     int32_t* counts, void* workspace, size_t workspace_bytes,
                     hbk_stream_t stream);
/* test hook: murmur3_hash32<int64, 0> of each key, on device */
int hbk_murmur3_hash32(const int64_t* keys, int64_t n_keys, uint32_t* out,
                       hbk_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Communicator lifecycle: HbGetNcclId / HbCreateNcclCollective /
 * HbIsNcclCollectiveInitialized / async-error polling.
 *     hbtf/distribute/nccl/nccl_get_id.cc:35-62, nccl_create.cc:45-132,
 *     nccl_collective.cc:434-465.  RCCL over xGMI; one communicator + one private comm
 *     stream per handle (hbtf/distribute/nccl/collective.h:41-126).
 *     The 128-byte id travels between ranks by whatever the host has (the reference uses
 *     a TF gRPC broadcast, hbtf/distribute/rpc.py:88-124).                              */
typedef struct hbk_comm* hbk_comm_t;
#define HBK_COMM_ID_BYTES 128
int hbk_comm_get_id(uint8_t id[HBK_COMM_ID_BYTES]);
/* RCCL version codes (major * 10000 + minor * 100 + patch): the headers the library was built
 * against and the RCCL the process actually runs (a framework may have loaded its own first).
 * hbk_comm_create refuses a different MAJOR version; minor skew is fine for the calls used. */
int hbk_comm_rccl_versions(int32_t* built, int32_t* runtime);
int hbk_comm_create(hbk_comm_t* comm, const uint8_t id[HBK_COMM_ID_BYTES],
                    int32_t world_size, int32_t local_size, int32_t rank);
int hbk_comm_destroy(hbk_comm_t comm);
/* 0 = healthy; on an async RCCL error aborts the communicator and returns HBK_INTERNAL */
int hbk_comm_check_async(hbk_comm_t comm);
int hbk_comm_world_size(hbk_comm_t comm);
int hbk_comm_rank(hbk_comm_t comm);
/* ncclCommCount of the RCCL communicator behind the handle: the ranks RCCL itself connected
 * (0: custom transport, -1: error).  Reported by bench.py as `rccl_ranks_seen`. */
int hbk_comm_rccl_ranks(hbk_comm_t comm);
/* the communicator's private stream (a hipStream_t) */
hbk_stream_t hbk_comm_stream(hbk_comm_t comm);
/* R4  Collective::compute_active_ranks, hbtf/distribute/collective.h:80-112.
 *     Writes the peer list for `topology`, returns its length (<= world_size). */
int hbk_comm_active_ranks(hbk_comm_t comm, int32_t topology, int32_t* ranks_out);

/* R5  HbNcclAlltoall / HbNcclAlltoallN (equal split; also the sizes exchange that precedes
 *     every Alltoallv): hbtf/distribute/nccl/nccl_alltoall.cc:169-180,242-258,
 *     nccl_collective.cc:112-248.  counts[c] elements per tensor, divisible by the active
 *     size.  Enqueued on the communicator's stream, fenced after `compute_stream`'s
 *     current tail and before its future work (hbtf/common/stream.cc:83-142).          */
int hbk_alltoall_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t topology,
                   const void* const* inputs, const int64_t* counts, void* const* outputs,
                   hbk_stream_t compute_stream);

/* R5/R6  HbNcclAlltoallv / HbNcclAlltoallvN payload exchange.
 *     ops hbtf/distribute/nccl/nccl_alltoallv.cc:200-223,359-387; arithmetic
 *     nccl_collective.cc:250-384: chunk i of inputs[c] starts at
 *     sum_{j<i} send_sizes[c][j]*common_sizes[c] elements, the chunk from peer i lands at
 *     sum_{j<i} recv_sizes[c][j]*common_sizes[c].  Offsets are 64-bit here (the reference's
 *     int32 byte offsets overflow past 2 GiB, nccl_collective.cc:261-262).
 *     send_sizes / recv_sizes are HOST arrays [n][active] (rows); recv_sizes must already
 *     be known (use hbk_alltoall_n on the sizes first -- once for all columns -- as
 *     HbNcclAlltoallvN does, nccl_alltoallv.cc:418-564).
 *     wire_dtype HBK_HALF with dtype HBK_FLOAT casts to fp16 before the send and back
 *     after the receive (collective.py:291-296); `wire_ws` then needs
 *     hbk_alltoallv_wire_workspace_bytes().  wire_dtype == dtype otherwise.             */
size_t hbk_alltoallv_wire_workspace_bytes(int32_t n, const int64_t* common_sizes,
                                          const int32_t* send_sizes,
                                          const int32_t* recv_sizes, int32_t active);
int hbk_alltoallv_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t wire_dtype,
                    int32_t topology, const int64_t* common_sizes,
                    const void* const* inputs, const int32_t* send_sizes,
                    void* const* outputs, const int32_t* recv_sizes, void* wire_ws,
                    size_t wire_ws_bytes, hbk_stream_t compute_stream);

/* ------------------------------------------------------------------------------------
 * SURVEY 8f-1  aggregation of replicated gradients (hbtf/training/gradient.py:119-177):
 *   HbNcclAllreduce / HbNcclAllreduceN / HbNcclAllreduceMergedN  (nccl_allreduce.cc:31-260)
 *     outputs[c] = reduce over ranks of inputs[c] (reduce_op 0 SUM, 1 PROD, 2 MAX, 3 MIN), then
 *     multiplied by `scale` (fp32 only; 1/W = the `_mean` of gradient.py:77-99 fused in).  The N
 *     tensors travel as ONE bucket (pack -> one ncclAllReduce -> unpack); in place allowed.
 *   HbNcclAllgatherv  (nccl_allgatherv.cc:31-120)
 *     output = inputs of ranks 0..W-1 concatenated; counts[r] (host, elements) = what rank r
 *     contributes.  The reference op exchanges the counts itself and syncs the host; here the
 *     caller obtains them (one hbk_alltoall_n of its own count). */
size_t hbk_allreduce_workspace_bytes(int32_t n, const int64_t* counts, int32_t dtype);
int hbk_allreduce_n(hbk_comm_t comm, int32_t n, int32_t dtype, int32_t reduce_op,
                    const void* const* inputs, const int64_t* counts, void* const* outputs,
                    float scale, void* workspace, size_t workspace_bytes,
                    hbk_stream_t compute_stream);
int hbk_allgatherv(hbk_comm_t comm, int32_t dtype, const void* input, const int64_t* counts,
                   void* output, hbk_stream_t compute_stream);
/* HbNcclBroadcast (hbtf/distribute/nccl/nccl_broadcast.cc:31-92): every rank ends with the root's
 * `count` elements in `output`; `input` is read on the root only (input == output is allowed).
 * HbNcclAllgather (nccl_allgather.cc:31-101, equal counts) is hbk_allgatherv with every count the
 * same. */
int hbk_broadcast(hbk_comm_t comm, int32_t dtype, const void* input, void* output, int64_t count,
                  int32_t root, hbk_stream_t compute_stream);

/* A communicator over a caller-provided transport instead of RCCL (the reference's Collective is
 * an abstract class with NCCL as one implementation, hbtf/distribute/collective.h:70-201).  Every
 * collective above, and the sharded pipeline below, then moves its data through these callbacks
 * with the same chunk / offset arithmetic as over RCCL:
 *   exchange   chunk k of sendbuf (send_off[k], send_len[k] elements of esize bytes) goes to
 *              ranks[k], the chunk from ranks[k] lands at recv_off[k] elements of recvbuf; all on
 *              `stream`; with skip_self the chunk for `rank` itself is left alone
 *   allreduce  out[i] = reduce over the world's ranks of in[i]  (may be NULL: no allreduce)
 *   destroy    called by hbk_comm_destroy (may be NULL)
 * tests/support builds an in-process world on it (host threads sharing one GPU, device copies);
 * that code is not part of this library. */
typedef struct {
  void* ctx;
  int (*exchange)(void* ctx, int32_t rank, const int32_t* ranks, int32_t n_ranks,
                  const void* sendbuf, const int64_t* send_off, const int64_t* send_len,
                  void* recvbuf, const int64_t* recv_off, size_t esize, int32_t skip_self,
                  hbk_stream_t stream);
  int (*allreduce)(void* ctx, int32_t rank, int32_t world_size, int32_t dtype, int32_t reduce_op,
                   const void* in, void* out, int64_t count, hbk_stream_t stream);
  void (*destroy)(void* ctx);
} hbk_transport_t;
int hbk_comm_create_custom(hbk_comm_t* comm, const hbk_transport_t* transport,
                           int32_t world_size, int32_t local_size, int32_t rank);

/* ------------------------------------------------------------------------------------
 * R12  The whole sharded pipeline of hbtf/embedding/sharding.py:171-205 for N columns in one
 *   call per direction (what the reference's Pack pass + executor run as ~10 ops per column):
 *     forward : bucketize -> partition by id mod W -> alltoallv(ids) -> owner gather (`// W`)
 *               -> alltoallv(rows, fp32 | fp16 wire) -> stitch + combiner
 *     backward: d(stitch+combiner) -> reverse alltoallv (forward's sizes,
 *               hbtf/distribute/collective.py:334-347) -> duplicate-row reduction on the owner
 *               (+ fused SGD on the shard when apply_lr != 0)
 *   One [N x W] size exchange and ONE host sync per forward for all columns (the reference
 *   syncs once per op: nccl_alltoallv.cc:316,533); none in the backward.  Every exchange
 *   is one message per peer (all columns of a peer travel together).  Exchange buffers whose
 *   size depends on the peers are owned by the plan and grow on demand.
 *   ids are int64.  outs[c] is [n_segments[c], dim] ([n_ids[c], dim] when row_splits[c] is
 *   NULL).  The backward differentiates the LAST forward of the plan; its outputs need
 *   capacity hbk_sharded_owned_ids(plan, c) rows (known after that forward).              */
typedef struct {
  const float* shard;   /* device [rows_local, dim]: rows r, r+W, r+2W.. of the logical table */
  int64_t rows_local;
  int32_t dim;
  int32_t combiner;
  int64_t bucket;       /* >0: ids are taken modulo this before the partition (R1) */
  float* accum;         /* Adagrad accumulator of the shard [rows_local, dim], or NULL */
  int32_t hot_rows;     /* != 0: skewed ids expected: the owner gather stages the rows repeated in a
                           tile in LDS (hbk_lookup_column_t.hot_rows; wide columns only) */
  int32_t dedup;        /* != 0: requester-side dedup -- every DISTINCT id of the column's batch is
                           sent once (the reference's tutorials do it in user code in front of the
                           lookup, docs/tutorial/ranking/data.py:180-182): unique over the partitioned
                           ids, exchanges sized by the distinct ids, the stitch reads the received
                           rows through inverse o shard_index, and the backward sums duplicate
                           positions on the requester before the reverse exchange.  Pays when ids
                           repeat inside a batch (Zipf) and the step is link-bound; costs the unique. */
} hbk_sharded_column_t;

/* Host arithmetic of the peer-major exchange buffers (pure host code, no device work): S is
 * [n_cols][world] (rows this rank requests from owner q), R is [world][n_cols] (rows requester q
 * asked this rank for).  Any output pointer may be NULL.  Offsets are in ids / floats. */
int hbk_sharded_layout(int32_t n_cols, int32_t world, const int32_t* dims, const int32_t* S,
                       const int32_t* R, int32_t* ids_send_peer, int32_t* ids_recv_peer,
                       int32_t* rows_send_peer, int32_t* rows_recv_peer, int64_t* req_id_off,
                       int64_t* req_row_off, int64_t* own_id_off, int64_t* own_row_off,
                       int64_t* col_shard_off);

typedef struct hbk_sharded* hbk_sharded_t;
int hbk_sharded_create(hbk_sharded_t* plan, hbk_comm_t comm, int32_t n_cols,
                       const hbk_sharded_column_t* cols, int32_t wire_dtype);
/* Replaces the hot_rows hints of the plan's columns (hbk_sharded_column_t.hot_rows, [n_cols]);
 * the next forward reads them.  (The host side derives them from the distinct rows / ids of the
 * last backward: hybridbackend_amd/embedding/sharded.py.) */
int hbk_sharded_set_hot_rows(hbk_sharded_t plan, const int32_t* hot_rows);
int hbk_sharded_destroy(hbk_sharded_t plan);
/* out_strides / grad_strides: NULL, or per column the row stride in floats of outs[c] /
 * grads[c] (0 = dim): the columns' blocks of one concatenated [segments, sum of dims] tensor. */
int hbk_sharded_lookup_fwd(hbk_sharded_t plan, const int64_t* const* ids, const int64_t* n_ids,
                           const int32_t* const* row_splits, const int64_t* n_segments,
                           float* const* outs, const int32_t* out_strides, hbk_stream_t stream);
/* The forward in two halves (round 5).  _begin: everything up to and including the owner-side
 * gather (the partition or its prefetched result, the step's one host wait, the id exchange, the
 * gather into the reply buffer -- in the p2p form into the requesters' outputs).  _end: rows
 * exchange + stitch + combiner into `outs`.  hbk_sharded_lookup_fwd = _begin + _end.  Two plans
 * over ONE communicator that alternate  begin(B, step i + 1); end(A, step i)  put the ids of step
 * i + 1 on the wire AHEAD of the rows of step i: B gathers while A's rows travel, A stitches while
 * B's travel -- exchanges overlapped with the local gather across steps (hb.embedding.
 * PipelinedLookup).  Forward-only use: nothing may change the tables between a step's _begin and
 * its _end; every rank makes the same calls in the same order. */
int hbk_sharded_lookup_fwd_begin(hbk_sharded_t plan, const int64_t* const* ids,
                                 const int64_t* n_ids, const int32_t* const* row_splits,
                                 const int64_t* n_segments, hbk_stream_t stream);
int hbk_sharded_lookup_fwd_end(hbk_sharded_t plan, float* const* outs, const int32_t* out_strides,
                               hbk_stream_t stream);
/* Optional pipelining hint: partition + size exchange (stages 1-2) of a FUTURE step on the plan's
 * own stream, overlapping what the last forward still has in flight (its exchanges, gather and
 * stitch).  The next hbk_sharded_lookup_fwd with the same id pointers and counts uses it and
 * skips its own stages 1-2; any other forward drops it.  All ranks must prefetch the same steps;
 * the ids must stay valid and unchanged until that forward. */
int hbk_sharded_prefetch(hbk_sharded_t plan, const int64_t* const* ids, const int64_t* n_ids,
                         void* ids_ready_event /* hipEvent_t recorded after the ids were written,
                                                  or NULL when they are complete already */);
/* The same on a stream of the caller's (round 5): no stream of the plan is involved; the caller
 * enqueues it behind the end of the step BEFORE the last one begun on this plan (stream order is
 * the only protection the overwritten partition state gets).  Used by hb.embedding.PipelinedLookup
 * right behind a step's _begin. */
int hbk_sharded_prefetch_on(hbk_sharded_t plan, const int64_t* const* ids, const int64_t* n_ids,
                            hbk_stream_t stream);
int64_t hbk_sharded_owned_ids(hbk_sharded_t plan, int32_t column);

/* The p2p form of the forward (round 5).  hbk_sharded_p2p_bind registers this rank's N output
 * tensors ([n_ids[c], dim] fp32, out_strides as in hbk_sharded_lookup_fwd; NULL = dense) -- a
 * COLLECTIVE over the plan's communicator: every rank calls it with its own tensors, the addresses
 * are exchanged once and mapped into every peer (ranks of one process -- told apart by pid + host
 * boot id + a per-process random number, since pids repeat across containers: the pointer itself,
 * with hipDeviceEnablePeerAccess when they sit on different devices; other processes of the same
 * host: hipIpcGetMemHandle / hipIpcOpenMemHandle).  out_rows[c] = rows of outs[c]: a later forward
 * with more ids than that is refused (remote owners would store outside the tensor).  From then on hbk_sharded_lookup_fwd --
 * which must be handed exactly these tensors, one id per segment -- sends (id, output row) pairs and
 * the owner-side gather stores every row straight into its place in the requester's output: no
 * reply buffer, no rows Alltoallv, no stitch (one random-row pass instead of two; the rows cross the
 * link as the gather's own stores, so the gather IS the exchange); the ids still travel through the
 * communicator, and a one-int token per peer orders the requester behind the owners' stores.  The
 * exchange form stays the contract default (hbtf/embedding/sharding.py:171-205 composes the same
 * result); requester-side dedup and the fp16 wire are not available in this form.  The backward is
 * unchanged.  HBK_UNIMPLEMENTED on every rank when some peer's memory cannot be mapped; the plan
 * then keeps the exchange form.  hbk_sharded_p2p_unbind returns to it (not collective). */
int hbk_sharded_p2p_bind(hbk_sharded_t plan, float* const* outs, const int32_t* out_strides,
                         const int64_t* out_rows, hbk_stream_t stream);
int hbk_sharded_p2p_unbind(hbk_sharded_t plan);
/* diagnostics: host time of the plan's last forward in microseconds -- [0] enqueueing the
 * partition and the size exchange, [1] waiting for the sizes (the device, not host work),
 * [2] enqueueing everything else */
int hbk_sharded_last_host_us(hbk_sharded_t plan, float* out3);
int hbk_sharded_lookup_bwd(hbk_sharded_t plan, const float* const* grads,
                           const int32_t* grad_strides, float apply_lr,
                           int64_t* const* unique_rows, float* const* grad_rows,
                           int32_t* const* n_unique, hbk_stream_t stream);
/* the same with the optimizer named (HBK_APPLY_SGD | HBK_APPLY_ADAGRAD, see
 * hbk_group_lookup_bwd_apply); Adagrad uses the columns' `accum` shards.  unique_rows and grad_rows
 * may both be NULL when apply_lr != 0 (step only: no IndexedSlices are written). */
int hbk_sharded_lookup_bwd_apply(hbk_sharded_t plan, const float* const* grads,
                                 const int32_t* grad_strides, int32_t apply, float apply_lr,
                                 int64_t* const* unique_rows, float* const* grad_rows,
                                 int32_t* const* n_unique, hbk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HBK_H_ */
