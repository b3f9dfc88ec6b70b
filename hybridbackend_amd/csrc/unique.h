// Internal interface of the N-ary first-occurrence unique (unique.hip), shared with the
// backward pass (lookup_bwd.hip).
#ifndef HBK_CSRC_UNIQUE_H_
#define HBK_CSRC_UNIQUE_H_

#include "common.h"

namespace hbk {

struct UniqueColumn {
  const int64_t* in;     // device [len]
  int64_t len;
  int64_t* unique_out;   // device [len]
  int32_t* index_out;    // device [len]
  int32_t* n_unique;     // device [1]
  int32_t* multiplicity; // device [len] or NULL: multiplicity[u] = occurrences of unique_out[u]
};

size_t unique_workspace_bytes(int32_t n_cols, const int64_t* lens);

// Enqueues the whole unique on `stream`; `workspace` must hold unique_workspace_bytes().
int unique_n_impl(int32_t n_cols, const UniqueColumn* cols, void* workspace,
                  size_t workspace_bytes, hipStream_t stream);

}  // namespace hbk

#endif  // HBK_CSRC_UNIQUE_H_
