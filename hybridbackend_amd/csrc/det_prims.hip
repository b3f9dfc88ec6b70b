// det_prims.hip -- the two library primitives of the DETERMINISTIC backward (option
// bwd_deterministic, lookup_bwd_det.h): a stable LSD radix sort of (key, value) pairs and an
// exclusive scan, both from rocPRIM (header-only, part of ROCm).  The hot path -- every kernel of
// the default backward -- is hand-written; the deterministic mode is a reproducibility tool (TF's
// TF_DETERMINISTIC_OPS analogue) whose cost is a full sort of the batch's pairs, and a generic,
// exhaustively tested sort is what that mode wants.  Compiled in a translation unit of its own so
// that rocPRIM's templates do not ride along with every rebuild of the kernels.
#include <hip/hip_runtime.h>
#include <cstring>   // (rocPRIM's texture iterator calls memset on the host)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "common.h"

namespace hbk {

// temp bytes of det_sort_pairs for n pairs (0 on error)
size_t det_sort_temp_bytes(size_t n, int end_bit) {
  size_t bytes = 0;
  const hipError_t e = rocprim::radix_sort_pairs(
      nullptr, bytes, static_cast<const uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
      static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0u,
      (unsigned)end_bit, hipStream_t(nullptr));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return bytes + 256;
}

// keys_out / vals_out = the pairs sorted by bits [0, end_bit) of their keys, STABLE: pairs with
// equal keys keep their input order (LSD radix sort) -- what turns "sorted by (column, row)" into
// "every row's terms in id order"
int det_sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                   const uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit,
                   hipStream_t stream) {
  if (n == 0) return HBK_OK;
  size_t bytes = temp_bytes;
  const hipError_t e = rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n,
                                                 0u, (unsigned)end_bit, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HBK_INTERNAL, "deterministic backward: radix sort failed: %s", hipGetErrorString(e));
  }
  return HBK_OK;
}

size_t det_scan_temp_bytes(size_t n) {
  size_t bytes = 0;
  const hipError_t e = rocprim::exclusive_scan(nullptr, bytes, static_cast<const int32_t*>(nullptr),
                                               static_cast<int32_t*>(nullptr), (int32_t)0, n,
                                               rocprim::plus<int32_t>(), hipStream_t(nullptr));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return bytes + 256;
}

// out[i] = sum of in[0 .. i)
int det_scan(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n,
             hipStream_t stream) {
  if (n == 0) return HBK_OK;
  size_t bytes = temp_bytes;
  const hipError_t e = rocprim::exclusive_scan(temp, bytes, in, out, (int32_t)0, n,
                                               rocprim::plus<int32_t>(), stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HBK_INTERNAL, "deterministic backward: scan failed: %s", hipGetErrorString(e));
  }
  return HBK_OK;
}

}  // namespace hbk
