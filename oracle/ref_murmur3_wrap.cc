// Thin extern "C" shim over the reference's OWN header, compiled where it lies
// (-I/root/reference): hybridbackend/common/murmur3.cu.h:32-77.  No reference source
// is copied into this repository; this file only instantiates the template.
#include <cstdint>

#include "hybridbackend/common/murmur3.cu.h"

extern "C" uint32_t ref_murmur3_hash32_i64(long long key) {
  return murmur3_hash32<long long>(key);
}

extern "C" void ref_murmur3_hash32_i64_n(const long long* keys, long long n,
                                         uint32_t* out) {
  for (long long i = 0; i < n; ++i) out[i] = murmur3_hash32<long long>(keys[i]);
}
