// Words that kernels of one launch use to wait for each other (partition.hip, unique.hip).
//
// A tile publishes count + 1 into its word and other tiles of the launch poll it: the words must
// read zero when the kernel starts.  They live in buffers of the library, one per (device,
// stream): launches on one stream are ordered, so a call takes one half of its stream's buffer
// and its first kernel clears what the call before it left in the other -- no launch is spent on
// a memset.  Buffers are a few hundred KB and are kept for the life of the process; a buffer that
// has become too small is replaced and the old one kept (work on the stream may still read it).
//
// Every wait is bounded: a wave that has polled for kSyncWaitTicks of the 100 MHz clock raises the
// host-visible status word and gives up, and the next entry call reports HBK_INTERNAL instead of
// the device hanging.
#include <map>
#include <mutex>
#include <utility>

#include <string.h>

#include "common.h"

namespace hbk {
namespace {
struct SyncSlot {
  int32_t* buf = nullptr;
  size_t half_words = 0;
  int half = 0;                    // half the next call takes
  size_t dirty_words[2] = {0, 0};  // words a call has left set
};
}  // namespace

int32_t* sync_status() {
  static int32_t* word = [] {
    void* q = nullptr;
    if (hipHostMalloc(&q, 64, hipHostMallocDefault) != hipSuccess) return (int32_t*)nullptr;
    memset(q, 0, 64);
    return reinterpret_cast<int32_t*>(q);
  }();
  return word;
}

bool sync_raised() {
  int32_t* st = sync_status();
  return st != nullptr && *reinterpret_cast<volatile int32_t*>(st) != 0;
}

bool sync_take(hipStream_t stream, size_t words, SyncTake* out) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SyncSlot> slots;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &capturing);
  if (capturing != hipStreamCaptureStatusNone) return false;   // a graph replays ONE launch
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || sync_status() == nullptr) return false;
  std::lock_guard<std::mutex> lock(mu);
  SyncSlot& s = slots[std::make_pair(dev, stream)];
  if (s.half_words < words) {
    const size_t half = (words + words / 2 + 16384 + 63) / 64 * 64;
    void* q = nullptr;
    if (hipMalloc(&q, 2 * half * sizeof(int32_t)) != hipSuccess) return false;
    if (hipMemsetAsync(q, 0, 2 * half * sizeof(int32_t), stream) != hipSuccess) return false;
    s.buf = reinterpret_cast<int32_t*>(q);
    s.half_words = half;
    s.half = 0;
    s.dirty_words[0] = s.dirty_words[1] = 0;
  }
  const int h = s.half;
  out->words = s.buf + (size_t)h * s.half_words;
  out->zero = s.buf + (size_t)(1 - h) * s.half_words;
  out->zero_words = (int64_t)s.dirty_words[1 - h];
  out->status = sync_status();
  s.dirty_words[h] = words;
  s.dirty_words[1 - h] = 0;
  s.half = 1 - h;
  return true;
}

}  // namespace hbk
