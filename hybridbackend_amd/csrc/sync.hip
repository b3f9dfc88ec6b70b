// Words that kernels of one launch use to wait for each other (partition.hip, unique.hip).
//
// A tile publishes count + 1 into its word and other tiles of the launch poll it: the words must
// read zero when the kernel starts.  They live in buffers of the library, one per (device,
// stream): launches on one stream are ordered, so a call takes one half of its stream's buffer
// and its first kernel clears what the call before it left in the other -- no launch is spent on
// a memset.  Buffers are a few hundred KB and are kept for the life of the process; a buffer that
// has become too small is replaced and the old one kept (work on the stream may still read it).
//
// Every wait is bounded (option sync_wait_ms, 2 s of the constant 100 MHz clock: far beyond any
// queue preemption, short of a hang).  A wave that gives up raises the host-visible status word
// AND the call's poison word in device memory: the later kernels of the same call read the poison
// first and leave, so nothing is computed from descriptors that were never written.  The host
// reports the failure once -- at the next entry call ON THE STREAM THE FAILED CALL WAS MADE ON (the
// status word is per (device, stream): a call on another stream neither sees nor consumes it), at
// hbk_sync_check_stream() / hbk_sync_check(), or in the SAME call where that call synchronises
// anyway (the sharded step) -- and from then on takes the multi-launch forms (option
// sync_onepass_off).  The one-launch forms are also refused when the
// device could not hold a whole column's workgroups at once (occupancy x CUs, cached).
#include <map>
#include <mutex>
#include <utility>
#include <vector>

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

namespace {
struct ChainState {
  std::mutex mu;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int cur = 0;
  bool have = false;     // ev[cur] holds a record
  bool seen = false;     // a launch has happened
  bool multi = false;    // launches have come from more than one stream
  hipStream_t last = nullptr;
};
ChainState* chain_of_device() {
  static std::mutex table_mu;
  static std::map<int, ChainState*> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(table_mu);
  auto it = table.find(dev);
  if (it != table.end()) return it->second;
  ChainState* c = new ChainState();
  if (hipEventCreateWithFlags(&c->ev[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev[1], hipEventDisableTiming) != hipSuccess) {
    delete c;
    c = nullptr;
  }
  table[dev] = c;
  return c;
}
}  // namespace

// A process that launches these kernels from ONE stream (the usual case) pays nothing: stream
// order already serialises them.  Events start when a second stream shows up: that launch waits
// for everything the first stream has been given so far (an event recorded on it now -- more than
// needed, never less), and from then on every such launch records an event behind itself.
SyncChain::SyncChain(hipStream_t stream) : stream_(stream), state_(chain_of_device()) {
  ChainState* c = reinterpret_cast<ChainState*>(state_);
  if (c == nullptr) return;
  c->mu.lock();
  if (!c->multi) {
    if (!c->seen || c->last == stream_) {
      c->seen = true;
      c->last = stream_;
      return;
    }
    c->multi = true;
    c->have = hipEventRecord(c->ev[c->cur], c->last) == hipSuccess;
    if (!c->have) (void)hipGetLastError();   // (the first stream is gone: nothing of it can be running)
  }
  if (c->have && c->last != stream_) (void)hipStreamWaitEvent(stream_, c->ev[c->cur], 0);
}

SyncChain::~SyncChain() {
  ChainState* c = reinterpret_cast<ChainState*>(state_);
  if (c == nullptr) return;
  if (c->multi) {
    c->cur ^= 1;
    c->have = hipEventRecord(c->ev[c->cur], stream_) == hipSuccess;
  }
  c->last = stream_;
  c->mu.unlock();
}

// Host-visible status words (pinned memory, one 64-byte line each): word 0 of the first page is the
// process-wide SUMMARY ("some word is raised": what every entry looks at, one volatile read); the
// others belong to one (device, owner stream) each and are handed out for the life of the process.
namespace {
constexpr int kWordsPerPage = 64;   // 64-byte lines of a 4 KB pinned page
struct StatusTable {
  std::mutex mu;
  std::vector<int32_t*> pages;
  int used = 0;                                              // lines handed out of the last page
  std::map<std::pair<int, hipStream_t>, int32_t*> words;     // (device, owner stream) -> its word
  int32_t* summary = nullptr;
  // a fresh line, or nullptr without pinned memory (mu held)
  int32_t* line() {
    if (pages.empty() || used == kWordsPerPage) {
      void* q = nullptr;
      if (hipHostMalloc(&q, kWordsPerPage * 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      memset(q, 0, kWordsPerPage * 64);
      pages.push_back(reinterpret_cast<int32_t*>(q));
      used = 0;
    }
    return pages.back() + 16 * used++;
  }
};
StatusTable& status_table() {
  static StatusTable* t = new StatusTable();   // (never destroyed: kernels may still write the words)
  return *t;
}
int32_t* summary_word() {
  StatusTable& t = status_table();
  std::lock_guard<std::mutex> lock(t.mu);
  if (t.summary == nullptr) t.summary = t.line();
  return t.summary;
}
int32_t* status_word(int dev, hipStream_t owner) {
  StatusTable& t = status_table();
  std::lock_guard<std::mutex> lock(t.mu);
  auto it = t.words.find(std::make_pair(dev, owner));
  if (it != t.words.end()) return it->second;
  int32_t* w = t.line();
  if (w != nullptr) t.words[std::make_pair(dev, owner)] = w;
  return w;
}
inline int32_t peek(const int32_t* w) { return *reinterpret_cast<const volatile int32_t*>(w); }
inline void poke(int32_t* w, int32_t v) { *reinterpret_cast<volatile int32_t*>(w) = v; }

int report(const char* who) {
  options().sync_onepass_off = 1;
  return fail(HBK_INTERNAL,
              "%s: a one-launch kernel (partition / unique / backward grouping) gave up waiting "
              "for the tiles of its column after %d ms; the outputs of the call it belonged to "
              "are not valid.  The library has switched to its multi-launch forms "
              "(option sync_onepass_off = 1)", who, options().sync_wait_ms);
}
}  // namespace

// Only the words of `stream` (all of them when `any`) are consumed; the summary is lowered first
// and raised again when another stream's word is still up -- a device that raises a word while
// this runs writes the word BEFORE the summary, so it either sees its summary survive or has its
// word found by the scan.
static int sync_check_impl(const char* who, bool any, hipStream_t stream) {
  StatusTable& t = status_table();
  {
    // fast path: nothing raised anywhere (one uncontended lock + one read of pinned memory)
    std::lock_guard<std::mutex> lock(t.mu);
    if (t.summary == nullptr || peek(t.summary) == 0) return HBK_OK;
  }
  int dev = 0;
  if (!any && hipGetDevice(&dev) != hipSuccess) return HBK_OK;
  bool mine = false, others = false;
  {
    std::lock_guard<std::mutex> lock(t.mu);
    poke(t.summary, 0);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    for (auto& kv : t.words) {
      if (peek(kv.second) == 0) continue;
      if (any || (kv.first.first == dev && kv.first.second == stream)) {
        poke(kv.second, 0);
        mine = true;
      } else {
        others = true;
      }
    }
    if (others) poke(t.summary, 1);
  }
  return mine ? report(who) : HBK_OK;
}

int sync_check(const char* who, hipStream_t stream) { return sync_check_impl(who, false, stream); }
int sync_check_any(const char* who) { return sync_check_impl(who, true, nullptr); }

namespace {
// can `max_column_wgs` workgroups of `kernel` be resident at once?  (cached per device and kernel)
bool fits_device(int dev, const void* kernel, int block, int max_column_wgs) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> capacity;
  std::lock_guard<std::mutex> lock(mu);
  auto it = capacity.find(std::make_pair(dev, kernel));
  if (it == capacity.end()) {
    int per_cu = 0, cus = 0;
    int cap = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
      cap = per_cu * cus;
    }
    it = capacity.emplace(std::make_pair(dev, kernel), cap).first;
  }
  // a column's workgroups wait while OTHER columns' workgroups occupy slots too: ask for twice
  return it->second >= 2 * max_column_wgs;
}
}  // namespace

bool sync_take(hipStream_t stream, size_t words, SyncTake* out, const void* kernel, int block,
               int max_column_wgs, hipStream_t owner) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, SyncSlot> slots;
  if (options().sync_onepass_off != 0) return false;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &capturing);
  if (capturing != hipStreamCaptureStatusNone) return false;   // a graph replays ONE launch
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  int32_t* const summary = summary_word();
  int32_t* const status = status_word(dev, owner == HBK_SYNC_SAME_STREAM ? stream : owner);
  if (summary == nullptr || status == nullptr) return false;
  if (kernel != nullptr && !fits_device(dev, kernel, block, max_column_wgs)) return false;
  words += 1;   // the call's poison word, behind its sync words
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
  out->status = status;
  out->summary = summary;
  out->poison = out->words + (words - 1);
  out->wait_ticks = (unsigned long long)(options().sync_wait_ms > 0 ? options().sync_wait_ms : 1) *
                    100000ull;
  out->withhold = options().sync_test_withhold;
  s.dirty_words[h] = words;
  s.dirty_words[1 - h] = 0;
  s.half = 1 - h;
  return true;
}

}  // namespace hbk
