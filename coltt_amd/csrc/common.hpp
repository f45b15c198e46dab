// common.hpp — host-side plumbing shared by the HIP translation units of libcoltt_gpu.so:
// thread-local error string, opaque-handle registry, RAII device buffers, launch checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/coltt_gpu.h"

namespace coltt {

extern thread_local std::string g_last_error;

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define COLTT_HIP(expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return ::coltt::fail(_e == hipErrorOutOfMemory ? COLTT_E_NOMEM : COLTT_E_DEVICE, "%s: %s", \
                           #expr, hipGetErrorString(_e));                                        \
  } while (0)

#define COLTT_TRY(expr)        \
  do {                         \
    int _rc = (expr);          \
    if (_rc != COLTT_OK) return _rc; \
  } while (0)

// Device buffer that only grows; contents are preserved on growth when keep=true.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int reserve(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
    if (bytes <= cap) return COLTT_OK;
    size_t ncap = keep ? std::max(bytes, cap + cap / 2) : bytes;
    void* np = nullptr;
    COLTT_HIP(hipMalloc(&np, ncap));
    if (keep && p && cap) {
      hipError_t e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {  // the old buffer stays valid; do not leak the new one
        (void)hipFree(np);
        return ::coltt::fail(COLTT_E_DEVICE, "DevBuf::reserve copy: %s", hipGetErrorString(e));
      }
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = ncap;
    return COLTT_OK;
  }
  template <class T> T* as() const { return (T*)p; }
};

// Page-locked host staging of a search context.  A small call (the reference's RPC shape: one query in, k answers out) through
// pageable user buffers costs one staged, blocking copy per array — four D2H copies were a third of a single-query call.  Small
// inputs are copied into this buffer and sent with one asynchronous H2D; the answers come back packed in ONE D2H.
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  int reserve(size_t bytes) {
    if (bytes <= cap) return COLTT_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    COLTT_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return COLTT_OK;
  }
  template <class T> T* as() const { return (T*)p; }
};
constexpr size_t SMALL_CALL_BYTES = 64 * 1024;   // queries and answers up to this size take the staged path

// Measurement / test knobs (COLTT_* environment variables).  They are read ONCE — when the library is first used — into a process-wide
// snapshot; no search call touches the environment (~10 getenv per call).  A program that changes a knob afterwards
// calls coltt_policy_reload() (the Python binding does it when it sees the environment change; tests and tools toggle knobs that way).
struct Policy {
  bool flat_one = true;        // COLTT_FLAT_ONE=0: <= 4-query FLAT searches through the scan + select chain
  bool staging = true;         // COLTT_STAGING=0: no page-locked staging of small host-buffer calls
  bool ev8 = true;             // COLTT_EV8=0: level-0 distances of the throughput kernels from the pair-owned core even over line-transposed rows
  bool f8_mfma = true;         // COLTT_F8_MFMA=0: new "f8" cosine stores keep no binary16 copy: their batches run the exact scan (read at create)
  int rows8 = 1;               // COLTT_ROWS8 (read at create): 0 = new indexes store their rows in natural order, 1 = line-transposed for dim >= 256 (default), 2 = for any shape rows8.hpp covers
  int visg = -1;               // COLTT_VISG: -1 default (byte map above ef 128), 0 LDS hash, 1 byte map
  int walk2 = 7;               // COLTT_WALK2: -1 off, else OPT bits | 8 deep profile
  int walk2_lds = 4;           // COLTT_WALK2_LDS: -1 off, 2 / 4 / 6
  int bloom_kb = 0;            // COLTT_BLOOM_KB: 0 = sized by the occupancy budget
  int waves_per_cu = 0;        // COLTT_WAVES_PER_CU: 0 = per row format
  int rows_nt = -1;            // COLTT_ROWS_NT: non-temporal row loads in the eight-lane walks: -1 = by the size of the row array (default), 0 never, 1 always
  long long rows_nt_min_mb = 12288;  // COLTT_ROWS_NT_MIN_MB: ... row arrays of at least this many MiB (see exact.hpp: row_ld; measured crossover: profiles/r06ag_nt_rows_ab.md)
  int pq_waves = 0;            // COLTT_PQ_WAVES: resident traversals per CU of the product-quantised walk, 0 = the default cap
  bool pq_nbr = true;          // COLTT_PQ_NBR=0: the product-quantised walk gathers its code rows by neighbour slot (round 5) instead of reading the neighbourhood blocks
  bool lat_seq = false;        // COLTT_LAT_SEQ=1
  int lat_helpers = 0;         // COLTT_LAT_HELPERS: cache-warming helper workgroups per walking workgroup of the latency kernel (batches <= 8 queries); 0 = none (default: measured 3-5 % SLOWER, profiles/r06f_latency_helpers_ab.md)
  bool lat_knob_set = false;   // COLTT_LAT_MAX_NQ / COLTT_MW_MAX_NQ present
  uint32_t lat_max_nq = 0;     // ... and its value
  long long visg_budget_mb = -1;  // COLTT_VISG_BUDGET_MB (test knob)
};
Policy policy();               // a copy of the current snapshot
inline bool small_call_staging() { return policy().staging; }

// Locking discipline = the reference's (RWMutex per shard / per vertex level: edge/none_vectorstore.go:40, core/vectorindex/hnsw.go:51,
// hnsw_vertex.go:39): searches hold the object's lock SHARED and run concurrently, each on its own stream with its own
// workspaces (CtxPool below); Insert / Remove / Load / upsert hold it EXCLUSIVE.  Mutations complete (stream synchronised)
// before they release the lock, so a search that starts afterwards sees them — the device mirror never needs an epoch.
struct Object {
  std::shared_mutex rw;
  int device = 0;  // the HIP device this object's memory lives on; every entry point selects it for the calling thread
  virtual ~Object() {}
};
typedef std::shared_lock<std::shared_mutex> ReadLock;
typedef std::unique_lock<std::shared_mutex> WriteLock;

// Pool of per-call search contexts (stream + events + workspaces).  C needs `int init()`.  At most max_ctx contexts exist;
// further callers wait for one to come back (COLTT_MAX_SEARCH_CTX, default 32).
size_t max_search_ctx();
template <class C> class CtxPool {
 public:
  C* acquire() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      if (!idle_.empty()) { C* c = idle_.back(); idle_.pop_back(); return c; }
      if (all_.size() < max_search_ctx()) {
        auto c = std::make_unique<C>();
        if (c->init() != COLTT_OK) return nullptr;  // message already in g_last_error
        C* p = c.get(); all_.push_back(std::move(c)); return p;
      }
      cv_.wait(lk);
    }
  }
  void release(C* c) { { std::lock_guard<std::mutex> lk(mu_); idle_.push_back(c); } cv_.notify_one(); }
  size_t created() { std::lock_guard<std::mutex> lk(mu_); return all_.size(); }
 private:
  std::mutex mu_; std::condition_variable cv_;
  std::vector<std::unique_ptr<C>> all_; std::vector<C*> idle_;
};
template <class C> struct CtxLease {
  CtxPool<C>& pool; C* c;
  explicit CtxLease(CtxPool<C>& p) : pool(p), c(p.acquire()) {}
  ~CtxLease() { if (c) pool.release(c); }
  CtxLease(const CtxLease&) = delete; CtxLease& operator=(const CtxLease&) = delete;
};

class Registry {
 public:
  static Registry& get();
  coltt_handle_t add(std::shared_ptr<Object> o);
  std::shared_ptr<Object> find(coltt_handle_t h);
  bool erase(coltt_handle_t h);

 private:
  std::mutex mu_;
  std::unordered_map<coltt_handle_t, std::shared_ptr<Object>> map_;
  coltt_handle_t next_ = 0x1000;
};

template <class T> std::shared_ptr<T> lookup(coltt_handle_t h) {
  return std::dynamic_pointer_cast<T>(Registry::get().find(h));
}

int ensure_device();          // selects the process default device (coltt_init) for the calling thread
int default_device();         // that device's index (after ensure_device succeeded)
int use_device(int device);   // selects an object's device for the calling thread (worker threads of the library only)

// Entry points run on the CALLER's thread (a cgo call, a torch program's Python thread), whose current HIP device is the caller's
// business: the object's device is selected for the duration of the call and the caller's device is put back on every exit path.
class DeviceScope {
 public:
  DeviceScope() = default;
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
  ~DeviceScope() { if (prev_ >= 0 && prev_ != cur_) (void)hipSetDevice(prev_); }
  int enter(int device) {  // device < 0: the process default (coltt_init)
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (device < 0) { COLTT_TRY(ensure_device()); cur_ = default_device(); return COLTT_OK; }
    COLTT_TRY(use_device(device));
    cur_ = device;
    return COLTT_OK;
  }
  int device() const { return cur_; }
 private:
  int prev_ = -1, cur_ = -1;
};
#define COLTT_DEVICE(dev) ::coltt::DeviceScope coltt_dev_scope_; COLTT_TRY(coltt_dev_scope_.enter(dev))

// stores / indexes on an explicit device (group.hip places one collection shard per GPU)
int flat_create_on(int device, uint32_t dim, int metric, int quant, coltt_handle_t* out);
int hnsw_create_on(int device, uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg, coltt_handle_t* out);

// the product quantiser's kernels for an index that carries row-major codes of its own (pq.hip -> hnsw.hip, product-quantised HNSW)
struct PqShape { uint32_t dim = 0, m = 0, C = 0, dsub = 0; int metric = 0; };
int pq_snapshot(coltt_handle_t pq, PqShape* shape, DevBuf* cb_out, hipStream_t s);
int pq_encode_rowmajor(hipStream_t s, const float* d_cb, const PqShape& sh, const float* d_vecs, uint64_t n, uint8_t* d_codes, uint32_t row_bytes);
// the walk's tables: [nq][mp][1 << shift] binary16 (round to nearest even: the f16 codec's rounding), rows j >= m and entries c >= C are +0.0
int pq_lut16_batch(hipStream_t s, const float* d_cb, const PqShape& sh, const float* d_queries, size_t nq, uint32_t mp, uint32_t shift, uint32_t* d_qmax, unsigned short* d_lut);
int pq_centroid_norm_max(hipStream_t s, const float* d_cb, const PqShape& sh, uint32_t* d_scratch, float* out);

inline size_t quant_bytes(int q) { return q == COLTT_Q_NONE ? 4 : (q == COLTT_Q_F8 ? 1 : 2); }

// Quantisation dispatch.  The reference's "BF16" is IEEE binary16 (pkg/compresshelper/bf16.go:233-317 is float16.go:237-321
// with the names changed), so COLTT_Q_BF16 deliberately shares the Q_F16 device code; anything else is an error, never a
// silent fall-through (edge/vectorstore.go:79 "not support quantization type").
#define COLTT_DISPATCH_QUANT(q, F)                                                                  \
  switch (q) {                                                                                      \
    case COLTT_Q_NONE: F(::coltt::dev::Q_NONE); break;                                              \
    case COLTT_Q_F8: F(::coltt::dev::Q_F8); break;                                                  \
    case COLTT_Q_F16: case COLTT_Q_BF16: F(::coltt::dev::Q_F16); break;                             \
    default: return ::coltt::fail(COLTT_E_UNSUPPORTED, "not support quantization type %d", (int)(q)); \
  }
inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace coltt
