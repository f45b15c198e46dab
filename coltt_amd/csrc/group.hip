// group.hip — one collection partitioned over the GPUs of a node (BASELINE.json north_star).
//
// The reference has no multi-device code: its only "sharding" is the 16 in-process map shards chosen by
// sharding.ShardVertex (pkg/sharding/shard.go:34-41) that `highCpu` scans with local queues and then merges
// (edge/none_vectorstore.go:148-178).  A group is that shape stretched over devices:
//   * SHARD layout: vertex `id` lives on shard ShardVertex(id, world) — the same FNV-1a rule; each shard is an ordinary FLAT
//     store / HNSW index (flat.hip / hnsw.hip) on its own GPU; a query batch is searched by every shard on its own stream,
//     the per-shard top-k are exchanged with ONE all-gather of packed {u64 id, f32 score, u32 valid} records (RCCL over
//     xGMI: ncclCommInitAll in one process, ncclCommInitRank with one process per GPU) and merged on the host in the
//     canonical (score, id) order — exactly the local-queue-then-global-queue structure of the reference;
//   * REPLICA layout: every member holds the whole collection, a query batch is split across members, nothing is exchanged.
// RCCL is loaded at run time (dlopen) so the library has no link-time dependency on it; when all members of a
// single-process group sit on ONE device (tests on a 1-GPU box) RCCL refuses duplicate devices and the records travel
// through pinned host memory instead — that is a transport choice, the merge is on the host either way.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <thread>

#include "common.hpp"
#include "exact.hpp"

using namespace coltt;
using namespace coltt::dev;

namespace {

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------
typedef void* nccl_comm_t;
struct NcclId { char internal[128]; };
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*CommAbort)(nccl_comm_t) = nullptr;   // optional: tears a communicator down without waiting for its peers (after an exchange timeout)
  int (*AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already mapped (torch bundles one) is preferred: two RCCLs in one process is asking for trouble
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.so) break; }
    if (!r.so) for (const char* n : names) { r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
    if (!r.so) { r.err = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?"); return; }
#define COLTT_SYM(F, N) r.F = reinterpret_cast<decltype(r.F)>(dlsym(r.so, N)); if (!r.F) { r.err = std::string("librccl lacks ") + N; r.so = nullptr; return; }
    COLTT_SYM(GetUniqueId, "ncclGetUniqueId") COLTT_SYM(CommInitAll, "ncclCommInitAll") COLTT_SYM(CommInitRank, "ncclCommInitRank")
    COLTT_SYM(CommDestroy, "ncclCommDestroy") COLTT_SYM(AllGather, "ncclAllGather") COLTT_SYM(GroupStart, "ncclGroupStart")
    COLTT_SYM(GroupEnd, "ncclGroupEnd") COLTT_SYM(GetErrorString, "ncclGetErrorString")
#undef COLTT_SYM
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.so, "ncclCommAbort"));
  });
  return r.so ? &r : nullptr;
}
#define COLTT_NCCL(R, expr)                                                                                      \
  do { int _e = (expr); if (_e != 0) return fail(COLTT_E_DEVICE, "%s: %s", #expr, (R)->GetErrorString(_e)); } while (0)

// ---- packed per-shard answers ---------------------------------------------------------------------------------------------
// valid: bit 0 = the record holds an answer; bits 8..31 = the STATUS of the rank that packed it (0 = its shard search succeeded), the same in every
// record of the rank's block.  A rank whose stage A failed still takes part in the exchange — with a block of status records — so that no peer is left
// waiting in an all-gather that never comes and every rank returns the same error for the batch (round 6; core shape: edge/none_vectorstore.go:148-178,
// where a failing shard goroutine still signals its WaitGroup).
struct Rec { uint64_t id; float score; uint32_t valid; };  // 16 bytes
static_assert(sizeof(Rec) == 16, "Rec");
constexpr uint32_t REC_STATUS_SHIFT = 8;

__global__ void pack_topk_kernel(const uint64_t* __restrict__ ids, const float* __restrict__ sc, const uint32_t* __restrict__ cnt,
                                 uint32_t nq, uint32_t k, Rec* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nq * k) return;
  const uint32_t q = (uint32_t)(i / k), j = (uint32_t)(i - (size_t)q * k);
  const bool v = j < cnt[q];
  out[i] = Rec{v ? ids[i] : 0ull, v ? sc[i] : 0.f, v ? 1u : 0u};
}
// the block of a rank whose shard search failed: no answers, its status in every record
__global__ void pack_status_kernel(size_t n, uint32_t status, Rec* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = Rec{0ull, 0.f, status << REC_STATUS_SHIFT};
}

// ---- all-gather between the processes of one box through POSIX shared memory -------------------------------------------------
// Transport of COLTT_EXCHANGE_SHM groups (several processes, possibly on ONE device, where RCCL refuses to run).  One segment
// per group: a header of process-shared atomics + world slots of `cap` bytes.  A gather of generation g:
//   wait consumed >= g * world      (every rank has read generation g - 1: the slots may be overwritten)
//   copy my slots, len[rank] = bytes, written += n_local (release)
//   wait written >= (g + 1) * world (acquire), copy every slot out, consumed += n_local
// The counters only grow, so the barrier needs no reset and a late process can never confuse two generations.
struct ShmHdr {
  std::atomic<uint32_t> magic;      // set last by the creating process
  uint32_t world;
  uint64_t cap;                     // bytes per rank slot
  std::atomic<uint64_t> attached, written, consumed;
  std::atomic<uint32_t> failed;     // a process that gives up (timeout, size mismatch) releases the others
  uint64_t len[64];                 // bytes written by each rank in the current generation
};
constexpr uint32_t SHM_MAGIC = 0xC0177511u;
constexpr size_t SHM_HDR_BYTES = 4096;
static_assert(sizeof(ShmHdr) <= SHM_HDR_BYTES, "ShmHdr");

struct ShmExchange : Object {
  std::string name; int fd = -1; ShmHdr* hdr = nullptr; uint8_t* data = nullptr; size_t map_bytes = 0;
  int world = 0, n_local = 0, rank_base = 0; uint64_t cap = 0, gen = 0; bool creator = false, unlinked = false;
  double timeout_s = 120.0;
  ~ShmExchange() override {
    if (hdr) (void)munmap(hdr, map_bytes);
    if (fd >= 0) (void)close(fd);
    if (creator && !unlinked && !name.empty()) (void)shm_unlink(name.c_str());
  }
  template <class P> bool wait_for(P&& pred) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; spin++) {
      if (pred()) return true;
      if (hdr && hdr->failed.load(std::memory_order_acquire)) return false;
      if (spin < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spin & 255u) == 255u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    }
  }
};

int shm_open_exchange(const uint8_t* unique_id, int world, int n_local, int rank_base, uint64_t bytes_per_rank, std::shared_ptr<ShmExchange>& out) {
  if (!unique_id) return fail(COLTT_E_INVALID, "shm exchange: unique_id is NULL");
  if (world <= 0 || world > 64 || n_local <= 0 || rank_base < 0 || rank_base + n_local > world) return fail(COLTT_E_INVALID, "shm exchange: ranks [%d,%d) outside world %d (<= 64)", rank_base, rank_base + n_local, world);
  if (bytes_per_rank == 0) return fail(COLTT_E_INVALID, "shm exchange: bytes_per_rank is 0");
  auto x = std::make_shared<ShmExchange>();
  if (const char* e = getenv("COLTT_SHM_TIMEOUT_S")) { if (*e) x->timeout_s = std::max(0.1, atof(e)); }
  char nm[64]; int k = snprintf(nm, sizeof(nm), "/coltt_");
  for (int i = 0; i < 16; i++) k += snprintf(nm + k, sizeof(nm) - (size_t)k, "%02x", unique_id[i]);
  x->name = nm; x->world = world; x->n_local = n_local; x->rank_base = rank_base;
  x->cap = (bytes_per_rank + 63) & ~63ull;
  x->map_bytes = SHM_HDR_BYTES + (size_t)world * x->cap;
  x->fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (x->fd >= 0) {
    x->creator = true;
    if (ftruncate(x->fd, (off_t)x->map_bytes) != 0) return fail(COLTT_E_NOMEM, "shm exchange: ftruncate(%zu): %s", x->map_bytes, strerror(errno));
  } else if (errno == EEXIST) {
    x->fd = shm_open(nm, O_RDWR, 0600);
    if (x->fd < 0) return fail(COLTT_E_DEVICE, "shm exchange: shm_open(%s): %s", nm, strerror(errno));
    // the creator may not have sized the segment yet
    if (!x->wait_for([&] { struct stat st; return fstat(x->fd, &st) == 0 && (size_t)st.st_size >= x->map_bytes; }))
      return fail(COLTT_E_DEVICE, "shm exchange: %s never reached %zu bytes (do all processes pass the same world / bytes_per_rank?)", nm, x->map_bytes);
  } else return fail(COLTT_E_DEVICE, "shm exchange: shm_open(%s): %s", nm, strerror(errno));
  void* m = mmap(nullptr, x->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
  if (m == MAP_FAILED) return fail(COLTT_E_NOMEM, "shm exchange: mmap(%zu): %s", x->map_bytes, strerror(errno));
  x->hdr = static_cast<ShmHdr*>(m); x->data = static_cast<uint8_t*>(m) + SHM_HDR_BYTES;
  if (x->creator) {   // a fresh segment is zero-filled: counters start at 0
    x->hdr->world = (uint32_t)world; x->hdr->cap = x->cap;
    x->hdr->magic.store(SHM_MAGIC, std::memory_order_release);
  } else {
    if (!x->wait_for([&] { return x->hdr->magic.load(std::memory_order_acquire) == SHM_MAGIC; })) return fail(COLTT_E_DEVICE, "shm exchange: %s was never initialised by its creator", nm);
    if (x->hdr->world != (uint32_t)world || x->hdr->cap != x->cap)
      return fail(COLTT_E_INVALID, "shm exchange: this process asks for world %d / %llu B per rank, the segment was made for %u / %llu", world,
                  (unsigned long long)x->cap, x->hdr->world, (unsigned long long)x->hdr->cap);
  }
  x->hdr->attached.fetch_add((uint64_t)n_local, std::memory_order_acq_rel);
  if (!x->wait_for([&] { return x->hdr->attached.load(std::memory_order_acquire) >= (uint64_t)world; })) {
    x->hdr->failed.store(1, std::memory_order_release);
    return fail(COLTT_E_DEVICE, "shm exchange: only %llu of %d ranks attached to %s within %.0f s", (unsigned long long)x->hdr->attached.load(), world, nm, x->timeout_s);
  }
  if (x->creator) { (void)shm_unlink(nm); x->unlinked = true; }   // everybody holds a mapping: the name can go (nothing leaks if a process dies later)
  out = x;
  return COLTT_OK;
}

int shm_allgather(ShmExchange* x, const void* local, uint64_t bytes, void* out) {
  ShmHdr* h = x->hdr;
  if (bytes > x->cap) {   // the peers are (or will be) waiting for this rank: release them
    h->failed.store(1, std::memory_order_release);
    return fail(COLTT_E_INVALID, "shm allgather: %llu bytes per rank > the segment's %llu", (unsigned long long)bytes, (unsigned long long)x->cap);
  }
  const uint64_t g = x->gen, W = (uint64_t)x->world;
  auto give_up = [&](const char* what) { h->failed.store(1, std::memory_order_release); return fail(COLTT_E_DEVICE, "shm allgather (generation %llu): %s", (unsigned long long)g, what); };
  if (!x->wait_for([&] { return h->consumed.load(std::memory_order_acquire) >= g * W; })) return give_up("a peer never finished reading the previous generation");
  for (int i = 0; i < x->n_local; i++) {
    const int r = x->rank_base + i;
    if (bytes) std::memcpy(x->data + (size_t)r * x->cap, static_cast<const uint8_t*>(local) + (size_t)i * bytes, bytes);
    h->len[r] = bytes;
  }
  h->written.fetch_add((uint64_t)x->n_local, std::memory_order_acq_rel);
  if (!x->wait_for([&] { return h->written.load(std::memory_order_acquire) >= (g + 1) * W; })) return give_up("a peer never arrived (did every process make the same call?)");
  for (int r = 0; r < x->world; r++) {
    if (h->len[r] != bytes) return give_up("ranks disagree on the size of their contribution");
    if (bytes) std::memcpy(static_cast<uint8_t*>(out) + (size_t)r * bytes, x->data + (size_t)r * x->cap, bytes);
  }
  h->consumed.fetch_add((uint64_t)x->n_local, std::memory_order_acq_rel);
  x->gen++;
  return COLTT_OK;
}

struct Member {
  int device = 0, rank = 0;
  coltt_handle_t h = 0;
  hipStream_t stream = nullptr;    // ingest, replica searches
  hipStream_t cstream = nullptr;   // the exchange of a shard search: pack, all-gather, D2H — never the stream a search runs on
  nccl_comm_t comm = nullptr;
  DevBuf d_q, d_ids, d_sc, d_cnt;  // ingest / replica searches (under call_mu)
};

// A shard search is a three-stage pipeline ("issued on a comm stream and overlapped with the next batch"):
//   A  every member searches the batch on its shard and packs its top-k       (the caller's thread, one batch at a time: call_mu)
//   B  ONE all-gather of the packed top-k + D2H, on the comm streams          (the group's exchange thread, in ticket order)
//   C  the host-side final merge, split over a few host threads               (same thread, right behind B)
// Stage A of batch i+1 runs while B and C of batch i are in flight; a batch owns one Slot (answer arrays, packed records, the
// gathered block, its pinned staging) from A until its merge is done, so nothing is shared between batches in flight.
// Tickets are handed out in stage-A order and the exchange thread serves them FIFO: in a multi-process group every process makes the
// same calls in the same order, hence issues the same collectives in the same order.
struct SlotMember { DevBuf d_q, d_ids, d_sc, d_cnt, d_pack, d_gather; };
struct Slot {
  std::vector<std::unique_ptr<SlotMember>> mb;
  PinnedBuf h_stage;                                  // [world][nq][k] gathered records
  std::vector<Rec> h_local, h_chunk_in, h_chunk_out;  // SHM transport: this process's records / one chunk of queries in flight
};
struct Job {
  uint64_t ticket = 0; int slot = -1;
  size_t nq = 0; uint32_t k = 0; int nearest = 1;
  uint64_t* out_ids = nullptr; float* out_scores = nullptr; uint32_t* out_counts = nullptr;
  int rc = COLTT_OK; std::string err; bool done = false;
  int local_rc = COLTT_OK; std::string local_err;   // stage A failed HERE: the exchange still runs (status records), the error is returned at the end
  bool clean_failure = false;                         // a rank's status said so: the transport itself is intact (nothing to poison)
  double search_ms = 0, exchange_ms = 0, merge_ms = 0;
};
constexpr int GROUP_SLOTS = 3;

struct Group : Object {
  int kind = 0, layout = 0, exchange = 0 /* in use: 1 RCCL, 2 host */, world = 0, rank_base = 0;
  uint32_t dim = 0; int metric = 0, quant = 0;
  std::vector<std::unique_ptr<Member>> m;  // DevBuf is neither copyable nor movable
  std::mutex call_mu;  // stage A, ingest and remove: one at a time (members' own locks still protect them against direct use)
  std::shared_ptr<ShmExchange> shm;                  // COLTT_EXCHANGE_SHM
  // pipeline state
  Slot slots[GROUP_SLOTS];
  std::mutex q_mu; std::condition_variable q_cv, done_cv, slot_cv;
  std::deque<std::shared_ptr<Job>> queue;                       // stage B/C work, FIFO = ticket order
  std::unordered_map<uint64_t, std::shared_ptr<Job>> jobs;      // begun, not yet ended
  bool slot_busy[GROUP_SLOTS] = {false, false, false};
  uint64_t next_ticket = 1; bool stop = false; std::thread worker;
  std::atomic<bool> broken{false};   // an exchange timed out: the communicator's state is unknown, every later shard search fails at once
  // cumulative timing of finished batches (coltt_group_timing)
  uint64_t t_batches = 0; double t_search = 0, t_exchange = 0, t_merge = 0;
  void exchange_loop();
  int exchange_and_merge(Job& j);
  ~Group() override {
    { std::lock_guard<std::mutex> lk(q_mu); stop = true; }
    q_cv.notify_all();
    if (worker.joinable()) worker.join();
    Rccl* r = rccl();
    for (auto& xp : m) {
      Member& x = *xp;
      (void)hipSetDevice(x.device);
      if (x.comm && r) (void)r->CommDestroy(x.comm);
      if (x.stream) (void)hipStreamDestroy(x.stream);
      if (x.cstream) (void)hipStreamDestroy(x.cstream);
      if (x.h) { if (kind == COLTT_GROUP_FLAT) (void)coltt_flat_destroy(x.h); else (void)coltt_hnsw_destroy(x.h); }
    }
  }
};

// run f(i) for every local member on its own host thread (each selects its device); returns the first error
template <class F> int for_members(Group* g, F&& f) {
  const size_t n = g->m.size();
  std::vector<int> rc(n, COLTT_OK); std::vector<std::string> msg(n);
  auto body = [&](size_t i) { rc[i] = use_device(g->m[i]->device); if (rc[i] == COLTT_OK) rc[i] = f(i); if (rc[i] != COLTT_OK) msg[i] = g_last_error; };
  if (n == 1) body(0);
  else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; i++) th.emplace_back(body, i);
    for (auto& t : th) t.join();
  }
  for (size_t i = 0; i < n; i++) if (rc[i] != COLTT_OK) { g_last_error = "member " + std::to_string(i) + ": " + msg[i]; return rc[i]; }
  return COLTT_OK;
}

// canonical order of the merge = the order every shard already returns: ascending (score, id)
inline bool rec_less(const Rec& a, const Rec& b) { return a.score < b.score || (a.score == b.score && a.id < b.id); }

}  // namespace

// Host-side final merge: recs = [world][nq][k] packed per-shard answers, each shard's valid records ascending by (score, id).
// nearest: the k smallest of the union; otherwise (edge SELECT_REFERENCE) the k LARGEST — both returned ascending, as every single
// store returns them.  Queries are independent: [q_lo, q_hi) is one thread's share.
namespace {
void merge_range(const Rec* recs, int world, size_t nq, uint32_t k, int nearest, size_t q_lo, size_t q_hi, uint64_t* out_ids, float* out_scores,
                 uint32_t* out_counts) {
  const size_t per = nq * (size_t)k;
  std::vector<uint32_t> head((size_t)world), len((size_t)world);
  std::vector<Rec> tmp(k);
  for (size_t q = q_lo; q < q_hi; q++) {
    size_t total = 0;
    for (int s = 0; s < world; s++) {
      const Rec* r = recs + (size_t)s * per + q * k;
      uint32_t c = 0; while (c < k && (r[c].valid & 1u)) c++;
      len[s] = c; head[s] = 0; total += c;
    }
    const uint32_t take = (uint32_t)std::min<size_t>(k, total);
    // k-way selection over the sorted heads (world <= a few dozen: a linear scan of the heads beats a heap)
    for (uint32_t j = 0; j < take; j++) {
      int best = -1;
      for (int s = 0; s < world; s++) {
        if (head[s] >= len[s]) continue;
        const Rec* r = recs + (size_t)s * per + q * k;
        const Rec& cand = nearest ? r[head[s]] : r[len[s] - 1 - head[s]];
        if (best < 0) { best = s; continue; }
        const Rec* rb = recs + (size_t)best * per + q * k;
        const Rec& cur = nearest ? rb[head[best]] : rb[len[best] - 1 - head[best]];
        if (nearest ? rec_less(cand, cur) : rec_less(cur, cand)) best = s;
      }
      const Rec* rb = recs + (size_t)best * per + q * k;
      tmp[j] = nearest ? rb[head[best]] : rb[len[best] - 1 - head[best]];
      head[best]++;
    }
    for (uint32_t j = 0; j < take; j++) {
      const Rec& r = nearest ? tmp[j] : tmp[take - 1 - j];  // farthest-k were picked largest first; return ascending
      out_ids[q * k + j] = r.id; out_scores[q * k + j] = r.score;
    }
    out_counts[q] = take;
  }
}
// host threads of one merge: COLTT_MERGE_THREADS, else one per 1024 queries, at most 8 (the merge of 10 000 x 8 x 10 records is ~1.5 ms on
// one thread — it sits behind every exchange, so it is split rather than left on the exchange thread alone)
int merge_threads(size_t nq) {
  static const int knob = [] { const char* e = getenv("COLTT_MERGE_THREADS"); return (e && *e) ? std::max(1, atoi(e)) : 0; }();
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  const int want = knob ? knob : (int)std::min<size_t>(8, (nq + 1023) / 1024);
  return std::max(1, std::min(want, hw));
}
}  // namespace

// exported for the unit tests, which run without a device
extern "C" int coltt_group_merge_host(const void* recs_v, int world, size_t nq, uint32_t k, int nearest, uint64_t* out_ids,
                                      float* out_scores, uint32_t* out_counts) {
  if (!recs_v || world <= 0 || k == 0) return fail(COLTT_E_INVALID, "group_merge_host: bad arguments");
  const Rec* recs = static_cast<const Rec*>(recs_v);
  const int T = merge_threads(nq);
  if (T <= 1) { merge_range(recs, world, nq, k, nearest, 0, nq, out_ids, out_scores, out_counts); return COLTT_OK; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++)
    th.emplace_back(merge_range, recs, world, nq, k, nearest, nq * (size_t)t / (size_t)T, nq * (size_t)(t + 1) / (size_t)T, out_ids, out_scores, out_counts);
  merge_range(recs, world, nq, k, nearest, 0, nq / (size_t)T, out_ids, out_scores, out_counts);
  for (auto& t : th) t.join();
  return COLTT_OK;
}

extern "C" {

int coltt_group_unique_id(uint8_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "group_unique_id: NULL out");
  Rccl* r = rccl();
  if (!r || coltt_device_count() <= 0) {
    // no RCCL (or no device) in this process: an id for the shared-memory exchange only — random bytes name the segment
    FILE* f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(out, 1, COLTT_UNIQUE_ID_BYTES, f) : 0;
    if (f) fclose(f);
    if (got != COLTT_UNIQUE_ID_BYTES) return fail(COLTT_E_DEVICE, "group_unique_id: neither librccl nor /dev/urandom is available");
    return COLTT_OK;
  }
  COLTT_DEVICE(-1);
  NcclId id;
  COLTT_NCCL(r, r->GetUniqueId(&id));
  std::memcpy(out, id.internal, COLTT_UNIQUE_ID_BYTES);
  return COLTT_OK;
}

int coltt_shm_open(const uint8_t* unique_id, int world, int n_local, int rank_base, uint64_t bytes_per_rank, coltt_handle_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "shm_open: out is NULL");
  std::shared_ptr<ShmExchange> x;
  COLTT_TRY(shm_open_exchange(unique_id, world, n_local, rank_base, bytes_per_rank, x));
  *out = Registry::get().add(x);
  return COLTT_OK;
}
int coltt_shm_allgather(coltt_handle_t h, const void* local, uint64_t bytes, void* out) {
  auto x = lookup<ShmExchange>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "shm_allgather: unknown handle");
  if (bytes && (!local || !out)) return fail(COLTT_E_INVALID, "shm_allgather: NULL buffer");
  WriteLock g(x->rw);   // one collective at a time per process
  return shm_allgather(x.get(), local, bytes, out);
}
int coltt_shm_close(coltt_handle_t h) {
  if (!Registry::get().erase(h)) return fail(COLTT_E_NOT_FOUND, "shm_close: unknown handle");
  return COLTT_OK;
}

int coltt_group_create(const int* devices, int n_devices, uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg,
                       const coltt_group_opts* opts, coltt_handle_t* out) {
  if (!out || !devices || n_devices <= 0 || n_devices > 64) return fail(COLTT_E_INVALID, "group_create: need 1..64 devices and an out handle");
  coltt_group_opts o{}; if (opts) o = *opts;
  if (o.kind != COLTT_GROUP_FLAT && o.kind != COLTT_GROUP_HNSW) return fail(COLTT_E_INVALID, "group_create: bad kind %d", o.kind);
  if (o.layout != COLTT_LAYOUT_SHARD && o.layout != COLTT_LAYOUT_REPLICA) return fail(COLTT_E_INVALID, "group_create: bad layout %d", o.layout);
  if (o.exchange < COLTT_EXCHANGE_AUTO || o.exchange > COLTT_EXCHANGE_SHM) return fail(COLTT_E_INVALID, "group_create: bad exchange %d", o.exchange);
  const int world = o.world_size > 0 ? o.world_size : n_devices;
  if (o.rank_base < 0 || o.rank_base + n_devices > world) return fail(COLTT_E_INVALID, "group_create: ranks [%d,%d) outside world %d", o.rank_base, o.rank_base + n_devices, world);
  const bool multi_process = world > n_devices;
  if (multi_process && o.layout != COLTT_LAYOUT_SHARD) return fail(COLTT_E_INVALID, "group_create: a replica group is per process (replicas exchange nothing)");
  if ((multi_process || o.exchange == COLTT_EXCHANGE_SHM) && !o.unique_id) return fail(COLTT_E_INVALID, "group_create: world_size > n_devices (and every shared-memory group) needs the shared unique_id (coltt_group_unique_id on one process)");
  if (o.exchange == COLTT_EXCHANGE_SHM && o.layout != COLTT_LAYOUT_SHARD) return fail(COLTT_E_INVALID, "group_create: replicas exchange nothing");
  int n_dev = coltt_device_count();
  bool distinct = true;
  for (int i = 0; i < n_devices; i++) {
    if (devices[i] < 0 || devices[i] >= n_dev) return fail(COLTT_E_INVALID, "group_create: device %d out of range [0,%d)", devices[i], n_dev);
    for (int j = 0; j < i; j++) if (devices[j] == devices[i]) distinct = false;
  }
  auto g = std::make_shared<Group>();
  g->kind = o.kind; g->layout = o.layout; g->world = world; g->rank_base = o.rank_base; g->dim = dim; g->metric = metric; g->quant = quant;
  for (int i = 0; i < n_devices; i++) g->m.emplace_back(new Member());
  for (int i = 0; i < n_devices; i++) {
    Member& x = *g->m[(size_t)i];
    x.device = devices[i]; x.rank = o.rank_base + i;
    COLTT_DEVICE(x.device);
    if (o.kind == COLTT_GROUP_FLAT) COLTT_TRY(flat_create_on(x.device, dim, metric, quant, &x.h));
    else COLTT_TRY(hnsw_create_on(x.device, dim, metric, quant, cfg, &x.h));
    COLTT_HIP(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
    COLTT_HIP(hipStreamCreateWithFlags(&x.cstream, hipStreamNonBlocking));
  }
  for (auto& sl : g->slots) for (int i = 0; i < n_devices; i++) sl.mb.emplace_back(new SlotMember());
  // exchange transport
  g->exchange = COLTT_EXCHANGE_HOST;
  if (o.exchange == COLTT_EXCHANGE_SHM) {
    // processes of one box (also: several ranks on ONE device, where a RCCL communicator cannot be formed): the same packed records
    // travel through a POSIX shared-memory segment; a batch larger than the segment is exchanged in chunks of queries
    uint64_t mb = 32;
    if (const char* e = getenv("COLTT_SHM_MB")) { if (*e) mb = (uint64_t)std::max(1L, atol(e)); }
    const uint64_t per_rank = std::max<uint64_t>(4096, (mb << 20) / (uint64_t)world);
    COLTT_TRY(shm_open_exchange(o.unique_id, world, n_devices, o.rank_base, per_rank, g->shm));
    g->exchange = COLTT_EXCHANGE_SHM;
  } else if (o.layout == COLTT_LAYOUT_SHARD && o.exchange != COLTT_EXCHANGE_HOST) {
    Rccl* r = rccl();
    const bool can = r && (distinct || n_devices == 1);
    if (!can && (o.exchange == COLTT_EXCHANGE_RCCL || multi_process))
      return fail(COLTT_E_UNSUPPORTED, "group_create: RCCL exchange unavailable (%s)", !r ? "librccl not loadable" : "a communicator cannot hold one device twice");
    if (can) {
      if (multi_process || o.unique_id) {   // rank-wise bootstrap (the only way across processes; also taken when an id is handed in)
        NcclId id; std::memcpy(id.internal, o.unique_id, COLTT_UNIQUE_ID_BYTES);
        COLTT_NCCL(r, r->GroupStart());
        int bad = 0; std::string why;
        for (auto& xp : g->m) {
          Member& x = *xp;
          if (use_device(x.device) != COLTT_OK) { bad = -1; why = g_last_error; break; }
          const int e = r->CommInitRank(&x.comm, world, id, x.rank);
          if (e != 0) { bad = e; why = std::string("ncclCommInitRank: ") + r->GetErrorString(e); break; }
        }
        const int ge = r->GroupEnd();   // always closed, also after a failure inside the group
        if (bad) return fail(COLTT_E_DEVICE, "group_create: %s", why.c_str());
        if (ge != 0) return fail(COLTT_E_DEVICE, "group_create: ncclGroupEnd: %s", r->GetErrorString(ge));
      } else {
        std::vector<nccl_comm_t> comms((size_t)n_devices);
        COLTT_NCCL(r, r->CommInitAll(comms.data(), n_devices, devices));
        for (int i = 0; i < n_devices; i++) g->m[(size_t)i]->comm = comms[(size_t)i];
      }
      g->exchange = COLTT_EXCHANGE_RCCL;
    }
  } else if (multi_process) return fail(COLTT_E_UNSUPPORTED, "group_create: shards in several processes exchange through RCCL or shared memory (COLTT_EXCHANGE_SHM), not through one process's host buffer");
  g->device = devices[0];
  if (g->layout == COLTT_LAYOUT_SHARD) g->worker = std::thread([gp = g.get()] { gp->exchange_loop(); });   // joined by ~Group
  *out = Registry::get().add(g);
  return COLTT_OK;
}

int coltt_group_destroy(coltt_handle_t h) {
  if (!lookup<Group>(h)) return fail(COLTT_E_NOT_FOUND, "group_destroy: unknown handle");
  Registry::get().erase(h);
  return COLTT_OK;
}

int coltt_group_info(coltt_handle_t h, int32_t* n_local, int32_t* world, int32_t* exchange, int32_t* rank_base) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_info: unknown handle");
  if (n_local) *n_local = (int32_t)g->m.size();
  if (world) *world = g->world;
  if (exchange) *exchange = g->exchange;
  if (rank_base) *rank_base = g->rank_base;
  return COLTT_OK;
}

int coltt_group_member(coltt_handle_t h, int i, coltt_handle_t* out) {
  auto g = lookup<Group>(h);
  if (!g || !out) return fail(COLTT_E_NOT_FOUND, "group_member: unknown handle");
  if (i < 0 || (size_t)i >= g->m.size()) return fail(COLTT_E_INVALID, "group_member: index %d outside [0,%zu)", i, g->m.size());
  *out = g->m[(size_t)i]->h;
  return COLTT_OK;
}

int coltt_group_shard_of(coltt_handle_t h, uint64_t id, int32_t* out_shard) {
  auto g = lookup<Group>(h);
  if (!g || !out_shard) return fail(COLTT_E_NOT_FOUND, "group_shard_of: unknown handle");
  *out_shard = g->layout == COLTT_LAYOUT_SHARD ? (int32_t)shard_vertex(id, (uint64_t)g->world) : -1;
  return COLTT_OK;
}

int coltt_group_len(coltt_handle_t h, uint64_t* out) {
  auto g = lookup<Group>(h);
  if (!g || !out) return fail(COLTT_E_NOT_FOUND, "group_len: unknown handle");
  uint64_t tot = 0;
  for (size_t i = 0; i < g->m.size(); i++) {
    uint64_t n = 0;
    COLTT_TRY(g->kind == COLTT_GROUP_FLAT ? coltt_flat_len(g->m[i]->h, &n) : coltt_hnsw_len(g->m[i]->h, &n));
    if (g->layout == COLTT_LAYOUT_REPLICA) { tot = n; break; }
    tot += n;
  }
  *out = tot;
  return COLTT_OK;
}

// ChangedVertex (FLAT) / Insert (HNSW) routed by ShardVertex(id, world).  A process is offered every vertex and keeps the
// ones whose shard it hosts (out_kept); a replica group gives every vertex to every member.  levels: HNSW only.
static int group_ingest(Group* g, const uint64_t* ids, const float* vecs, const int32_t* levels, size_t n, uint32_t batch, uint64_t* out_kept) {
  std::lock_guard<std::mutex> lk(g->call_mu);
  const size_t nm = g->m.size();
  std::vector<std::vector<size_t>> pick(nm);
  for (size_t i = 0; i < n; i++) {
    if (g->layout == COLTT_LAYOUT_REPLICA) { for (size_t j = 0; j < nm; j++) pick[j].push_back(i); continue; }
    const int s = (int)shard_vertex(ids[i], (uint64_t)g->world) - g->rank_base;
    if (s >= 0 && (size_t)s < nm) pick[(size_t)s].push_back(i);
  }
  uint64_t kept = 0;
  for (auto& p : pick) kept += p.size();
  if (out_kept) *out_kept = g->layout == COLTT_LAYOUT_REPLICA ? n : kept;
  return for_members(g, [&](size_t j) -> int {
    const auto& p = pick[j];
    if (p.empty()) return COLTT_OK;
    std::vector<uint64_t> sid(p.size()); std::vector<float> sv(p.size() * g->dim); std::vector<int32_t> sl;
    for (size_t t = 0; t < p.size(); t++) { sid[t] = ids[p[t]]; std::memcpy(&sv[t * g->dim], vecs + p[t] * g->dim, (size_t)g->dim * 4); }
    if (g->kind == COLTT_GROUP_FLAT) return coltt_flat_upsert(g->m[j]->h, sid.data(), sv.data(), p.size());
    sl.resize(p.size());
    for (size_t t = 0; t < p.size(); t++) sl[t] = levels[p[t]];
    Member& x = *g->m[j];
    COLTT_TRY(x.d_q.reserve(sv.size() * 4));
    COLTT_HIP(hipMemcpyAsync(x.d_q.p, sv.data(), sv.size() * 4, hipMemcpyHostToDevice, x.stream));
    COLTT_HIP(hipStreamSynchronize(x.stream));
    return coltt_hnsw_insert_batch_device(x.h, sid.data(), 0, x.d_q.as<float>(), sl.data(), p.size(), batch ? batch : 1);
  });
}

int coltt_group_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n, uint64_t* out_kept) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_upsert: unknown handle");
  if (g->kind != COLTT_GROUP_FLAT) return fail(COLTT_E_INVALID, "group_upsert: not a FLAT group (use coltt_group_insert)");
  if (n && (!ids || !vecs)) return fail(COLTT_E_INVALID, "group_upsert: NULL input");
  ReadLock rl(g->rw);
  return group_ingest(g.get(), ids, vecs, nullptr, n, 0, out_kept);
}

int coltt_group_insert(coltt_handle_t h, const uint64_t* ids, const float* vecs, const int32_t* levels, size_t n, uint32_t batch,
                       uint64_t* out_kept) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_insert: unknown handle");
  if (g->kind != COLTT_GROUP_HNSW) return fail(COLTT_E_INVALID, "group_insert: not an HNSW group (use coltt_group_upsert)");
  if (n && (!ids || !vecs || !levels)) return fail(COLTT_E_INVALID, "group_insert: NULL input");
  ReadLock rl(g->rw);
  return group_ingest(g.get(), ids, vecs, levels, n, batch, out_kept);
}

int coltt_group_remove(coltt_handle_t h, const uint64_t* ids, size_t n) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_remove: unknown handle");
  if (n && !ids) return fail(COLTT_E_INVALID, "group_remove: NULL ids");
  ReadLock rl(g->rw);
  std::lock_guard<std::mutex> lk(g->call_mu);
  const size_t nm = g->m.size();
  std::vector<std::vector<uint64_t>> pick(nm);
  for (size_t i = 0; i < n; i++) {
    if (g->layout == COLTT_LAYOUT_REPLICA) { for (size_t j = 0; j < nm; j++) pick[j].push_back(ids[i]); continue; }
    const int s = (int)shard_vertex(ids[i], (uint64_t)g->world) - g->rank_base;
    if (s >= 0 && (size_t)s < nm) pick[(size_t)s].push_back(ids[i]);
  }
  return for_members(g.get(), [&](size_t j) -> int {
    if (pick[j].empty()) return COLTT_OK;
    if (g->kind == COLTT_GROUP_FLAT) return coltt_flat_remove(g->m[j]->h, pick[j].data(), pick[j].size());
    for (uint64_t id : pick[j]) COLTT_TRY(coltt_hnsw_remove(g->m[j]->h, id));  // ItemNotFoundError surfaces as in the reference
    return COLTT_OK;
  });
}

// ---- VertexSearch / Hnsw.Search over the whole collection ----------------------------------------------------------------------
}  // extern "C" (the pipeline below is C++; the entry points follow it)

namespace {
double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// a failed exchange of a shared-memory group is final: the peers, which are or will be waiting for this process, stop at once
void poison(Group* g) { if (g->exchange == COLTT_EXCHANGE_SHM && g->shm && g->shm->hdr) g->shm->hdr->failed.store(1, std::memory_order_release); }

// seconds an exchange may take before it is given up (COLTT_EXCHANGE_TIMEOUT_S, default 120): a peer that died, or never made the call, must not
// hold this process in hipStreamSynchronize for ever
double exchange_timeout_s() {
  static const double v = [] { const char* e = getenv("COLTT_EXCHANGE_TIMEOUT_S"); return (e && *e) ? std::max(0.05, atof(e)) : 120.0; }();
  return v;
}
// hipStreamSynchronize with a deadline: polls hipStreamQuery
int stream_wait(hipStream_t s, double timeout_s, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spin = 0;; spin++) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return COLTT_OK;
    if (e != hipErrorNotReady) return fail(COLTT_E_DEVICE, "%s: %s", what, hipGetErrorString(e));
    if (spin < 4000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 127u) == 127u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      return fail(COLTT_E_DEVICE, "%s: not finished after %.1f s (COLTT_EXCHANGE_TIMEOUT_S) — a peer died or never made this call", what, timeout_s);
  }
}
// the first rank whose block carries a status (record 0 of every rank's block of `per` records), or -1
int first_failed_rank(const Rec* recs, int world, size_t per, uint32_t* status) {
  for (int r = 0; r < world; r++) {
    const uint32_t st = recs[(size_t)r * per].valid >> REC_STATUS_SHIFT;
    if (st) { *status = st; return r; }
  }
  return -1;
}
int fail_for_rank(Job& j, int rank, uint32_t status) {
  j.clean_failure = true;
  if (j.local_rc != COLTT_OK) return fail(j.local_rc, "group_search: this process's shard search failed (every rank of the group returns an error for this batch): %s", j.local_err.c_str());
  return fail(COLTT_E_DEVICE, "group_search: rank %d failed its shard search (status %u); every rank of the group returns an error for this batch", rank, status);
}
}  // namespace

// exported for the unit tests, which run without a device: the rank whose block of `per` 16-byte records carries a status, or -1
extern "C" int coltt_group_first_failed_rank_host(const void* recs, int world, size_t per, uint32_t* out_status) {
  uint32_t st = 0;
  if (!recs || world <= 0 || per == 0) return -1;
  const int r = first_failed_rank(static_cast<const Rec*>(recs), world, per, &st);
  if (out_status) *out_status = st;
  return r;
}

// stages B + C of one batch, on the exchange thread
int Group::exchange_and_merge(Job& j) {
  Slot& sl = slots[j.slot];
  const size_t nm = m.size(), per = j.nq * (size_t)j.k;
  const auto t0 = std::chrono::steady_clock::now();
  // (the records were packed at the end of stage A, on the member's own stream: with the host and shared-memory transports this stage is pure DMA and
  //  proceeds while the NEXT batch's search kernel holds every CU — a pack kernel on the comm stream would queue behind that persistent grid: call N
  //  measured a 19 ms "exchange" that was 18.5 ms of waiting for a CU.  The RCCL all-gather is a kernel and still waits its turn; it is overlapped either way.)
  if (exchange == COLTT_EXCHANGE_SHM) {
    // the local members' records come to the host, then travel rank-major through the shared segment, a chunk of queries at a
    // time when the batch is larger than a slot; every process merges every chunk itself (an all-gather, like the RCCL path)
    sl.h_local.resize(nm * per);
    for (size_t i = 0; i < nm; i++) {
      Member& x = *m[i];
      COLTT_TRY(use_device(x.device));
      COLTT_HIP(hipMemcpyAsync(sl.h_local.data() + i * per, sl.mb[i]->d_pack.p, per * sizeof(Rec), hipMemcpyDeviceToHost, x.cstream));
    }
    for (auto& xp : m) { COLTT_TRY(use_device(xp->device)); COLTT_HIP(hipStreamSynchronize(xp->cstream)); }
    const size_t q_chunk = std::max<size_t>(1, (size_t)(shm->cap / ((size_t)j.k * sizeof(Rec))));
    for (size_t q0 = 0; q0 < j.nq; q0 += q_chunk) {
      const size_t qn = std::min(q_chunk, j.nq - q0), cper = qn * (size_t)j.k;
      sl.h_chunk_in.resize(nm * cper); sl.h_chunk_out.resize((size_t)world * cper);
      for (size_t i = 0; i < nm; i++) std::memcpy(sl.h_chunk_in.data() + i * cper, sl.h_local.data() + i * per + q0 * j.k, cper * sizeof(Rec));
      const auto te = std::chrono::steady_clock::now();
      COLTT_TRY(shm_allgather(shm.get(), sl.h_chunk_in.data(), cper * sizeof(Rec), sl.h_chunk_out.data()));
      j.exchange_ms += ms_since(te);
      { uint32_t st = 0; const int bad = first_failed_rank(sl.h_chunk_out.data(), world, cper, &st); if (bad >= 0) return fail_for_rank(j, bad, st); }   // (every chunk carries it: all ranks stop at the first)
      const auto tm = std::chrono::steady_clock::now();
      COLTT_TRY(coltt_group_merge_host(sl.h_chunk_out.data(), world, qn, j.k, j.nearest, j.out_ids + q0 * j.k, j.out_scores + q0 * j.k, j.out_counts + q0));
      j.merge_ms += ms_since(tm);
    }
    j.exchange_ms = ms_since(t0) - j.merge_ms;
    return COLTT_OK;
  }
  COLTT_TRY(sl.h_stage.reserve((size_t)world * per * sizeof(Rec)));
  if (exchange == COLTT_EXCHANGE_RCCL) {
    Rccl* r = rccl();
    COLTT_NCCL(r, r->GroupStart());
    int bad = 0; std::string why;
    for (size_t i = 0; i < nm; i++) { Member& x = *m[i]; SlotMember& b = *sl.mb[i];
      if (use_device(x.device) != COLTT_OK) { bad = -1; why = g_last_error; break; }
      const int e = r->AllGather(b.d_pack.p, b.d_gather.p, per * sizeof(Rec), 0 /*ncclInt8*/, x.comm, x.cstream);
      if (e != 0) { bad = e; why = std::string("ncclAllGather: ") + r->GetErrorString(e); break; }
    }
    const int ge = r->GroupEnd();   // always closed, also after a failure inside the group
    if (bad) return fail(COLTT_E_DEVICE, "group_search: %s", why.c_str());
    if (ge != 0) return fail(COLTT_E_DEVICE, "group_search: ncclGroupEnd: %s", r->GetErrorString(ge));
    Member& x0 = *m[0];
    COLTT_TRY(use_device(x0.device));
    COLTT_HIP(hipMemcpyAsync(sl.h_stage.p, sl.mb[0]->d_gather.p, (size_t)world * per * sizeof(Rec), hipMemcpyDeviceToHost, x0.cstream));
    // bounded: a peer that never issues its all-gather (it died, or it never made this call) must not hold this process for ever
    for (auto& xp : m) {
      COLTT_TRY(use_device(xp->device));
      if (stream_wait(xp->cstream, exchange_timeout_s(), "group_search: RCCL all-gather") != COLTT_OK) {
        broken.store(true);
        const std::string why = g_last_error;
        if (r->CommAbort) for (auto& yp : m) if (yp->comm) { (void)use_device(yp->device); (void)r->CommAbort(yp->comm); yp->comm = nullptr; }
        return fail(COLTT_E_DEVICE, "%s", why.c_str());
      }
    }
  } else {
    for (size_t i = 0; i < nm; i++) {
      Member& x = *m[i];
      COLTT_TRY(use_device(x.device));
      COLTT_HIP(hipMemcpyAsync(sl.h_stage.as<Rec>() + (size_t)x.rank * per, sl.mb[i]->d_pack.p, per * sizeof(Rec), hipMemcpyDeviceToHost, x.cstream));
    }
    for (auto& xp : m) { COLTT_TRY(use_device(xp->device)); COLTT_HIP(hipStreamSynchronize(xp->cstream)); }
  }
  j.exchange_ms = ms_since(t0);
  { uint32_t st = 0; const int bad = first_failed_rank(sl.h_stage.as<Rec>(), world, per, &st); if (bad >= 0) return fail_for_rank(j, bad, st); }
  const auto tm = std::chrono::steady_clock::now();
  COLTT_TRY(coltt_group_merge_host(sl.h_stage.p, world, j.nq, j.k, j.nearest, j.out_ids, j.out_scores, j.out_counts));
  j.merge_ms = ms_since(tm);
  return COLTT_OK;
}

void Group::exchange_loop() {
  for (;;) {
    std::shared_ptr<Job> j;
    {
      std::unique_lock<std::mutex> lk(q_mu);
      q_cv.wait(lk, [&] { return stop || !queue.empty(); });
      if (queue.empty()) return;   // stop, nothing left in flight
      j = queue.front(); queue.pop_front();
    }
    g_last_error.clear();
    j->rc = exchange_and_merge(*j);
    if (j->rc != COLTT_OK) { j->err = g_last_error; if (!j->clean_failure) poison(this); }   // (a rank's status record is a CLEAN failure: the transport is intact, the next batch may run)
    {
      std::lock_guard<std::mutex> lk(q_mu);
      j->done = true; slot_busy[j->slot] = false;
      t_batches++; t_search += j->search_ms; t_exchange += j->exchange_ms; t_merge += j->merge_ms;
    }
    slot_cv.notify_all(); done_cv.notify_all();
  }
}

namespace {

// stage A of a shard search + hand-over to the exchange thread.  d_queries_per_member != NULL: the batch already lives on every
// member's device ([n_local] pointers, nq x dim f32 each); otherwise `queries` is a host array broadcast to the members.
int shard_begin(Group* g, const float* queries, const float* const* d_queries_per_member, size_t nq, uint32_t k, int select, int mode,
                uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts, uint64_t* out_ticket) {
  const bool hn = g->kind == COLTT_GROUP_HNSW;
  const size_t per = nq * (size_t)k;
  auto job = std::make_shared<Job>();
  job->nq = nq; job->k = k; job->nearest = hn ? 1 : (select == COLTT_SELECT_NEAREST);
  job->out_ids = out_ids; job->out_scores = out_scores; job->out_counts = out_counts;
  if (g->broken.load()) return fail(COLTT_E_DEVICE, "group_search: an earlier exchange of this group timed out — its communicator is gone (destroy the group)");
  // test hook: COLTT_TEST_FAIL_STAGE_A=<rank> makes that rank's shard search fail before anything is searched (tests/test_gpu_group.py)
  int inject_rank = -1;
  if (const char* e = getenv("COLTT_TEST_FAIL_STAGE_A")) { if (*e) inject_rank = atoi(e); }
  std::lock_guard<std::mutex> lk(g->call_mu);   // one stage A at a time; tickets are taken in this order
  {
    std::unique_lock<std::mutex> ql(g->q_mu);   // a slot: at most GROUP_SLOTS batches between their search and the end of their merge
    g->slot_cv.wait(ql, [&] { for (int i = 0; i < GROUP_SLOTS; i++) if (!g->slot_busy[i]) return true; return false; });
    for (int i = 0; i < GROUP_SLOTS; i++) if (!g->slot_busy[i]) { job->slot = i; g->slot_busy[i] = true; break; }
  }
  Slot& sl = g->slots[job->slot];
  const auto t0 = std::chrono::steady_clock::now();
  int rc = for_members(g, [&](size_t j) -> int {
    Member& x = *g->m[j]; SlotMember& b = *sl.mb[j];
    const float* dq;
    if (d_queries_per_member) dq = d_queries_per_member[j];
    else {
      COLTT_TRY(b.d_q.reserve(nq * g->dim * 4));
      COLTT_HIP(hipMemcpyAsync(b.d_q.p, queries, nq * g->dim * 4, hipMemcpyHostToDevice, x.stream));
      COLTT_HIP(hipStreamSynchronize(x.stream));
      dq = b.d_q.as<float>();
    }
    COLTT_TRY(b.d_ids.reserve(per * 8)); COLTT_TRY(b.d_sc.reserve(per * 4)); COLTT_TRY(b.d_cnt.reserve(nq * 4));
    COLTT_TRY(b.d_pack.reserve(per * sizeof(Rec)));
    if (g->exchange == COLTT_EXCHANGE_RCCL) COLTT_TRY(b.d_gather.reserve((size_t)g->world * per * sizeof(Rec)));
    if (x.rank == inject_rank) return fail(COLTT_E_DEVICE, "injected stage-A failure on rank %d (COLTT_TEST_FAIL_STAGE_A)", x.rank);
    if (hn) COLTT_TRY(coltt_hnsw_search_device(x.h, dq, nq, k, ef_override, b.d_ids.as<uint64_t>(), b.d_sc.as<float>(), b.d_cnt.as<uint32_t>(), nullptr));
    else COLTT_TRY(coltt_flat_search_device(x.h, dq, nq, k, select, mode, b.d_ids.as<uint64_t>(), b.d_sc.as<float>(), b.d_cnt.as<uint32_t>()));
    // the packed {id, score, valid} records of this member: a ~10 us kernel right behind its search, while the device is still this batch's
    pack_topk_kernel<<<ceil_div(per, 256), 256, 0, x.stream>>>(b.d_ids.as<uint64_t>(), b.d_sc.as<float>(), b.d_cnt.as<uint32_t>(), (uint32_t)nq, k, b.d_pack.as<Rec>());
    COLTT_HIP(hipGetLastError());
    COLTT_HIP(hipStreamSynchronize(x.stream));
    return COLTT_OK;
  });
  job->search_ms = ms_since(t0);
  if (rc != COLTT_OK) {
    // This rank's shard search failed.  Its peers are (or will be) in the all-gather of this batch: the exchange still happens, with a block of STATUS
    // records from every local member, and every rank returns an error for the batch.  Only if even that cannot be done (no memory for the records, a
    // dead device) is the batch dropped here — the peers' bounded wait (RCCL) / the poisoned segment (shared memory) releases them.
    job->local_rc = rc; job->local_err = g_last_error;
    const uint32_t status = (uint32_t)(rc < 0 ? -rc : rc) & 0xffffffu;
    const int rc2 = for_members(g, [&](size_t j) -> int {
      Member& x = *g->m[j]; SlotMember& b = *sl.mb[j];
      COLTT_TRY(b.d_pack.reserve(per * sizeof(Rec)));
      if (g->exchange == COLTT_EXCHANGE_RCCL) COLTT_TRY(b.d_gather.reserve((size_t)g->world * per * sizeof(Rec)));
      pack_status_kernel<<<ceil_div(per, 256), 256, 0, x.stream>>>(per, status ? status : 1u, b.d_pack.as<Rec>());
      COLTT_HIP(hipGetLastError());
      COLTT_HIP(hipStreamSynchronize(x.stream));
      return COLTT_OK;
    });
    if (rc2 != COLTT_OK) {
      std::lock_guard<std::mutex> ql(g->q_mu);
      g->slot_busy[job->slot] = false; g->slot_cv.notify_all();
      poison(g);
      g_last_error = job->local_err;
      return rc;
    }
  }
  std::lock_guard<std::mutex> ql(g->q_mu);
  job->ticket = g->next_ticket++;
  g->jobs[job->ticket] = job;
  g->queue.push_back(job);
  g->q_cv.notify_one();
  *out_ticket = job->ticket;
  return COLTT_OK;
}

int shard_end(Group* g, uint64_t ticket) {
  std::shared_ptr<Job> j;
  {
    std::unique_lock<std::mutex> lk(g->q_mu);
    auto it = g->jobs.find(ticket);
    if (it == g->jobs.end()) return fail(COLTT_E_NOT_FOUND, "group_search_end: unknown ticket %llu", (unsigned long long)ticket);
    j = it->second;
    g->done_cv.wait(lk, [&] { return j->done; });
    g->jobs.erase(it);
  }
  if (j->rc != COLTT_OK) { g_last_error = j->err; return j->rc; }
  return COLTT_OK;
}

// rank-symmetric argument errors are returned BEFORE anything is begun: every process sees the same bad argument, nobody is left
// waiting, and the shared segment is not poisoned by them
int check_search_args(Group* g, uint32_t k) {
  if (k == 0) return fail(COLTT_E_INVALID, "group_search: k must be >= 1");
  if (g->exchange == COLTT_EXCHANGE_SHM && g->shm && (uint64_t)k * sizeof(Rec) > g->shm->cap)
    return fail(COLTT_E_INVALID, "group_search: k = %u needs %llu bytes per rank and query, the shared segment holds %llu per rank (COLTT_SHM_MB)", k,
                (unsigned long long)((uint64_t)k * sizeof(Rec)), (unsigned long long)g->shm->cap);
  return COLTT_OK;
}

int replica_search(Group* g, const float* queries, const float* const* d_queries_per_member, size_t nq, uint32_t k, int select, int mode,
                   uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  std::lock_guard<std::mutex> lk(g->call_mu);
  const size_t nm = g->m.size();
  const bool hn = g->kind == COLTT_GROUP_HNSW;
  // the batch is split into contiguous slices, one per member; answers land in place — no exchange
  return for_members(g, [&](size_t j) -> int {
    const size_t lo = nq * j / nm, hi = nq * (j + 1) / nm;
    if (hi == lo) return COLTT_OK;
    Member& x = *g->m[j];
    if (d_queries_per_member) {
      COLTT_TRY(x.d_ids.reserve((hi - lo) * k * 8)); COLTT_TRY(x.d_sc.reserve((hi - lo) * k * 4)); COLTT_TRY(x.d_cnt.reserve((hi - lo) * 4));
      const float* dq = d_queries_per_member[j] + lo * g->dim;
      if (hn) COLTT_TRY(coltt_hnsw_search_device(x.h, dq, hi - lo, k, ef_override, x.d_ids.as<uint64_t>(), x.d_sc.as<float>(), x.d_cnt.as<uint32_t>(), nullptr));
      else COLTT_TRY(coltt_flat_search_device(x.h, dq, hi - lo, k, select, mode, x.d_ids.as<uint64_t>(), x.d_sc.as<float>(), x.d_cnt.as<uint32_t>()));
      COLTT_HIP(hipMemcpyAsync(out_ids + lo * k, x.d_ids.p, (hi - lo) * k * 8, hipMemcpyDeviceToHost, x.stream));
      COLTT_HIP(hipMemcpyAsync(out_scores + lo * k, x.d_sc.p, (hi - lo) * k * 4, hipMemcpyDeviceToHost, x.stream));
      COLTT_HIP(hipMemcpyAsync(out_counts + lo, x.d_cnt.p, (hi - lo) * 4, hipMemcpyDeviceToHost, x.stream));
      COLTT_HIP(hipStreamSynchronize(x.stream));
      return COLTT_OK;
    }
    if (hn) return coltt_hnsw_search(x.h, queries + lo * g->dim, hi - lo, k, ef_override, out_ids + lo * k, out_scores + lo * k, out_counts + lo, nullptr);
    return coltt_flat_search(x.h, queries + lo * g->dim, hi - lo, k, select, mode, out_ids + lo * k, out_scores + lo * k, out_counts + lo);
  });
}

// One synchronous call.  COLTT_GROUP_SUBBATCH=<queries> splits a larger call into sub-batches that go through the pipeline one behind
// the other (the exchange + merge of sub-batch i under the search of i + 1); default: the call is one batch — callers that stream
// batches overlap them with coltt_group_search_begin / _end instead, which costs the search kernels no occupancy.
int group_search(Group* g, const float* queries, const float* const* d_queries_per_member, size_t nq, uint32_t k, int select,
                 int mode, uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  if (nq == 0) return COLTT_OK;
  COLTT_TRY(check_search_args(g, k));
  if (g->layout == COLTT_LAYOUT_REPLICA) return replica_search(g, queries, d_queries_per_member, nq, k, select, mode, ef_override, out_ids, out_scores, out_counts);
  static const size_t sub_knob = [] { const char* e = getenv("COLTT_GROUP_SUBBATCH"); return (e && *e) ? (size_t)std::max(1L, atol(e)) : (size_t)0; }();
  const size_t sub = sub_knob ? sub_knob : nq;
  std::vector<uint64_t> tickets;
  std::vector<const float*> dq(g->m.size());
  int rc = COLTT_OK;
  for (size_t q0 = 0; q0 < nq && rc == COLTT_OK; q0 += sub) {
    const size_t qn = std::min(sub, nq - q0);
    if (d_queries_per_member) for (size_t j = 0; j < dq.size(); j++) dq[j] = d_queries_per_member[j] + q0 * g->dim;
    uint64_t t = 0;
    rc = shard_begin(g, queries ? queries + q0 * g->dim : nullptr, d_queries_per_member ? dq.data() : nullptr, qn, k, select, mode, ef_override,
                     out_ids + q0 * k, out_scores + q0 * k, out_counts + q0, &t);
    if (rc == COLTT_OK) tickets.push_back(t);
  }
  std::string first_err = rc != COLTT_OK ? g_last_error : std::string();
  for (uint64_t t : tickets) { const int e = shard_end(g, t); if (e != COLTT_OK && rc == COLTT_OK) { rc = e; first_err = g_last_error; } }
  if (rc != COLTT_OK) g_last_error = first_err;
  return rc;
}

}  // namespace

extern "C" {

int coltt_group_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode, uint32_t ef_override,
                       uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_search: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "group_search: NULL buffer");
  ReadLock rl(g->rw);
  return group_search(g.get(), queries, nullptr, nq, k, select, mode, ef_override, out_ids, out_scores, out_counts);
}

int coltt_group_search_device(coltt_handle_t h, const float* const* d_queries_per_member, size_t nq, uint32_t k, int select, int mode,
                              uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_search_device: unknown handle");
  if (nq && (!d_queries_per_member || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "group_search_device: NULL buffer");
  ReadLock rl(g->rw);
  return group_search(g.get(), nullptr, d_queries_per_member, nq, k, select, mode, ef_override, out_ids, out_scores, out_counts);
}

// Streaming form of a shard search: _begin returns when every local member has searched the batch (stage A) and the batch's exchange
// and merge have been queued; _end blocks until the merged answers are in the out arrays handed to _begin (which must stay valid until
// then).  A caller that keeps one batch begun while it ends the previous one hides the all-gather and the host merge under the next
// search.  At most 3 batches may be begun and not ended (a fourth _begin waits for a slot).  Multi-process groups: every process makes
// the same _begin calls in the same order.
int coltt_group_search_begin(coltt_handle_t h, const float* queries, const float* const* d_queries_per_member, size_t nq, uint32_t k,
                             int select, int mode, uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts,
                             uint64_t* out_ticket) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_search_begin: unknown handle");
  if (!out_ticket) return fail(COLTT_E_INVALID, "group_search_begin: out_ticket is NULL");
  if (g->layout != COLTT_LAYOUT_SHARD) return fail(COLTT_E_INVALID, "group_search_begin: replicas exchange nothing (use coltt_group_search)");
  if (nq == 0 || (!queries && !d_queries_per_member) || !out_ids || !out_scores || !out_counts) return fail(COLTT_E_INVALID, "group_search_begin: empty batch or NULL buffer");
  ReadLock rl(g->rw);
  COLTT_TRY(check_search_args(g.get(), k));
  return shard_begin(g.get(), d_queries_per_member ? nullptr : queries, d_queries_per_member, nq, k, select, mode, ef_override, out_ids, out_scores, out_counts, out_ticket);
}

int coltt_group_search_end(coltt_handle_t h, uint64_t ticket) {
  auto g = lookup<Group>(h);
  if (!g) return fail(COLTT_E_NOT_FOUND, "group_search_end: unknown handle");
  return shard_end(g.get(), ticket);
}

// cumulative wall-clock of the finished shard-search batches of this group: out = {batches, search_ms, exchange_ms, merge_ms} —
// search = stage A (every member's search, concurrent), exchange = pack + all-gather + D2H, merge = the host merge
int coltt_group_timing(coltt_handle_t h, double* out4) {
  auto g = lookup<Group>(h);
  if (!g || !out4) return fail(COLTT_E_NOT_FOUND, "group_timing: unknown handle");
  std::lock_guard<std::mutex> lk(g->q_mu);
  out4[0] = (double)g->t_batches; out4[1] = g->t_search; out4[2] = g->t_exchange; out4[3] = g->t_merge;
  return COLTT_OK;
}

}  // extern "C"
