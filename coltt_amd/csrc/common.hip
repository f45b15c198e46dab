// common.hip — registry, error string, device selection.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace coltt {

thread_local std::string g_last_error;
static int g_device = -1;
static std::mutex g_dev_mu;

Registry& Registry::get() {
  static Registry r;
  return r;
}
coltt_handle_t Registry::add(std::shared_ptr<Object> o) {
  std::lock_guard<std::mutex> g(mu_);
  coltt_handle_t h = next_++;
  map_[h] = std::move(o);
  return h;
}
std::shared_ptr<Object> Registry::find(coltt_handle_t h) {
  std::lock_guard<std::mutex> g(mu_);
  auto it = map_.find(h);
  return it == map_.end() ? nullptr : it->second;
}
bool Registry::erase(coltt_handle_t h) {
  std::shared_ptr<Object> keep;
  {
    std::lock_guard<std::mutex> g(mu_);
    auto it = map_.find(h);
    if (it == map_.end()) return false;
    keep = it->second;
    map_.erase(it);
  }
  WriteLock g2(keep->rw);  // wait for in-flight calls (searches hold it shared)
  return true;
}

// cgo calls arrive on arbitrary OS threads; the HIP "current device" is per thread.
int ensure_device() {
  int dev;
  {
    std::lock_guard<std::mutex> g(g_dev_mu);
    if (g_device < 0) {
      int n = 0;
      hipError_t e = hipGetDeviceCount(&n);
      if (e != hipSuccess || n <= 0)
        return fail(COLTT_E_DEVICE, "libcoltt_gpu: no HIP device visible (%s) — this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
      g_device = 0;
    }
    dev = g_device;
  }
  COLTT_HIP(hipSetDevice(dev));
  return COLTT_OK;
}

int default_device() { std::lock_guard<std::mutex> g(g_dev_mu); return g_device < 0 ? 0 : g_device; }

int use_device(int device) {
  COLTT_HIP(hipSetDevice(device));
  return COLTT_OK;
}

namespace {
std::shared_mutex g_pol_mu;
Policy g_policy;
bool g_policy_loaded = false;
Policy read_policy_from_env() {
  Policy p;
  auto off = [](const char* n) { const char* e = getenv(n); return e && *e == '0'; };
  auto num = [](const char* n, bool& set) -> long long { const char* e = getenv(n); set = e && *e; return set ? atoll(e) : 0; };
  bool set = false; long long v;
  p.flat_one = !off("COLTT_FLAT_ONE"); p.staging = !off("COLTT_STAGING"); p.ev8 = !off("COLTT_EV8"); { const char* e = getenv("COLTT_ROWS8"); p.rows8 = (!e || !*e) ? 1 : (*e == '0' ? 0 : (*e == '2' ? 2 : 1)); } p.f8_mfma = !off("COLTT_F8_MFMA");
  v = num("COLTT_VISG", set); p.visg = set ? (int)v : -1;
  { const char* e = getenv("COLTT_WALK2"); p.walk2 = (!e || !*e) ? 7 : (!strcmp(e, "off") ? -1 : (atoi(e) & 15)); }
  { const char* e = getenv("COLTT_WALK2_LDS"); if (!e || !*e) p.walk2_lds = 4; else if (!strcmp(e, "off")) p.walk2_lds = -1; else { const int w = atoi(e) & 6; p.walk2_lds = w ? w : -1; } }
  v = num("COLTT_BLOOM_KB", set); p.bloom_kb = set ? (int)std::max<long long>(1, std::min<long long>(64, v)) : 0;
  v = num("COLTT_WAVES_PER_CU", set); p.waves_per_cu = set ? (int)std::max<long long>(1, std::min<long long>(12, v)) : 0;
  v = num("COLTT_ROWS_NT", set); p.rows_nt = set ? (v > 0 ? 1 : (v < 0 ? -1 : 0)) : -1;
  v = num("COLTT_ROWS_NT_MIN_MB", set); p.rows_nt_min_mb = set ? std::max<long long>(0, v) : 12288;
  v = num("COLTT_PQ_WAVES", set); p.pq_waves = set ? (int)std::max<long long>(1, std::min<long long>(16, v)) : 0;
  p.pq_nbr = !off("COLTT_PQ_NBR");
  { const char* e = getenv("COLTT_LAT_SEQ"); p.lat_seq = e && *e == '1'; }
  v = num("COLTT_LAT_HELPERS", set); p.lat_helpers = set ? (int)std::max<long long>(0, std::min<long long>(3, v)) : 0;
  v = num("COLTT_LAT_MAX_NQ", set); if (!set) v = num("COLTT_MW_MAX_NQ", set);
  p.lat_knob_set = set; p.lat_max_nq = set ? (uint32_t)std::max<long long>(0, v) : 0;
  v = num("COLTT_VISG_BUDGET_MB", set); p.visg_budget_mb = set ? v : -1;
  return p;
}
}  // namespace

Policy policy() {
  {
    std::shared_lock<std::shared_mutex> g(g_pol_mu);
    if (g_policy_loaded) return g_policy;
  }
  std::unique_lock<std::shared_mutex> g(g_pol_mu);
  if (!g_policy_loaded) { g_policy = read_policy_from_env(); g_policy_loaded = true; }
  return g_policy;
}

size_t max_search_ctx() {
  static const size_t v = [] { const char* e = getenv("COLTT_MAX_SEARCH_CTX"); long n = e && *e ? atol(e) : 32; return (size_t)(n < 1 ? 1 : (n > 256 ? 256 : n)); }();
  return v;
}

}  // namespace coltt

using namespace coltt;

extern "C" {

int coltt_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(COLTT_E_DEVICE, "coltt_init: no HIP device visible (%s) — this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(COLTT_E_INVALID, "coltt_init: device %d out of range [0,%d)", device, n);
  {
    std::lock_guard<std::mutex> g(g_dev_mu);
    g_device = device;
  }
  COLTT_HIP(hipSetDevice(device));
  return COLTT_OK;
}

int coltt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* coltt_last_error(void) { return g_last_error.c_str(); }

int coltt_policy_reload(void) {
  Policy p = read_policy_from_env();
  std::unique_lock<std::shared_mutex> g(g_pol_mu);
  g_policy = p; g_policy_loaded = true;
  return COLTT_OK;
}
const char* coltt_version(void) {
  return "coltt_gpu 0.6 (gfx950)";
}

}  // extern "C"

#include "exact.hpp"
extern "C" uint64_t coltt_shard_vertex_host(uint64_t id, uint64_t shard_count) {
  return shard_count ? coltt::dev::shard_vertex(id, shard_count) : 0;
}
