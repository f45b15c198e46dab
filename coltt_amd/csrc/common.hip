// common.hip — registry, error string, device selection.
#include "common.hpp"

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
const char* coltt_version(void) {
#ifdef COLTT_EXPERIMENTS
  return "coltt_gpu 0.3 (gfx950) +experiments";   // superseded kernel generations compiled in (tools/experiments/)
#else
  return "coltt_gpu 0.3 (gfx950)";
#endif
}

}  // extern "C"

#include "exact.hpp"
extern "C" uint64_t coltt_shard_vertex_host(uint64_t id, uint64_t shard_count) {
  return shard_count ? coltt::dev::shard_vertex(id, shard_count) : 0;
}
