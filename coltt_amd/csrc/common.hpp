// common.hpp — host-side plumbing shared by the HIP translation units of libcoltt_gpu.so:
// thread-local error string, opaque-handle registry, RAII device buffers, launch checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
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
      COLTT_HIP(hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s));
      COLTT_HIP(hipStreamSynchronize(s));
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = ncap;
    return COLTT_OK;
  }
  template <class T> T* as() const { return (T*)p; }
};

struct Object {
  std::mutex mu;  // one call at a time per object (the Go side micro-batches; SURVEY.md §8b threading)
  virtual ~Object() {}
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

int ensure_device();
hipStream_t main_stream();

inline size_t quant_bytes(int q) { return q == COLTT_Q_NONE ? 4 : (q == COLTT_Q_F8 ? 1 : 2); }
inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace coltt
