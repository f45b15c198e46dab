// cflat.hip — experimental CFLAT on the GPU: multi-vector weighted FLAT scan
// (experimental/multi_vector_vertex.go:60-137).
//
// A vertex carries n_fields vectors; a query supplies one vector, a ratio and an include flag per field; the score is
//   sum over included fields of scoreHelper(Distance(node[f], q[f])) * (float32(ratio[f]) / 100)        (:113-119)
// accumulated in f32 in field order; the K LARGEST scores are kept and returned DESCENDING
// (experimental/multi_priority_queue.go:46-77).  Distances use the same pair-owned exact-order code as everything else.
// HBM layout: one row array per field sharing slots (rows[f][cap][stride], norms[f][cap]); removal swaps the last slot in.
#include <algorithm>

#include "common.hpp"
#include "exact.hpp"
#include "prep.hpp"
#include "select.hpp"

using namespace coltt;
using namespace coltt::dev;

namespace {

constexpr int CF_MAX_FIELDS = 8;

struct CFields { const uint8_t* rows[CF_MAX_FIELDS]; const float* norms[CF_MAX_FIELDS]; };

// scoreHelper (edge/edge_helper.go:143-148 / experimental twin)
template <int METRIC> __device__ __forceinline__ float score_helper(float d) {
  if constexpr (METRIC == M_COS) return ((2.0f - d) / 2.0f) * 100.0f;
  else return (float)fmax(0.0, (double)(100.0f - d));
}

// one query per launch (the reference RPC is single-query); each wave owns 32 vertices per iteration
template <int METRIC>
__global__ __launch_bounds__(256) void cflat_scan_kernel(CFields F, size_t stride, uint64_t n, int nf, int dim, const float* __restrict__ q_eff,
                                                        const float* __restrict__ qnorms, const float* __restrict__ weight /* ratio/100 or <0 = excluded */,
                                                        const uint32_t* __restrict__ thr, unsigned long long* __restrict__ cand,
                                                        uint32_t* __restrict__ cnt, uint32_t cap, uint64_t begin, uint64_t end) {
  extern __shared__ __attribute__((aligned(16))) float qs[];  // [nf][dim]
  for (int i = threadIdx.x; i < nf * dim; i += blockDim.x) qs[i] = q_eff[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane & 1, p = lane >> 1;
  const uint32_t th = thr[0];
  const uint64_t ngroups = (end - begin + 31) / 32;
  for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += (uint64_t)gridDim.x * 4) {
    uint64_t pos = begin + g * 32 + p;
    bool valid = pos < end;
    uint64_t slot = valid ? pos : begin;
    float score = 0.f;
    for (int f = 0; f < nf; f++) {
      const float w = weight[f];
      if (w < 0.f) continue;  // IncludeOrNot == false
      float rn = METRIC == M_COS ? F.norms[f][slot] : 0.f;
      float d = pair_distance<METRIC, Q_NONE, 4>(F.rows[f] + slot * stride, qs + (size_t)f * dim, dim, qnorms[f], rn, half);
      score += score_helper<METRIC>(d) * w;
    }
    uint32_t sk = score_key(score);
    if (valid && half == 0 && sk >= th) {
      uint32_t idx = atomicAdd(&cnt[0], 1u);
      if (idx < cap) cand[idx] = ((unsigned long long)sk << 32) | (uint32_t)slot;
    }
  }
}

__global__ void cflat_weights_kernel(const uint32_t* ratio, const uint8_t* include, int nf, float* w) {
  int f = threadIdx.x;
  if (f < nf) w[f] = include[f] ? div_rn((float)ratio[f], 100.0f) : -1.0f;  // float32(Ratio) / 100
}

struct CFlat : Object {
  uint32_t dim = 0, nf = 0; int metric = 0; size_t stride = 0;
  uint64_t n = 0, cap = 0;
  DevBuf rows[CF_MAX_FIELDS], norms[CF_MAX_FIELDS], ids;
  std::unordered_map<uint64_t, uint32_t> id2slot; std::vector<uint64_t> h_ids;
  hipStream_t stream = nullptr;
  DevBuf w_raw, w_q, w_qn, w_misc, w_cand, w_out_ids, w_out_sc, w_out_cnt;
  ~CFlat() override { if (stream) (void)hipStreamDestroy(stream); }
  int reserve(uint64_t need) {
    if (need <= cap) return COLTT_OK;
    uint64_t nc = std::max<uint64_t>({need, cap + cap / 2, 1024});
    for (uint32_t f = 0; f < nf; f++) { COLTT_TRY(rows[f].reserve(nc * stride, true, stream)); COLTT_TRY(norms[f].reserve(nc * 4, true, stream)); }
    COLTT_TRY(ids.reserve(nc * 8, true, stream));
    cap = nc;
    return COLTT_OK;
  }
};

// strided gather of one field out of the [n][nf][dim] upload, then Normalize (cosine) into the field's rows
__global__ void cflat_take_field_kernel(const float* __restrict__ all, uint64_t n, int nf, int f, int dim, float* __restrict__ out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (uint64_t)dim) return;
  uint64_t i = t / dim; int e = (int)(t - i * dim);
  out[t] = all[(i * nf + f) * (uint64_t)dim + e];
}

}  // namespace

extern "C" {

int coltt_cflat_create(uint32_t dim, int metric, uint32_t n_fields, coltt_handle_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "cflat_create: out is NULL");
  if (dim == 0 || dim > 4096 || dim % 4) return fail(COLTT_E_INVALID, "cflat_create: dim %u must be a multiple of 4 in [4,4096]", dim);
  if (n_fields == 0 || n_fields > CF_MAX_FIELDS) return fail(COLTT_E_INVALID, "cflat_create: n_fields %u outside [1,%d]", n_fields, CF_MAX_FIELDS);
  if (metric != COLTT_COSINE && metric != COLTT_EUCLIDEAN) return fail(COLTT_E_INVALID, "cflat_create: bad metric %d", metric);
  if ((size_t)n_fields * dim * 4 > 150 * 1024) return fail(COLTT_E_UNSUPPORTED, "cflat_create: n_fields x dim too large for the LDS query tile");
  COLTT_DEVICE(-1);
  auto c = std::make_shared<CFlat>();
  c->dim = dim; c->nf = n_fields; c->metric = metric; c->stride = ((size_t)dim * 4 + 15) & ~(size_t)15;
  COLTT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->device = default_device();
  *out = Registry::get().add(c);
  return COLTT_OK;
}

int coltt_cflat_destroy(coltt_handle_t h) {
  if (!Registry::get().erase(h)) return fail(COLTT_E_NOT_FOUND, "cflat_destroy: unknown handle");
  return COLTT_OK;
}

int coltt_cflat_len(coltt_handle_t h, uint64_t* out) {
  auto c = lookup<CFlat>(h);
  if (!c || !out) return fail(COLTT_E_NOT_FOUND, "cflat_len: unknown handle");
  WriteLock g(c->rw);
  *out = c->n;
  return COLTT_OK;
}

/* ChangedVertex (experimental/multi_vector_vertex.go:60-75): vecs is [n][n_fields][dim]; every field is normalised for cosine */
int coltt_cflat_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n) {
  auto c = lookup<CFlat>(h);
  if (!c) return fail(COLTT_E_NOT_FOUND, "cflat_upsert: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids || !vecs) return fail(COLTT_E_INVALID, "cflat_upsert: NULL input");
  WriteLock g(c->rw);
  COLTT_DEVICE(c->device);
  for (size_t i = 0; i < n; i++) {  // one vertex at a time keeps "last write wins" trivially right; this path is not hot
    uint32_t slot;
    auto it = c->id2slot.find(ids[i]);
    if (it != c->id2slot.end()) slot = it->second;
    else { COLTT_TRY(c->reserve(c->n + 1)); slot = (uint32_t)c->n; c->id2slot[ids[i]] = slot; c->h_ids.push_back(ids[i]); c->n++;
           COLTT_HIP(hipMemcpyAsync(c->ids.as<uint64_t>() + slot, &ids[i], 8, hipMemcpyHostToDevice, c->stream)); }
    COLTT_TRY(c->w_raw.reserve((size_t)c->nf * c->dim * 4));
    COLTT_HIP(hipMemcpyAsync(c->w_raw.p, vecs + i * (size_t)c->nf * c->dim, (size_t)c->nf * c->dim * 4, hipMemcpyHostToDevice, c->stream));
    for (uint32_t f = 0; f < c->nf; f++) {
      prep_rows_kernel<Q_NONE><<<1, 64, 0, c->stream>>>(c->w_raw.as<float>() + (size_t)f * c->dim, 1, (int)c->dim, c->metric == COLTT_COSINE, nullptr, slot,
                                                          c->rows[f].as<uint8_t>(), c->stride);
      row_norms_kernel<Q_NONE><<<1, 64, 0, c->stream>>>(c->rows[f].as<uint8_t>(), c->stride, nullptr, slot, 1, (int)c->dim, c->norms[f].as<float>());
    }
    COLTT_HIP(hipStreamSynchronize(c->stream));
  }
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

/* RemoveVertex (multi_vector_vertex.go:77-83) */
int coltt_cflat_remove(coltt_handle_t h, const uint64_t* ids, size_t n) {
  auto c = lookup<CFlat>(h);
  if (!c) return fail(COLTT_E_NOT_FOUND, "cflat_remove: unknown handle");
  if (n && !ids) return fail(COLTT_E_INVALID, "cflat_remove: NULL ids");
  WriteLock g(c->rw);
  COLTT_DEVICE(c->device);
  for (size_t i = 0; i < n; i++) {
    auto it = c->id2slot.find(ids[i]);
    if (it == c->id2slot.end()) continue;
    uint32_t s = it->second; uint64_t last = c->n - 1;
    c->id2slot.erase(it);
    if (s != last) {
      for (uint32_t f = 0; f < c->nf; f++) {
        uint8_t* R = c->rows[f].as<uint8_t>();
        COLTT_HIP(hipMemcpyAsync(R + (size_t)s * c->stride, R + (size_t)last * c->stride, c->stride, hipMemcpyDeviceToDevice, c->stream));
        COLTT_HIP(hipMemcpyAsync(c->norms[f].as<float>() + s, c->norms[f].as<float>() + last, 4, hipMemcpyDeviceToDevice, c->stream));
      }
      uint64_t moved = c->h_ids[last]; c->h_ids[s] = moved; c->id2slot[moved] = s;
      COLTT_HIP(hipMemcpyAsync(c->ids.as<uint64_t>() + s, &c->h_ids[s], 8, hipMemcpyHostToDevice, c->stream));
    }
    c->h_ids.pop_back(); c->n--;
    COLTT_HIP(hipStreamSynchronize(c->stream));
  }
  return COLTT_OK;
}

/* MultiVertexSearch (multi_vector_vertex.go:85-137): queries is [nq][n_fields][dim]; ratios / include are per field.
 * Rows of out_* are DESCENDING by (score, id). */
int coltt_cflat_search(coltt_handle_t h, const float* queries, const uint32_t* ratios, const uint8_t* include, size_t nq, uint32_t k,
                       uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  auto c = lookup<CFlat>(h);
  if (!c) return fail(COLTT_E_NOT_FOUND, "cflat_search: unknown handle");
  if (nq == 0) return COLTT_OK;
  if (!queries || !ratios || !include || !out_ids || !out_scores || !out_counts) return fail(COLTT_E_INVALID, "cflat_search: NULL buffer");
  if (k == 0 || k > K_MAX) return fail(COLTT_E_UNSUPPORTED, "cflat_search: k=%u outside [1,%u]", k, K_MAX);
  WriteLock g(c->rw);
  COLTT_DEVICE(c->device);
  const size_t per = (size_t)c->nf * c->dim;
  const uint32_t cap = std::max<uint32_t>(65536u, 8u * k);
  COLTT_TRY(c->w_raw.reserve(per * 4)); COLTT_TRY(c->w_q.reserve(per * 4)); COLTT_TRY(c->w_qn.reserve(c->nf * 4 + 256));
  COLTT_TRY(c->w_misc.reserve(4096)); COLTT_TRY(c->w_cand.reserve((size_t)cap * 8));
  COLTT_TRY(c->w_out_ids.reserve((size_t)k * 8)); COLTT_TRY(c->w_out_sc.reserve((size_t)k * 4)); COLTT_TRY(c->w_out_cnt.reserve(4));
  uint32_t* cnt = c->w_misc.as<uint32_t>(); uint32_t* thr = cnt + 256; uint32_t* ovf = cnt + 512;
  uint32_t* d_ratio = cnt + 600; uint8_t* d_inc = reinterpret_cast<uint8_t*>(cnt + 640); float* d_w = reinterpret_cast<float*>(cnt + 700);
  COLTT_HIP(hipMemcpyAsync(d_ratio, ratios, c->nf * 4, hipMemcpyHostToDevice, c->stream));
  COLTT_HIP(hipMemcpyAsync(d_inc, include, c->nf, hipMemcpyHostToDevice, c->stream));
  cflat_weights_kernel<<<1, 64, 0, c->stream>>>(d_ratio, d_inc, (int)c->nf, d_w);
  CFields F{};
  for (uint32_t f = 0; f < c->nf; f++) { F.rows[f] = c->rows[f].as<uint8_t>(); F.norms[f] = c->norms[f].as<float>(); }
  std::vector<uint64_t> hi(k); std::vector<float> hs(k);
  for (size_t qi = 0; qi < nq; qi++) {
    COLTT_HIP(hipMemcpyAsync(c->w_raw.p, queries + qi * per, per * 4, hipMemcpyHostToDevice, c->stream));
    // included fields are normalised for cosine (multi_vector_vertex.go:96-100); excluded ones are never read
    launch_prep_queries<Q_NONE>(c->stream, c->w_raw.as<float>(), c->nf, (int)c->dim, c->metric == COLTT_COSINE, c->w_q.as<float>());
    query_norms_kernel<<<1, 64, 0, c->stream>>>(c->w_q.as<float>(), c->nf, (int)c->dim, c->w_qn.as<float>());
    init_group_kernel<<<1, 256, 0, c->stream>>>(cnt, thr, ovf, 0);
    auto scan = [&](uint64_t b, uint64_t e) {
      if (e > b) {
        uint64_t groups = (e - b + 31) / 32;
        uint32_t grid = (uint32_t)std::min<uint64_t>((groups + 3) / 4, 2048);
        size_t lds = per * 4;
        if (c->metric == COLTT_COSINE) {
          auto kern = cflat_scan_kernel<M_COS>;
          if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          kern<<<grid, 256, lds, c->stream>>>(F, c->stride, c->n, (int)c->nf, (int)c->dim, c->w_q.as<float>(), c->w_qn.as<float>(), d_w, thr,
                                              c->w_cand.as<unsigned long long>(), cnt, cap, b, e);
        } else {
          auto kern = cflat_scan_kernel<M_L2>;
          if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          kern<<<grid, 256, lds, c->stream>>>(F, c->stride, c->n, (int)c->nf, (int)c->dim, c->w_q.as<float>(), c->w_qn.as<float>(), d_w, thr,
                                              c->w_cand.as<unsigned long long>(), cnt, cap, b, e);
        }
      }
      flat_select_kernel<<<1, 256, 0, c->stream>>>(c->w_cand.as<unsigned long long>(), cnt, thr, cap, k, 0, c->ids.as<uint64_t>(), 0, ovf,
                                                   c->w_out_ids.as<uint64_t>(), c->w_out_sc.as<float>(), c->w_out_cnt.as<uint32_t>());
    };
    // segments of at most cap - k vertices can never overflow the candidate list
    const uint64_t seg = cap - std::min<uint32_t>(k, cap / 2);
    if (c->n == 0) scan(0, 0);
    for (uint64_t b = 0; b < c->n; b += seg) scan(b, std::min<uint64_t>(c->n, b + seg));
    uint32_t hc = 0;
    COLTT_HIP(hipMemcpyAsync(&hc, c->w_out_cnt.p, 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(hi.data(), c->w_out_ids.p, (size_t)k * 8, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(hs.data(), c->w_out_sc.p, (size_t)k * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipStreamSynchronize(c->stream));
    out_counts[qi] = hc;
    for (uint32_t j = 0; j < hc; j++) { out_ids[qi * k + j] = hi[hc - 1 - j]; out_scores[qi * k + j] = hs[hc - 1 - j]; }  // ascending -> descending
  }
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

}  // extern "C"
