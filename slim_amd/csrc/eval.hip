// eval.hip -- the consumers of a learned model that the reference runs per user on the host,
// as HIP kernels:
//   * hit counting, HR / ARHR and the head / tail split of the leave-k-out protocol
//     (/root/reference/src/programs/slim_predict.c:181-236, slim_mselect.c:122-187,
//     src/libslim/pyapi.c:309-366), one user per lane, then ONE wavefront adding the users'
//     terms in user order with the reference's own arithmetic (float accumulators fed with
//     double terms, pyapi.c:223-230) -- so the four figures are the host loop's, bit for bit;
//   * the 1-vs-k protocol (src/libslim/predict.c:77-133, pyapi.c:483-528): every user ranks
//     a given list of candidate items; one wavefront per user, one candidate per lane, the
//     candidate's entry of each history item's model row found by binary search (model rows
//     ascending by id) and added in history order -- the float additions of the host loop in
//     the same order, hence the same scores; ties keep candidate order like the host's
//     stable sort.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "engine.hpp"
#include "host_csr.hpp"

namespace slimamd {

namespace {

struct HipFail {
  hipError_t code;
  const char* where;
};
#define EVAL_TRY(expr)                                          \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) throw HipFail{_e, #expr};             \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  explicit DevBuf(size_t n) { EVAL_TRY(hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (n ? n : 1))); }
  ~DevBuf() { (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  void upload(const T* src, size_t n) {
    if (n) EVAL_TRY(hipMemcpy(p, src, sizeof(T) * n, hipMemcpyHostToDevice));
  }
};

// ---- HR / ARHR -------------------------------------------------------------------------

struct UserTerms {      // what one user adds to the accumulators of pyapi.c:309-366
  double hr_all, hr_head, hr_tail;
  float arhr;
  int32_t flags;        // 1 valid, 2 has a head test item, 4 has a tail test item
};

__global__ void k_user_terms(int32_t nusers, int32_t nrcmds, const int32_t* __restrict__ lists,
                             const int32_t* __restrict__ counts,
                             const int64_t* __restrict__ tptr, const int32_t* __restrict__ tind,
                             const int32_t* __restrict__ fmarker, int32_t fm_ncols,
                             UserTerms* __restrict__ out) {
  for (int32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < nusers; u += gridDim.x * blockDim.x) {
    UserTerms t = {0.0, 0.0, 0.0, 0.0f, 0};
    const int64_t t0 = tptr[u], t1 = tptr[u + 1];
    if (t1 - t0 >= 1) {
      int ntrue[2] = {0, 0}, nhits[3] = {0, 0, 0};
      float gain = 0.0f, ideal = 0.0f;
      t.flags = 1;
      for (int64_t z = t0; z < t1; ++z) {
        const int32_t it = tind[z];
        const int cls = (it >= 0 && it < fm_ncols) ? fmarker[it] : 1;
        ++ntrue[cls];
        t.flags |= cls ? 4 : 2;
        ideal = (float)((double)ideal + 1.0 / (1.0 + double(z - t0)));
      }
      const int n = counts[u];
      for (int r = 0; r < n; ++r) {
        const int32_t id = lists[(int64_t)u * nrcmds + r];
        bool hit = false;
        for (int64_t z = t0; z < t1 && !hit; ++z) hit = tind[z] == id;
        if (hit) {
          const int cls = (id >= 0 && id < fm_ncols) ? fmarker[id] : 1;
          ++nhits[cls];
          ++nhits[2];
          gain = (float)((double)gain + 1.0 / (1.0 + r));
        }
      }
      t.hr_head = nhits[0] > 0 ? 1.0 * nhits[0] / ntrue[0] : 0.0;
      t.hr_tail = nhits[1] > 0 ? 1.0 * nhits[1] / ntrue[1] : 0.0;
      t.hr_all = 1.0 * nhits[2] / double(t1 - t0);
      t.arhr = gain / ideal;
    }
    out[u] = t;
  }
}

// one wavefront: 64 users' terms per coalesced load, added by lane order = user order
__global__ __launch_bounds__(64) void k_sum_in_user_order(int32_t nusers,
                                                          const UserTerms* __restrict__ terms,
                                                          float* __restrict__ out_f,
                                                          int32_t* __restrict__ out_n) {
  const int lane = threadIdx.x;
  float hr_all = 0, hr_head = 0, hr_tail = 0, arhr = 0;
  int nvalid = 0, nhead = 0, ntail = 0;
  for (int32_t b = 0; b < nusers; b += 64) {
    const int32_t u = b + lane;
    UserTerms t = {0.0, 0.0, 0.0, 0.0f, 0};
    if (u < nusers) t = terms[u];
    const int cnt = nusers - b < 64 ? nusers - b : 64;
    for (int k = 0; k < cnt; ++k) {
      const int fl = __shfl(t.flags, k);
      if (!(fl & 1)) continue;
      const double a = __shfl(t.hr_all, k), h = __shfl(t.hr_head, k), tl = __shfl(t.hr_tail, k);
      const float ar = __shfl(t.arhr, k);
      ++nvalid;
      nhead += (fl & 2) ? 1 : 0;
      ntail += (fl & 4) ? 1 : 0;
      hr_head = (float)((double)hr_head + h);   // float += double, as the host loop
      hr_tail = (float)((double)hr_tail + tl);
      hr_all = (float)((double)hr_all + a);
      arhr += ar;
    }
  }
  if (lane == 0) {
    out_f[0] = hr_all; out_f[1] = hr_head; out_f[2] = hr_tail; out_f[3] = arhr;
    out_n[0] = nvalid; out_n[1] = nhead; out_n[2] = ntail;
  }
}

// ---- 1-vs-k ----------------------------------------------------------------------------

__global__ void k_rows_ascending(int32_t nrows, const int64_t* __restrict__ ptr,
                                 const int32_t* __restrict__ ind, int32_t* __restrict__ bad) {
  for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x)
    for (int64_t k = ptr[r] + 1; k < ptr[r + 1]; ++k)
      if (ind[k] <= ind[k - 1]) {
        *bad = 1;
        break;
      }
}

constexpr int kMaxCandPerLane = 16;  // up to 1024 candidates per user

__global__ __launch_bounds__(64) void k_topn_1vsk(int32_t nusers, int32_t wrows, int32_t ncols,
                                                  int32_t nrcmds, int32_t nnegs,
                                                  const int64_t* __restrict__ wptr,
                                                  const int32_t* __restrict__ wind,
                                                  const float* __restrict__ wval,
                                                  const int64_t* __restrict__ hptr,
                                                  const int32_t* __restrict__ hind,
                                                  const float* __restrict__ hval,
                                                  const int32_t* __restrict__ negitems,
                                                  int32_t* __restrict__ out_ids,
                                                  float* __restrict__ out_scores) {
  const int lane = threadIdx.x;
  const int per = (nnegs + 63) / 64;
  for (int32_t u = blockIdx.x; u < nusers; u += gridDim.x) {
    const int32_t* neg = negitems + (int64_t)u * nnegs;
    int cid[kMaxCandPerLane];
    float key[kMaxCandPerLane];
    bool live[kMaxCandPerLane];  // receives scores: a valid id not repeated later in the list
#pragma unroll
    for (int k = 0; k < kMaxCandPerLane; ++k) {
      const int c = lane + 64 * k;
      cid[k] = (k < per && c < nnegs) ? neg[c] : -1;
      key[k] = 0.0f;
      live[k] = k < per && c < nnegs && cid[k] >= 0 && cid[k] < ncols;
      // predict.c:92-99: the position table keeps the LAST occurrence of a repeated id
      if (live[k])
        for (int c2 = c + 1; c2 < nnegs; ++c2)
          if (neg[c2] == cid[k]) {
            live[k] = false;
            break;
          }
    }
    const int64_t h0 = hptr[u], h1 = hptr[u + 1];
    for (int64_t e = h0; e < h1; ++e) {  // history order, as the host loop
      const int32_t i = hind[e];
      if (i < 0 || i >= wrows) continue;
      const float rating = hval ? hval[e] : 1.0f;
      const int64_t s = wptr[i], t = wptr[i + 1];
#pragma unroll
      for (int k = 0; k < kMaxCandPerLane; ++k) {
        if (!live[k]) continue;
        int64_t lo = s, hi = t;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (wind[mid] < cid[k]) lo = mid + 1; else hi = mid;
        }
        // product and sum rounded separately, like the host's `key += rating * w`
        if (lo < t && wind[lo] == cid[k]) key[k] = __fadd_rn(key[k], __fmul_rn(rating, wval[lo]));
      }
    }
    // emit_best: descending score, ties in candidate order; nrcmds rounds of a wave arg-max
    const int n = nnegs < nrcmds ? nnegs : nrcmds;
    bool used[kMaxCandPerLane];
#pragma unroll
    for (int k = 0; k < kMaxCandPerLane; ++k) used[k] = !(k < per && lane + 64 * k < nnegs);
    for (int r = 0; r < n; ++r) {
      float bs = -__builtin_huge_valf();
      int bc = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < kMaxCandPerLane; ++k) {
        const int c = lane + 64 * k;
        if (!used[k] && (key[k] > bs || (key[k] == bs && c < bc))) {
          bs = key[k];
          bc = c;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_xor(bs, off);
        const int oc = __shfl_xor(bc, off);
        if (os > bs || (os == bs && oc < bc)) {
          bs = os;
          bc = oc;
        }
      }
      if ((bc & 63) == lane) {
        const int k = bc >> 6;
#pragma unroll
        for (int kk = 0; kk < kMaxCandPerLane; ++kk)
          if (kk == k) {
            used[kk] = true;
            out_ids[(int64_t)u * nrcmds + r] = neg[bc];
            out_scores[(int64_t)u * nrcmds + r] = key[kk];
          }
      }
    }
  }
}

int cu_count() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

int32_t fail(const char* who, const HipFail& e) {
  set_error(std::string(who) + ": HIP error '" + hipGetErrorString(e.code) + "' in " + e.where);
  return e.code == hipErrorOutOfMemory ? SLIM_ERROR_MEMORY : SLIM_ERROR;
}

}  // namespace

int32_t evaluate_device(int32_t nusers, int32_t nrcmds, const int32_t* lists, const int32_t* counts,
                        const slim_csr_t* tst, const int32_t* fmarker, int32_t fm_ncols,
                        EvalResult* out) {
  if (!lists || !counts || !tst || !tst->rowptr || !fmarker || !out || nrcmds < 1) {
    set_error("SLIMGPU_Evaluate: bad arguments");
    return SLIM_ERROR_INPUT;
  }
  nusers = std::min(nusers, tst->nrows);
  *out = EvalResult();
  if (nusers <= 0) return SLIM_OK;
  try {
    (void)hipGetLastError();
    const int64_t tnnz = tst->rowptr[nusers];
    DevBuf<int32_t> d_lists((size_t)nusers * nrcmds), d_counts((size_t)nusers), d_tind((size_t)tnnz),
        d_fm((size_t)std::max(fm_ncols, 1)), d_n(3);
    DevBuf<int64_t> d_tptr((size_t)nusers + 1);
    DevBuf<UserTerms> d_terms((size_t)nusers);
    DevBuf<float> d_f(4);
    d_lists.upload(lists, (size_t)nusers * nrcmds);
    d_counts.upload(counts, (size_t)nusers);
    static_assert(sizeof(ssize_t) == sizeof(int64_t), "LP64 expected");
    d_tptr.upload(reinterpret_cast<const int64_t*>(tst->rowptr), (size_t)nusers + 1);
    d_tind.upload(tst->rowind, (size_t)tnnz);
    d_fm.upload(fmarker, (size_t)fm_ncols);
    const int blocks = std::max(1, std::min((nusers + 255) / 256, cu_count() * 8));
    hipLaunchKernelGGL(k_user_terms, dim3(blocks), dim3(256), 0, 0, nusers, nrcmds, d_lists.p,
                       d_counts.p, d_tptr.p, d_tind.p, d_fm.p, fm_ncols, d_terms.p);
    EVAL_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_sum_in_user_order, dim3(1), dim3(64), 0, 0, nusers, d_terms.p, d_f.p, d_n.p);
    EVAL_TRY(hipGetLastError());
    float f[4];
    int32_t n[3];
    EVAL_TRY(hipMemcpy(f, d_f.p, sizeof(f), hipMemcpyDeviceToHost));
    EVAL_TRY(hipMemcpy(n, d_n.p, sizeof(n), hipMemcpyDeviceToHost));
    out->nvalid = n[0];
    out->nvalid_head = n[1];
    out->nvalid_tail = n[2];
    out->hr = n[0] > 0 ? f[0] / n[0] : 0;
    out->hr_head = n[1] > 0 ? f[1] / n[1] : 0;
    out->hr_tail = n[2] > 0 ? f[2] / n[2] : 0;
    out->arhr = n[0] > 0 ? f[3] / n[0] : 0;
    return SLIM_OK;
  } catch (const HipFail& e) {
    return fail("SLIMGPU_Evaluate", e);
  }
}

int32_t predict_1vsk_device(const slim_csr_t* W, const slim_csr_t* hist, int32_t nrcmds,
                            int32_t nnegs, const int32_t* negitems, int32_t* output,
                            float* scores) {
  if (!W || !hist || !W->rowptr || !hist->rowptr || nrcmds < 1 || nnegs < 1 || !negitems ||
      nnegs > 64 * kMaxCandPerLane) {
    set_error("SLIMGPU_Predict1vsK: bad arguments (1 <= nnegs <= 1024)");
    return SLIM_ERROR_INPUT;
  }
  const int32_t nusers = hist->nrows;
  if (nusers <= 0) return SLIM_ERROR;
  try {
    (void)hipGetLastError();
    const int64_t wnnz = W->rowptr[W->nrows], hnnz = hist->rowptr[nusers];
    DevBuf<int64_t> d_wptr((size_t)W->nrows + 1), d_hptr((size_t)nusers + 1);
    DevBuf<int32_t> d_wind((size_t)wnnz), d_hind((size_t)hnnz), d_neg((size_t)nusers * nnegs),
        d_oid((size_t)nusers * nrcmds), d_bad(1);
    DevBuf<float> d_wval((size_t)wnnz), d_hval(hist->rowval ? (size_t)hnnz : 1),
        d_osc((size_t)nusers * nrcmds);
    d_wptr.upload(reinterpret_cast<const int64_t*>(W->rowptr), (size_t)W->nrows + 1);
    d_hptr.upload(reinterpret_cast<const int64_t*>(hist->rowptr), (size_t)nusers + 1);
    d_wind.upload(W->rowind, (size_t)wnnz);
    d_wval.upload(W->rowval, (size_t)wnnz);
    d_hind.upload(hist->rowind, (size_t)hnnz);
    if (hist->rowval) d_hval.upload(hist->rowval, (size_t)hnnz);
    d_neg.upload(negitems, (size_t)nusers * nnegs);
    EVAL_TRY(hipMemset(d_bad.p, 0, sizeof(int32_t)));
    const int cus = cu_count();
    hipLaunchKernelGGL(k_rows_ascending, dim3(std::max(1, std::min(W->nrows / 256 + 1, cus * 8))),
                       dim3(256), 0, 0, W->nrows, d_wptr.p, d_wind.p, d_bad.p);
    EVAL_TRY(hipGetLastError());
    int32_t bad = 0;
    EVAL_TRY(hipMemcpy(&bad, d_bad.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (bad) {
      set_error("SLIMGPU_Predict1vsK: model rows are not ascending by item id");
      return SLIM_ERROR_INPUT;
    }
    hipLaunchKernelGGL(k_topn_1vsk, dim3(std::max(1, std::min(nusers, cus * 16))), dim3(64), 0, 0,
                       nusers, W->nrows, W->ncols, nrcmds, nnegs, d_wptr.p, d_wind.p, d_wval.p,
                       d_hptr.p, d_hind.p, hist->rowval ? d_hval.p : nullptr, d_neg.p, d_oid.p,
                       d_osc.p);
    EVAL_TRY(hipGetLastError());
    const int32_t n = std::min(nnegs, nrcmds);
    std::vector<int32_t> h_id((size_t)nusers * nrcmds);
    std::vector<float> h_sc((size_t)nusers * nrcmds);
    EVAL_TRY(hipMemcpy(h_id.data(), d_oid.p, sizeof(int32_t) * h_id.size(), hipMemcpyDeviceToHost));
    EVAL_TRY(hipMemcpy(h_sc.data(), d_osc.p, sizeof(float) * h_sc.size(), hipMemcpyDeviceToHost));
    for (int32_t u = 0; u < nusers; ++u)
      for (int32_t r = 0; r < n; ++r) {
        output[(int64_t)u * nrcmds + r] = h_id[(size_t)u * nrcmds + r];
        scores[(int64_t)u * nrcmds + r] = h_sc[(size_t)u * nrcmds + r];
      }
    return SLIM_OK;
  } catch (const HipFail& e) {
    return fail("SLIMGPU_Predict1vsK", e);
  }
}

}  // namespace slimamd
