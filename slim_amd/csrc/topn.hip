// topn.hip -- top-N recommendation for every user on the GPU.
//
// What it computes is GetRecommendations of the reference
// (/root/reference/src/libslim/predict.c:15-71) applied to every row of a history matrix
// (Py_SLIM_Predict, src/libslim/pyapi.c:530-563): the score of candidate k is the sum over
// the user's history items i of rating_i * W[i,k] (row i of the model), items of the
// history are excluded, the N best candidates are returned in descending score order.
//
// One wavefront per user, results BIT-IDENTICAL to the library's host path
// (host_csr.cpp::top_n):
//   * history rows are walked in order and the nnz of a row go to different lanes (ids in a
//     row are distinct), so every candidate receives its float additions in exactly the
//     host's order; products and sums are rounded separately (no FMA contraction);
//   * ties are broken by discovery order like the host (the reference's gk_fkvsortd leaves
//     tie order undefined): the first touch of a candidate records (history index, position
//     in the W row), which sorts like the host's discovery counter;
//   * selection: one pass over the score vector with a per-lane sorted list of the N best
//     (LDS), then N rounds of a wave-wide arg-max over the 64 list heads.
// The score/discovery vectors (12 bytes per item) live in a per-wavefront HBM slab.
#include <hip/hip_runtime.h>

#include <new>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "engine.hpp"
#include "host_csr.hpp"

namespace slimamd {

namespace {

constexpr unsigned long long kUntouched = ~0ull;
constexpr unsigned long long kExcluded = ~0ull - 1ull;

struct TopNArgs {
  int32_t nusers, nitems_rows, ncols, nrcmds;
  const int64_t* wptr;
  const int32_t* wind;
  const float* wval;
  const int64_t* hptr;
  const int32_t* hind;
  const float* hval;  // nullptr: implicit ratings of 1
  float* score;                // [nwaves][ncols]
  unsigned long long* disc;    // [nwaves][ncols]
  int32_t* out_ids;
  float* out_scores;
  int32_t* out_cnt;
  int32_t* queue;
};

// a candidate is "better" when its score is higher, or equal with an earlier discovery
__device__ __forceinline__ bool better(float sa, unsigned long long da, float sb,
                                       unsigned long long db) {
  return sa > sb || (sa == sb && da < db);
}

__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Control flow is wave-uniform wherever the data allows it: users are strided statically
// over the wavefronts, loop bounds over history rows and output ranks are scalars, and the
// only divergent loops are the lane-strided walks and the per-lane list insertion.
__global__ __launch_bounds__(64) void topn_kernel(const TopNArgs T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int N = T.nrcmds;
  // per-lane sorted lists, lane-major so a lane's slots sit in different banks
  float* l_score = reinterpret_cast<float*>(smem);                                       // [N][64]
  unsigned long long* l_disc = reinterpret_cast<unsigned long long*>(l_score + N * 64);  // [N][64]
  int* l_id = reinterpret_cast<int*>(l_disc + N * 64);                                   // [N][64]

  float* score = T.score + (int64_t)blockIdx.x * T.ncols;
  unsigned long long* disc = T.disc + (int64_t)blockIdx.x * T.ncols;

  for (int u = (int)blockIdx.x; u < T.nusers; u += (int)gridDim.x) {
    const int64_t h0 = uni64(T.hptr[u]), h1 = uni64(T.hptr[u + 1]);

    for (int k = lane; k < T.ncols; k += 64) disc[k] = kUntouched;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // history items are never recommended (predict.c:35-38)
    for (int64_t h = h0 + lane; h < h1; h += 64) {
      const int i = T.hind[h];
      if (i >= 0 && i < T.ncols) disc[i] = kExcluded;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

    // accumulate, history order outside, W-row entries across lanes (predict.c:40-58)
    for (int64_t h = h0; h < h1; ++h) {
      const int i = __builtin_amdgcn_readfirstlane(T.hind[h]);
      if (i >= 0 && i < T.nitems_rows) {
        const float rating = T.hval ? T.hval[h] : 1.0f;
        const int64_t w0 = uni64(T.wptr[i]), w1 = uni64(T.wptr[i + 1]);
        for (int64_t j = w0 + lane; j < w1; j += 64) {
          const int k = T.wind[j];
          const unsigned long long d = disc[k];
          if (d != kExcluded) {
            // the host scorer rounds the product and the sum separately: no FMA here
#pragma clang fp contract(off)
            float acc = 0.0f;
            if (d == kUntouched)
              disc[k] = ((unsigned long long)(h - h0) << 32) | (unsigned long long)(j - w0);
            else
              acc = score[k];
            const float prod = rating * T.wval[j];
            score[k] = acc + prod;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
    }

    // per-lane N best of the lane's stride of the score vector
    int cnt = 0;  // entries in this lane's list
    for (int k = lane; k < T.ncols; k += 64) {
      const unsigned long long d = disc[k];
      const float sc = score[k];
      bool want = d < kExcluded;
      if (want && cnt == N)
        want = better(sc, d, l_score[(N - 1) * 64 + lane], l_disc[(N - 1) * 64 + lane]);
      if (want) {
        int pos = cnt < N ? cnt : N - 1;  // insertion from the tail
        while (pos > 0 &&
               better(sc, d, l_score[(pos - 1) * 64 + lane], l_disc[(pos - 1) * 64 + lane])) {
          l_score[pos * 64 + lane] = l_score[(pos - 1) * 64 + lane];
          l_disc[pos * 64 + lane] = l_disc[(pos - 1) * 64 + lane];
          l_id[pos * 64 + lane] = l_id[(pos - 1) * 64 + lane];
          --pos;
        }
        l_score[pos * 64 + lane] = sc;
        l_disc[pos * 64 + lane] = d;
        l_id[pos * 64 + lane] = k;
        if (cnt < N) ++cnt;
      }
    }

    // N rounds: the best of the 64 list heads wins and is popped
    int head = 0, nout = 0;
    for (int r = 0; r < N; ++r) {
      const bool has = head < cnt;
      float bs = has ? l_score[head * 64 + lane] : 0.0f;
      unsigned long long bd = has ? l_disc[head * 64 + lane] : kUntouched;  // empty sorts last
      const int my_id = has ? l_id[head * 64 + lane] : 0;
      int bl = lane;
      int bh = has ? 1 : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_xor(bs, off);
        const unsigned int od_lo = __shfl_xor((unsigned int)bd, off);
        const unsigned int od_hi = __shfl_xor((unsigned int)(bd >> 32), off);
        const unsigned long long od = ((unsigned long long)od_hi << 32) | od_lo;
        const int ol = __shfl_xor(bl, off);
        const int oh = __shfl_xor(bh, off);
        const bool take = oh != 0 && (bh == 0 || better(os, od, bs, bd));
        bs = take ? os : bs;
        bd = take ? od : bd;
        bl = take ? ol : bl;
        bh = take ? oh : bh;
      }
      const int winner = __builtin_amdgcn_readfirstlane(bl);
      const int any = __builtin_amdgcn_readfirstlane(bh);
      const int id = __shfl(my_id, winner);
      if (any) {
        if (lane == 0) {
          T.out_ids[(int64_t)u * N + r] = id;
          T.out_scores[(int64_t)u * N + r] = bs;
        }
        if (lane == winner) ++head;
        ++nout;
      }
    }
    if (lane == 0) T.out_cnt[u] = nout;
  }
}

struct HipFail {
  hipError_t code;
  const char* where;
};
#define TOPN_TRY(expr)                                          \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) throw HipFail{_e, #expr};             \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  explicit DevBuf(size_t n) { TOPN_TRY(hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (n ? n : 1))); }
  ~DevBuf() { (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

}  // namespace

// Top-N lists of every history row.  output/scores are [nusers][nrcmds], slots beyond a
// user's list length are left as the caller filled them; counts (optional) = list lengths.
int32_t predict_device(const slim_csr_t* W, const slim_csr_t* hist, int32_t nrcmds,
                       int32_t* output, float* scores, int32_t* counts) {
  if (!W || !hist || !W->rowptr || !hist->rowptr || nrcmds < 1 || nrcmds > 128) {
    set_error("SLIMGPU_Predict: bad arguments (1 <= nrcmds <= 128)");
    return SLIM_ERROR_INPUT;
  }
  const int32_t nusers = hist->nrows;
  const int32_t ncols = std::max(W->ncols, 1);
  const int64_t wnnz = W->rowptr[W->nrows], hnnz = hist->rowptr[nusers];
  try {
    int ndev = 0;
    TOPN_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw HipFail{hipErrorNoDevice, "hipGetDeviceCount"};
    hipDeviceProp_t prop;
    int dev = 0;
    TOPN_TRY(hipGetDevice(&dev));
    TOPN_TRY(hipGetDeviceProperties(&prop, dev));
    const size_t lds = (size_t)nrcmds * 64 * (sizeof(float) + sizeof(unsigned long long) + sizeof(int));
    int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (128 * 1024) / lds));
    const int nwaves = std::max(1, std::min<int>(nusers, prop.multiProcessorCount * per_cu));

    DevBuf<int64_t> d_wptr((size_t)W->nrows + 1), d_hptr((size_t)nusers + 1);
    DevBuf<int32_t> d_wind((size_t)wnnz), d_hind((size_t)hnnz);
    DevBuf<float> d_wval((size_t)wnnz), d_hval(hist->rowval ? (size_t)hnnz : 1);
    DevBuf<float> d_score((size_t)nwaves * ncols), d_oscore((size_t)nusers * nrcmds);
    DevBuf<unsigned long long> d_disc((size_t)nwaves * ncols);
    DevBuf<int32_t> d_oid((size_t)nusers * nrcmds), d_ocnt((size_t)nusers), d_queue(1);
    TOPN_TRY(hipMemcpy(d_wptr.p, W->rowptr, sizeof(int64_t) * ((size_t)W->nrows + 1), hipMemcpyHostToDevice));
    TOPN_TRY(hipMemcpy(d_hptr.p, hist->rowptr, sizeof(int64_t) * ((size_t)nusers + 1), hipMemcpyHostToDevice));
    if (wnnz) {
      TOPN_TRY(hipMemcpy(d_wind.p, W->rowind, sizeof(int32_t) * (size_t)wnnz, hipMemcpyHostToDevice));
      TOPN_TRY(hipMemcpy(d_wval.p, W->rowval, sizeof(float) * (size_t)wnnz, hipMemcpyHostToDevice));
    }
    if (hnnz) {
      TOPN_TRY(hipMemcpy(d_hind.p, hist->rowind, sizeof(int32_t) * (size_t)hnnz, hipMemcpyHostToDevice));
      if (hist->rowval)
        TOPN_TRY(hipMemcpy(d_hval.p, hist->rowval, sizeof(float) * (size_t)hnnz, hipMemcpyHostToDevice));
    }
    TOPN_TRY(hipMemset(d_queue.p, 0, sizeof(int32_t)));
    TOPN_TRY(hipMemset(d_ocnt.p, 0, sizeof(int32_t) * (size_t)nusers));

    TopNArgs T;
    T.nusers = nusers;
    T.nitems_rows = W->nrows;
    T.ncols = ncols;
    T.nrcmds = nrcmds;
    T.wptr = d_wptr.p; T.wind = d_wind.p; T.wval = d_wval.p;
    T.hptr = d_hptr.p; T.hind = d_hind.p; T.hval = hist->rowval ? d_hval.p : nullptr;
    T.score = d_score.p; T.disc = d_disc.p;
    T.out_ids = d_oid.p; T.out_scores = d_oscore.p; T.out_cnt = d_ocnt.p; T.queue = d_queue.p;
    if (lds > 64 * 1024)
      TOPN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(topn_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(topn_kernel, dim3(nwaves), dim3(64), lds, 0, T);
    TOPN_TRY(hipGetLastError());
    TOPN_TRY(hipDeviceSynchronize());

    std::vector<int32_t> h_id((size_t)nusers * nrcmds), h_cnt((size_t)nusers);
    std::vector<float> h_sc((size_t)nusers * nrcmds);
    TOPN_TRY(hipMemcpy(h_id.data(), d_oid.p, sizeof(int32_t) * h_id.size(), hipMemcpyDeviceToHost));
    TOPN_TRY(hipMemcpy(h_sc.data(), d_oscore.p, sizeof(float) * h_sc.size(), hipMemcpyDeviceToHost));
    TOPN_TRY(hipMemcpy(h_cnt.data(), d_ocnt.p, sizeof(int32_t) * h_cnt.size(), hipMemcpyDeviceToHost));
    for (int32_t u = 0; u < nusers; ++u) {
      for (int32_t r = 0; r < h_cnt[u]; ++r) {
        output[(int64_t)u * nrcmds + r] = h_id[(size_t)u * nrcmds + r];
        scores[(int64_t)u * nrcmds + r] = h_sc[(size_t)u * nrcmds + r];
      }
      if (counts) counts[u] = h_cnt[u];
    }
    return SLIM_OK;
  } catch (const HipFail& e) {
    set_error(std::string("SLIMGPU_Predict: HIP error '") + hipGetErrorString(e.code) + "' in " +
              e.where);
    return e.code == hipErrorOutOfMemory ? SLIM_ERROR_MEMORY : SLIM_ERROR;
  } catch (const std::bad_alloc&) {
    set_error("SLIMGPU_Predict: out of host memory");
    return SLIM_ERROR_MEMORY;
  }
}

}  // namespace slimamd
