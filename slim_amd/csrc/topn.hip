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
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

// ---- second kernel: score chunks in LDS -------------------------------------------------
//
// topn_kernel keeps the 12-byte-per-item score/discovery vectors of a user in HBM, so every
// multiply-add costs ~4 random sector requests: measured on a C4-shaped model (100K items,
// 2700 entries per row, histories of ~890 items) it is bound by the request rate at
// 14e9 adds/s = 5.8K users/s -- slower than the host scorer on a 128-core box.
//
// Here a workgroup of 8 wavefronts serves one user and the ITEMS are cut into chunks of CW ids
// whose score/discovery arrays live in LDS (12 bytes x CW per wavefront).  Wavefront w owns
// chunks w, w+8, ...; for each of its chunks it walks the user's history in order and, for
// history item i, only the entries of row i of W whose ids fall into the chunk -- rows are
// sorted, and wsplit[i][c] (built once per call) is where chunk c starts in row i.  Every
// candidate still receives its additions in history order, products and sums rounded
// separately, and the first touch still records (history index, position in the row), so the
// result is bit-identical to topn_kernel and to the host.  HBM sees each W entry once per
// user, in coalesced segments; everything else is LDS.
//   * the (start, length, rating) of up to 64 history items are fetched lane-parallel, then
//     consumed one item at a time through v_readlane; the segment loads run kT2Depth items
//     ahead of the LDS updates;
//   * selection: a wavefront keeps its N best in REGISTERS (lane t = rank t); a candidate that
//     beats the current N-th is inserted with one ballot + one lane shift; the 8 lists are
//     merged through LDS by wavefront 0.
constexpr int kT2Waves = 8;       // default workgroup: 8 wavefronts x 1536-id chunks
constexpr int kT2Depth = 16;
constexpr int kT2MaxN = 64;       // lists live one rank per lane: up to a wavefront's width
constexpr int kT2MaxCW = 1536;

struct TopN2Args {
  int32_t nusers, nitems_rows, ncols, nrcmds;
  int32_t cw, nchunks;
  uint32_t wlast;    // nnz(W) - 1 (0 for an empty model): clamp for the unconditional loads
  int32_t pos_bits;  // discovery key = history index << pos_bits | position in the row
  const int64_t* wptr;
  const int32_t* wind;
  const float* wval;
  const uint32_t* wsplit;  // [nitems_rows][nchunks + 1]: offset in row i of the first id >= c * cw
  const int64_t* hptr;
  const int32_t* hind;
  const float* hval;
  int32_t* out_ids;
  float* out_scores;
  int32_t* out_cnt;
  int32_t* queue;
};

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1);
  const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long readlane_key(unsigned long long v, int l) { return readlane64(v, l); }
__device__ __forceinline__ uint32_t readlane_key(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ unsigned long long shfl_up_key(unsigned long long v) { return shfl_up64(v); }
__device__ __forceinline__ uint32_t shfl_up_key(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1); }
template <typename KeyT>
__device__ __forceinline__ bool better(float sa, KeyT da, float sb, KeyT db) {
  return sa > sb || (sa == sb && da < db);
}

// rows of W sorted by id?  (one wavefront per row; flag set when an inversion is found)
__global__ void k_rows_sorted(int32_t nrows, const int64_t* __restrict__ ptr,
                              const int32_t* __restrict__ ind, int32_t* __restrict__ unsorted) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const int64_t s = ptr[r], e = ptr[r + 1];
    bool bad = false;
    for (int64_t j = s + 1 + lane; j < e; j += 64) bad |= ind[j - 1] >= ind[j];
    if (bad) atomicExch(unsorted, 1);
  }
}

// wsplit[r][c] = number of ids of row r below c * cw (binary search; rows are sorted)
__global__ void k_row_split(int32_t nrows, int32_t nchunks, int32_t cw,
                            const int64_t* __restrict__ ptr, const int32_t* __restrict__ ind,
                            uint32_t* __restrict__ split) {
  const int64_t total = (int64_t)nrows * (nchunks + 1);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = (int32_t)(t / (nchunks + 1)), c = (int32_t)(t % (nchunks + 1));
    const int64_t s = ptr[r], e = ptr[r + 1];
    const int64_t bound = (int64_t)c * cw;
    int64_t lo = s, hi = e;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)ind[mid] < bound) lo = mid + 1; else hi = mid;
    }
    split[t] = (uint32_t)(lo - s);
  }
}

// KeyT: discovery key (history index << pos_bits | position in the model row).  32 bits when
// the longest history and the longest model row allow it (8 bytes of LDS per item: chunks of
// 2304 ids), else 64.
template <int NW, typename KeyT>
__global__ __launch_bounds__(64 * NW) void topn_chunk_kernel(const TopN2Args T) {
  constexpr KeyT kUnt = ~KeyT(0), kExc = ~KeyT(0) - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = kT2Depth;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = T.nrcmds, CW = T.cw;
  KeyT* disc = reinterpret_cast<KeyT*>(smem) + (size_t)wave * CW;
  float* score = reinterpret_cast<float*>(smem + (size_t)NW * CW * sizeof(KeyT)) + (size_t)wave * CW;
  char* marea = smem + (size_t)NW * CW * (sizeof(KeyT) + 4);
  float* m_s = reinterpret_cast<float*>(marea);
  KeyT* m_d = reinterpret_cast<KeyT*>(marea + NW * kT2MaxN * 4);
  int* m_id = reinterpret_cast<int*>(marea + NW * kT2MaxN * 12);
  int* m_cnt = reinterpret_cast<int*>(marea + NW * kT2MaxN * 16);
  __shared__ int s_user;

  for (;;) {
    if (tid == 0) s_user = atomicAdd(T.queue, 1);
    __syncthreads();
    const int u = __builtin_amdgcn_readfirstlane(s_user);
    if (u >= T.nusers) break;
    const int64_t h0 = uni64(T.hptr[u]), h1 = uni64(T.hptr[u + 1]);

    // this wavefront's N best so far: lane t holds rank t
    float ls = 0.0f;
    KeyT ld = kUnt;
    int lid = -1;
    int count = 0;
    float worst_s = 0.0f;
    KeyT worst_d = 0;
    auto insert = [&](const float cs, const KeyT cd, const int cid) {
      const bool ahead = lane < count && better(ls, ld, cs, cd);
      const int p = __popcll(__ballot(ahead));  // sorted list: the entries ahead are ranks 0..p-1
      const float us = __shfl_up(ls, 1);
      const KeyT ud = shfl_up_key(ld);
      const int uid = __shfl_up(lid, 1);
      if (lane > p && lane < N) {
        ls = us;
        ld = ud;
        lid = uid;
      }
      if (lane == p) {
        ls = cs;
        ld = cd;
        lid = cid;
      }
      if (count < N) ++count;
      if (count == N) {
        worst_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ls), N - 1));
        worst_d = readlane_key(ld, N - 1);
      }
    };
    auto offer = [&](const float cs, const KeyT cd, const int cid) {
      if (count < N || better(cs, cd, worst_s, worst_d)) insert(cs, cd, cid);
    };

    for (int c = wave; c < T.nchunks; c += NW) {
      const int base = c * CW;
      const int width = (T.ncols - base) < CW ? (T.ncols - base) : CW;
      for (int k = lane; k < width; k += 64) disc[k] = kUnt;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      for (int64_t h = h0 + lane; h < h1; h += 64) {  // history items are never recommended
        const int i = T.hind[h];
        if (i >= base && i < base + width) disc[i - base] = kExc;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

      // branch-free: both reads issue together; an untouched slot reads garbage as its score and
      // selects 0; an excluded slot keeps its key and accumulates a score nobody reads
      auto update = [&](const int idx, const KeyT key, const float prod) {
#pragma clang fp contract(off)
        const KeyT d = disc[idx];
        const float old = score[idx];
        const bool first = d == kUnt;
        disc[idx] = first ? key : d;
        score[idx] = (first ? 0.0f : old) + prod;
      };

      for (int64_t hb = h0; hb < h1; hb += 64) {
        const int nb = (h1 - hb) < 64 ? (int)(h1 - hb) : 64;
        // lane l: where history item hb + l meets this chunk
        uint32_t my_s = 0;  // element offset of the segment in wind / wval (nnz(W) < 2^31)
        int my_len = 0;
        uint32_t my_p0 = 0;
        float my_r = 1.0f;
        if (lane < nb) {
          const int i = T.hind[hb + lane];
          if (T.hval) my_r = T.hval[hb + lane];
          if (i >= 0 && i < T.nitems_rows) {
            const uint32_t* sp = T.wsplit + (int64_t)i * (T.nchunks + 1) + c;
            my_p0 = sp[0];
            my_len = (int)(sp[1] - my_p0);
            my_s = (uint32_t)T.wptr[i] + my_p0;
          }
        }
        // The segment loads are UNCONDITIONAL instructions (clamped address, predicate applied
        // when the entry is consumed) and every step issues exactly one fetch: the number of
        // loads in flight is then the same on every path, so the compiler can wait for the oldest
        // fetch only (s_waitcnt vmcnt(2*(D-1))).  With loads under `if (lane < len)` it had to
        // drain the queue at every step, and the kernel ran at one L2 round trip per step
        // whatever the depth.  Items past the batch have length 0: their steps do nothing.
        int qk[D], qlen[D];
        float qv[D];
        auto fetch = [&](const int l, int& k, float& v, int& len) {
          const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)my_s, l);
          len = __builtin_amdgcn_readlane(my_len, l);
          uint32_t j = s + (uint32_t)lane;
          j = j < T.wlast ? j : T.wlast;
          k = T.wind[j];
          v = T.wval[j];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(d, qk[d], qv[d], qlen[d]);
        const int nbp = (nb + D - 1) / D * D;  // <= 64: lanes past nb hold length 0
        for (int lb = 0; lb < nbp; lb += D) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            const int l = lb + d;
            const int kraw = qk[d];
            const float v = qv[d];
            const int len = qlen[d];
            fetch((l + D) & 63, qk[d], qv[d], qlen[d]);
            const float rating = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_r), l));
            const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)my_p0, l);
            const KeyT hkey = (KeyT)(uint32_t)(hb - h0 + l) << T.pos_bits;
            if (lane < len) {
#pragma clang fp contract(off)
              const float prod = rating * v;
              update(kraw - base, hkey | (KeyT)(p0 + (uint32_t)lane), prod);
            }
            if (len > 64) {  // a segment longer than one wavefront step (dense rows)
              const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)my_s, l);
              for (uint32_t t = 64 + (uint32_t)lane; t < (uint32_t)len; t += 64) {
#pragma clang fp contract(off)
                const float prod = rating * T.wval[s + t];
                update(T.wind[s + t] - base, hkey | (KeyT)(p0 + t), prod);
              }
            }
          }
        }
      }

      // candidates of this chunk against the wavefront's N best
      for (int kb = 0; kb < width; kb += 64) {
        const int k = kb + lane;
        KeyT d = kUnt;
        float sc = 0.0f;
        if (k < width) {
          d = disc[k];
          sc = score[k];
        }
        bool want = d < kExc;
        if (want && count == N) want = better(sc, d, worst_s, worst_d);
        unsigned long long mask = __ballot(want);
        while (mask) {
          const int l = __builtin_ctzll(mask);
          mask &= mask - 1;
          const float cs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), l));
          const KeyT cd = readlane_key(d, l);
          offer(cs, cd, base + kb + l);
        }
      }
    }

    // merge the wavefronts' lists (wavefront 0), write the user's row
    if (lane < kT2MaxN) {
      m_s[wave * kT2MaxN + lane] = ls;
      m_d[wave * kT2MaxN + lane] = ld;
      m_id[wave * kT2MaxN + lane] = lid;
    }
    if (lane == 0) m_cnt[wave] = count;
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < NW; ++w) {
        const int cw_ = __builtin_amdgcn_readfirstlane(m_cnt[w]);
        for (int t = 0; t < cw_; ++t) {
          const float cs = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m_s[w * kT2MaxN + t])));
          const KeyT cd = readlane_key(m_d[w * kT2MaxN + t], 0);
          const int cid = __builtin_amdgcn_readfirstlane(m_id[w * kT2MaxN + t]);
          offer(cs, cd, cid);
        }
      }
      if (lane < count) {
        T.out_ids[(int64_t)u * N + lane] = lid;
        T.out_scores[(int64_t)u * N + lane] = ls;
      }
      if (lane == 0) T.out_cnt[u] = count;
    }
    __syncthreads();
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
// W: its row view on the device -- uploaded by predict_device (host model), or where a resident model
// already holds it (predict_device_view: nothing of W crosses PCIe).
int32_t predict_device_view(const DeviceRowView& W, const slim_csr_t* hist, int32_t nrcmds,
                            int32_t* output, float* scores, int32_t* counts) {
  if (!hist || !hist->rowptr || !W.d_ptr || nrcmds < 1 || nrcmds > 128) {
    set_error("SLIMGPU_Predict: bad arguments (1 <= nrcmds <= 128)");
    return SLIM_ERROR_INPUT;
  }
  const int32_t nusers = hist->nrows;
  const int32_t ncols = std::max(W.ncols, 1);
  const int64_t wnnz = W.nnz, hnnz = hist->rowptr[nusers];
  const auto t_begin = std::chrono::steady_clock::now();
  try {
    (void)hipGetLastError();  // a failure of an earlier call must not be reported by this one
    int ndev = 0;
    TOPN_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw HipFail{hipErrorNoDevice, "hipGetDeviceCount"};
    hipDeviceProp_t prop;
    int dev = 0;
    TOPN_TRY(hipGetDevice(&dev));
    TOPN_TRY(hipGetDeviceProperties(&prop, dev));
    struct { const int64_t* p; } d_wptr{W.d_ptr};
    struct { const int32_t* p; } d_wind{W.d_ind};
    struct { const float* p; } d_wval{W.d_val};
    DevBuf<int64_t> d_hptr((size_t)nusers + 1);
    DevBuf<int32_t> d_hind((size_t)hnnz);
    DevBuf<float> d_hval(hist->rowval ? (size_t)hnnz : 1);
    DevBuf<float> d_oscore((size_t)nusers * nrcmds);
    DevBuf<int32_t> d_oid((size_t)nusers * nrcmds), d_ocnt((size_t)nusers), d_queue(2);
    TOPN_TRY(hipMemcpy(d_hptr.p, hist->rowptr, sizeof(int64_t) * ((size_t)nusers + 1), hipMemcpyHostToDevice));
    if (hnnz) {
      TOPN_TRY(hipMemcpy(d_hind.p, hist->rowind, sizeof(int32_t) * (size_t)hnnz, hipMemcpyHostToDevice));
      if (hist->rowval)
        TOPN_TRY(hipMemcpy(d_hval.p, hist->rowval, sizeof(float) * (size_t)hnnz, hipMemcpyHostToDevice));
    }
    TOPN_TRY(hipMemset(d_queue.p, 0, 2 * sizeof(int32_t)));
    TOPN_TRY(hipMemset(d_ocnt.p, 0, sizeof(int32_t) * (size_t)nusers));

    // kernel choice: score chunks in LDS (lists of up to 64, rows of W sorted by id), else the
    // one-wavefront-per-user kernel with its vectors in HBM.  SLIM_TOPN_KERNEL=wave|chunk and
    // SLIM_TOPN_CW=<chunk width> override (tests).
    const char* kenv = std::getenv("SLIM_TOPN_KERNEL");
    bool chunked = nrcmds <= kT2MaxN && wnnz < (int64_t(1) << 31) &&
                   !(kenv && std::strcmp(kenv, "wave") == 0);
    if (chunked) {
      int32_t unsorted = 0;
      if (wnnz > 0) {
        hipLaunchKernelGGL(k_rows_sorted, dim3(std::max(1, std::min(W.nrows / 4 + 1, prop.multiProcessorCount * 8))),
                           dim3(256), 0, 0, W.nrows, d_wptr.p, d_wind.p, d_queue.p + 1);
        TOPN_TRY(hipGetLastError());
        TOPN_TRY(hipMemcpy(&unsorted, d_queue.p + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
      }
      if (unsorted) chunked = false;
    }
    int t2w = kT2Waves;
    if (const char* e = std::getenv("SLIM_TOPN_WAVES")) t2w = std::atoi(e) == 16 ? 16 : 8;
    // discovery keys: 32 bits when (longest history, longest model row) fit, else 64
    int64_t max_hist = 0, max_row = 0;
    for (int32_t u = 0; u < nusers; ++u) max_hist = std::max<int64_t>(max_hist, hist->rowptr[u + 1] - hist->rowptr[u]);
    max_row = W.max_row;
    auto bits_for = [](int64_t v) { int b = 0; while ((int64_t(1) << b) <= v) ++b; return b; };
    bool key32 = bits_for(max_row) + bits_for(max_hist) <= 31;
    if (const char* e = std::getenv("SLIM_TOPN_KEY")) key32 = key32 && std::atoi(e) != 64;
    const int pos_bits = key32 ? bits_for(max_row) : 32;
    const int item_bytes = key32 ? 8 : 12;
    // chunk width: round 1's footprint (1536 ids x 12 bytes x 8 wavefronts), less whatever the
    // merge area of this geometry needs beyond it, so that chunks + lists always fit the 160 KB
    const size_t merge_bytes = (size_t)t2w * kT2MaxN * 16 + (size_t)t2w * sizeof(int) + 256;
    const size_t chunk_bytes = std::min<size_t>((size_t)kT2MaxCW * 12 * kT2Waves,
                                                (size_t)160 * 1024 - merge_bytes);
    const int max_cw = (int)(chunk_bytes / ((size_t)item_bytes * t2w)) / 64 * 64;
    int cw = std::max(64, std::min(max_cw, ((ncols + t2w - 1) / t2w + 63) / 64 * 64));
    if (const char* e = std::getenv("SLIM_TOPN_CW")) {
      const int v = std::atoi(e);
      if (v >= 64 && v <= max_cw && v % 64 == 0) cw = v;
    }
    const int nchunks = (ncols + cw - 1) / cw;
    if (chunked && (size_t)W.nrows * ((size_t)nchunks + 1) * sizeof(uint32_t) > (size_t(2) << 30))
      chunked = false;
    if (kenv && std::strcmp(kenv, "chunk") == 0 && !chunked) {
      set_error("SLIMGPU_Predict: SLIM_TOPN_KERNEL=chunk needs nrcmds <= 64 and model rows sorted by id");
      return SLIM_ERROR_INPUT;
    }

    if (chunked) {
      DevBuf<uint32_t> d_split((size_t)std::max(W.nrows, 1) * ((size_t)nchunks + 1));
      if (W.nrows > 0) {
        const int64_t total = (int64_t)W.nrows * (nchunks + 1);
        hipLaunchKernelGGL(k_row_split, dim3((unsigned)std::min<int64_t>((total + 255) / 256, prop.multiProcessorCount * 16)),
                           dim3(256), 0, 0, W.nrows, nchunks, cw, d_wptr.p, d_wind.p, d_split.p);
        TOPN_TRY(hipGetLastError());
      }
      TopN2Args T;
      T.nusers = nusers;
      T.nitems_rows = W.nrows;
      T.ncols = ncols;
      T.nrcmds = nrcmds;
      T.cw = cw;
      T.nchunks = nchunks;
      T.pos_bits = pos_bits;
      T.wlast = wnnz > 0 ? (uint32_t)(wnnz - 1) : 0u;
      T.wptr = d_wptr.p; T.wind = d_wind.p; T.wval = d_wval.p; T.wsplit = d_split.p;
      T.hptr = d_hptr.p; T.hind = d_hind.p; T.hval = hist->rowval ? d_hval.p : nullptr;
      T.out_ids = d_oid.p; T.out_scores = d_oscore.p; T.out_cnt = d_ocnt.p; T.queue = d_queue.p;
      const size_t lds = (size_t)t2w * cw * item_bytes + (size_t)t2w * kT2MaxN * 16 + t2w * sizeof(int);
      auto kfn = key32 ? (t2w == 16 ? topn_chunk_kernel<16, uint32_t> : topn_chunk_kernel<8, uint32_t>)
                       : (t2w == 16 ? topn_chunk_kernel<16, unsigned long long>
                                    : topn_chunk_kernel<8, unsigned long long>);
      if (lds > 64 * 1024)
        TOPN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(32 / t2w, (160 * 1024) / (lds + 64)));
      const int nwg = std::max(1, std::min<int>(nusers, prop.multiProcessorCount * per_cu));
      const auto t_k0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(kfn, dim3(nwg), dim3(64 * t2w), lds, 0, T);
      TOPN_TRY(hipGetLastError());
      TOPN_TRY(hipDeviceSynchronize());
      if (std::getenv("SLIM_GPU_TRACE"))
        std::fprintf(stderr, "[trace] top-N chunk kernel: %d users, %d workgroups of %d wavefronts, chunks of %d ids, "
                             "%d-bit keys: %.1f ms (upload + split table before it: %.1f ms)\n",
                     nusers, nwg, t2w, cw, key32 ? 32 : 64,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_k0).count(),
                     std::chrono::duration<double, std::milli>(t_k0 - t_begin).count());
    } else {
      const size_t lds = (size_t)nrcmds * 64 * (sizeof(float) + sizeof(unsigned long long) + sizeof(int));
      int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (128 * 1024) / lds));
      const int nwaves = std::max(1, std::min<int>(nusers, prop.multiProcessorCount * per_cu));
      DevBuf<float> d_score((size_t)nwaves * ncols);
      DevBuf<unsigned long long> d_disc((size_t)nwaves * ncols);
      TopNArgs T;
      T.nusers = nusers;
      T.nitems_rows = W.nrows;
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
    }

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


int32_t predict_device(const slim_csr_t* W, const slim_csr_t* hist, int32_t nrcmds,
                       int32_t* output, float* scores, int32_t* counts) {
  if (!W || !hist || !W->rowptr || !hist->rowptr || nrcmds < 1 || nrcmds > 128) {
    set_error("SLIMGPU_Predict: bad arguments (1 <= nrcmds <= 128)");
    return SLIM_ERROR_INPUT;
  }
  try {
    (void)hipGetLastError();
    int ndev = 0;
    TOPN_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw HipFail{hipErrorNoDevice, "hipGetDeviceCount"};
    const int64_t wnnz = W->rowptr[W->nrows];
    DevBuf<int64_t> d_wptr((size_t)W->nrows + 1);
    DevBuf<int32_t> d_wind((size_t)wnnz);
    DevBuf<float> d_wval((size_t)wnnz);
    TOPN_TRY(hipMemcpy(d_wptr.p, W->rowptr, sizeof(int64_t) * ((size_t)W->nrows + 1), hipMemcpyHostToDevice));
    if (wnnz) {
      TOPN_TRY(hipMemcpy(d_wind.p, W->rowind, sizeof(int32_t) * (size_t)wnnz, hipMemcpyHostToDevice));
      TOPN_TRY(hipMemcpy(d_wval.p, W->rowval, sizeof(float) * (size_t)wnnz, hipMemcpyHostToDevice));
    }
    DeviceRowView v;
    v.nrows = W->nrows;
    v.ncols = W->ncols;
    v.nnz = wnnz;
    v.d_ptr = d_wptr.p;
    v.d_ind = d_wind.p;
    v.d_val = d_wval.p;
    for (int32_t r = 0; r < W->nrows; ++r) v.max_row = std::max<int64_t>(v.max_row, W->rowptr[r + 1] - W->rowptr[r]);
    return predict_device_view(v, hist, nrcmds, output, scores, counts);
  } catch (const HipFail& e) {
    set_error(std::string("SLIMGPU_Predict: HIP error '") + hipGetErrorString(e.code) + "' in " + e.where);
    return e.code == hipErrorOutOfMemory ? SLIM_ERROR_MEMORY : SLIM_ERROR;
  }
}

}  // namespace slimamd
