// cd_wave.hpp -- the SLIM coordinate-descent solver as HIP for gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns one item column iC at a time and runs the whole
// per-item pipeline of the reference's EstimateModelCD loop body
// (src/libslim/estimate.c:402-530) + CoordinateDescent (src/libslim/cd.c:101-142):
//
//   1. y  <- column iC (scatter into the residual r = y - yhat)   estimate.c:406-408
//   2. aTy <- R^T y through the users of column iC ("Gram column":
//        sum over u in col iC of val * row_u); the reference scans the whole
//        matrix per item instead (estimate.c:412-421), same vector
//   3. active list {i != iC : aTy_i > l1} by ballot compaction, ascending ids
//                                                                 estimate.c:433-444
//   4. warm start from the previous model's column iC            estimate.c:453-471
//   5. CD sweeps, visiting the active list through the keyed permutation of
//      cd_perm.hpp; one fused pass per visit:
//        dot = a_i . r   (lane-strided column walk + DPP wave reduction)
//        num = dot + x_i * |a_i|^2     ( == aTy_i - a_i.(yhat - x_i a_i), cd.c:121-123 )
//        x_i' = num > l1 ? (num - l1) / (cnorm_i^2 + l2) : 0      cd.c:124-127
//        r -= (x_i' - x_i) a_i   only when the coefficient changed  cd.c:27,121,128
//      stop when sum (x_i'-x_i)^2 < optTol                         cd.c:135
//   6. 1/2||r||^2, objective, compaction of |x| > 1e-7 into the output arena
//                                                                 estimate.c:477-505
//
// Work vectors (r over users, aTy/ids and x over items) live in LDS when they
// fit (USE_LDS) and in a per-wave HBM slab otherwise.  Wavefronts are
// persistent and pull item columns from an atomic queue ordered by descending
// cost (the device form of `omp for schedule(dynamic,32)`, estimate.c:402).
// Arithmetic is fp32 (matrix values, residual, dot accumulators); no MFMA --
// this is sparse dot/axpy, bound by gather latency and HBM/L2 bandwidth.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cd_perm.hpp"

namespace slimamd {

struct DevMatrix {
  int32_t nrows, ncols;
  int64_t nnz;
  const int64_t* rowptr;  // CSR
  const int32_t* rowind;
  const float* rowval;    // nullptr: binary matrix
  const int64_t* colptr;  // CSC, user ids ascending inside each column
  const int32_t* colind;
  const float* colval;    // nullptr iff rowval is
  const float* cnorm;     // (float)sqrt(fp32 sum of squares)     setup.c:130
  const float* csq;       // fp32 sum of squares of the column
};

struct SolveArgs {
  float l1, l2, opt_tol;
  int32_t maxniters;
  uint32_t seed;
  int32_t nnbrs;    // > 0: FSLIM, the active set is the nnbrs most similar columns
  int32_t simtype;  // 0 cos, 1 jac, 2 dotp (slim.h:196-200)
  // work list (item ids, most expensive first) and the queue head
  const int32_t* order;
  int32_t nwork;
  int32_t* queue;
  // warm start: column view of the previous model (nullptr: cold start)
  const int64_t* icolptr;
  const int32_t* icolind;
  const float* icolval;
  int32_t incols;
  // per-wave HBM slab for the work vectors (HBM kernel only); for the tile kernel
  // (cd_tile.hpp) slab = interleaved residuals, xslab = interleaved x, ulist = union list
  float* slab;
  int64_t slab_stride;  // floats per wavefront / workgroup
  int32_t nrows_pad, ncols_pad;
  float* xslab;
  int64_t x_stride;
  int32_t* ulist;
  int64_t u_stride;
  int32_t ngroups;      // tiles of 16 item columns in the work list
  uint64_t* trace;      // optional per-tile timeline (SLIM_GPU_TRACE), 8 words per tile
  // tile clusters (cd_tile.hpp): K workgroups share a tile, users split in K ranges
  int32_t cluster;               // K
  const int32_t* ubounds;        // [K+1] user-range boundaries
  const int64_t* csplit;         // [ncols][K+1] column slice boundaries (K = 1: colptr pairs)
  unsigned long long* mailbox;   // per cluster: 2 x 8 x P granules (+8), zeroed per launch
  float* atypart;                // per workgroup (stride x_stride): [ncols][P] partial aTy of
                                 // the screen pass over the member's users
  int32_t exact_gram;            // 1: no float atomics in the aTy sums (ratings are not small
                                 // integers, where any order gives the same float)
  int32_t bm_shift, bm_words;    // LDS user bitmap: one bit per 1 << bm_shift users, bm_words words
  // heavy-tile phase: the first nheavy tiles of the work list (the most expensive ones) are
  // solved by clusters of cluster_hi workgroups before the launch regroups into clusters
  // of `cluster` (cluster divides cluster_hi, so the small clusters nest in the big ones)
  int32_t nheavy;                // 0: no heavy phase
  int32_t cluster_hi;
  const int32_t* ubounds_hi;
  const int64_t* csplit_hi;
  unsigned long long* mailbox_hi;
  int32_t* queue_hi;
  int64_t nnz_last;              // nnz - 1 (0 for an empty matrix): clamp for unconditional loads
  int32_t hi_prefetch;           // heavy phase: request the next visit's ids early
  // shards: this launch solves granules (32 work-list entries) shard_index, shard_index +
  // shard_count, ... of a larger cost-ordered list; the visiting permutation is keyed by the
  // tile's position in THAT list, so a column's result does not depend on the shard count
  int32_t shard_count, shard_index;
  // tile kernel: 1 = undo the round-robin block -> XCD placement so that the members of a
  // cluster share an XCD (requires gridDim.x % 8 == 0); see tile_block_id()
  int32_t xcd_swizzle;
  // tile kernel: screen sums a_i . y of every tile, [tile][ncols][P], kept across solves of the
  // same columns (0: unused, 1: record, 2: read instead of running the screen pass)
  // (3: build rows of G = R^T R instead of solving: the sums a_i . y of tile item q are row
  // item_q of G -- cd_gram.hpp)
  float* gram;
  int32_t gram_mode;
  // item-space CD (cd_gram.hpp): G = R^T R, row-major with row stride G_ld (>= ncols_pad, the
  // padding holds 0); ulist = [tile][ncols_pad] union lists, tile_nunion their lengths
  float* G;
  int64_t G_ld;
  int32_t* tile_nunion;
  const int32_t* gram_pos;       // gram_mode 3: position of every column in the work list
  int32_t gram_bits;             // gram_mode 3, binary matrix: y packed one word per user in LDS
  int32_t gram_split_stride;     // gram_bits: entries per column of the slice table the member reads (K + 1; with
                                 // user passes, the passes' common table: the launch's ubounds / csplit point at
                                 // the pass's first range)
  int32_t gram_accum;            // gram_bits, user passes after the first: the sums are ADDED to G
  // packed item-space kernel (cd_gramr.hpp): g of every problem as the solve leaves it (g_save) / as an
  // earlier solve of the same problems left it (g_load: starts from it instead of set-up row + fold);
  // [item][g_stride] floats in the kernel's own thread layout; both may point at the same buffer
  float* g_save;
  const float* g_load;
  int64_t g_stride;
  // output arena: column iC's kept entries land at [out_off[iC], +out_cnt[iC])
  int32_t* out_cnt;
  int64_t* out_off;
  int32_t* out_ind;
  float* out_val;
  unsigned long long* out_cursor;
  int64_t out_cap;
  int32_t* overflow;
  // per-column counters
  int32_t* st_na;
  int32_t* st_sweeps;
  int32_t* st_conv;
  int64_t* st_G;
  int64_t* st_D;
  int64_t* st_U;
  int64_t* st_B;  // item-space kernels: bytes of G streamed by the column's updates and folds
  float* st_err;
  float* st_obj;
};

constexpr float kEps = 1e-7f;  // def.h:14

// ---- wave-level primitives ---------------------------------------------------

// The update rule's arithmetic in the item-space kernels (cd_gram.hpp, cd_gramr.hpp, cd_gramrp.hpp),
// spelled out: left to -ffp-contract, `g + x * sq` came out as a multiply and an add in one kernel
// (the product hoisted out of a loop) and as one fma in another -- models an ulp apart, where the
// tests ask for equal bits.  The product is rounded, then the sum; the denominator is one fma.
// (__fmul_rn / __fadd_rn are ordinary operations to the optimizer, which fused them all the same:
// the product passes through an empty asm)
__device__ __forceinline__ float cd_num(float g, float xeff, float sq) {
  float p = xeff * sq;
  asm volatile("" : "+v"(p));
  return g + p;
}
__device__ __forceinline__ float cd_den(float cn, float l2) { return fmaf(cn, cn, l2); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float uni(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int64_t uni(int64_t v) {
  uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int lane_bcast(int v, int src) {
  return __builtin_amdgcn_readlane(v, src);
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int64_t lane_bcast(int64_t v, int src) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes; the result is wave-uniform.  Four DPP butterfly steps
// inside each 16-lane row (quad xor 1, quad xor 2, half-row mirror, row
// mirror), then the four row totals are combined through the scalar unit.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}

// Lanes of one wavefront exchange data through r/aty/x.  LDS requests of a wave
// are served in issue order, so the LDS form only has to stop the compiler
// from reordering; the HBM form needs the stores of every lane visible to the
// whole wave (shared vector L1 of the CU) before the next loads.
template <bool USE_LDS>
__device__ __forceinline__ void wave_sync() {
  if (USE_LDS) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- the kernel ---------------------------------------------------------------

template <bool USE_LDS, bool HAS_VAL>
__global__ __launch_bounds__(64) void cd_wave_kernel(const DevMatrix A, const SolveArgs S) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const uint64_t lane_lt = (1ull << lane) - 1ull;

  float* r;      // residual y - yhat over users
  float* aty;    // dense aTy over items, then reused as the active id list
  float* x;      // coefficients of the active list (slot order = ascending id)
  if (USE_LDS) {
    r = reinterpret_cast<float*>(smem);
  } else {
    r = S.slab + (int64_t)blockIdx.x * S.slab_stride;
  }
  aty = r + S.nrows_pad;
  x = aty + S.ncols_pad;
  int* ids = reinterpret_cast<int*>(aty);

  const int nrows = A.nrows, ncols = A.ncols;
  const float l1 = S.l1, l2 = S.l2;

  for (;;) {
    int w = 0;
    if (lane == 0) w = atomicAdd(S.queue, 1);
    w = uni(w);
    if (w >= S.nwork) break;
    const int iC = uni(S.order[w]);
    const int64_t cs = uni(A.colptr[iC]), ce = uni(A.colptr[iC + 1]);

    // -- clear the work vectors
    for (int u = lane; u < nrows; u += 64) r[u] = 0.0f;
    for (int i = lane; i < ncols; i += 64) aty[i] = 0.0f;
    wave_sync<USE_LDS>();

    // -- 1+2: y scatter and aTy.
    int64_t G = 0;
    if (S.exact_gram) {
      // Ratings that are not small integers: float atomics would make the sums (and with them
      // the strict screen aTy > l1) depend on the arrival order.  Form them as the reference
      // does (estimate.c:412-421): a_i . y for every column i, one lane per column, users
      // ascending -- a fixed order.  Costs one pass over the column view per item.
      for (int64_t j = cs + lane; j < ce; j += 64) {
        const int u = A.colind[j];
        r[u] = HAS_VAL ? A.colval[j] : 1.0f;
        G += A.rowptr[u + 1] - A.rowptr[u];
      }
      for (int off = 32; off > 0; off >>= 1) G += __shfl_xor(G, off);
      wave_sync<USE_LDS>();
      for (int ib = 0; ib < ncols; ib += 64) {
        const int i = ib + lane;
        float acc = 0.0f;
        if (i < ncols) {
          const int64_t s = A.colptr[i], e = A.colptr[i + 1];
          bool co = false;  // a user of iC rated i too (their products may still cancel)
          for (int64_t k = s; k < e; ++k) {
            const float yv = r[A.colind[k]];
            acc += (HAS_VAL ? A.colval[k] : 1.0f) * yv;
            co = co || yv != 0.0f;
          }
          // FSLIM's candidates are the CO-RATED items, whatever their sum (neighbors.c:46-60):
          // a sum that cancelled to 0 is kept apart from "no common user" by its sign bit
          aty[i] = (acc == 0.0f && co) ? -0.0f : acc;
        }
      }
    } else {
    //    Gram column: 64 users of the column per step;
    //    each user's row is then spread over the lanes (ids inside a row are
    //    distinct, rows can collide -> float atomics, order-free for the
    //    integer-valued ratings of every shipped dataset)
    for (int64_t jb = cs; jb < ce; jb += 64) {
      const int64_t j = jb + lane;
      const bool ok = j < ce;
      const int u_l = ok ? A.colind[j] : 0;
      const float v_l = ok ? (HAS_VAL ? A.colval[j] : 1.0f) : 0.0f;
      const int64_t rs_l = ok ? A.rowptr[u_l] : 0;
      const int64_t re_l = ok ? A.rowptr[u_l + 1] : 0;
      if (ok) r[u_l] = v_l;
      const int cnt = (int)((ce - jb) < 64 ? (ce - jb) : 64);
      for (int k = 0; k < cnt; ++k) {
        const int64_t rs = lane_bcast(rs_l, k), re = lane_bcast(re_l, k);
        const float v = lane_bcast(v_l, k);
        G += re - rs;
        for (int64_t e = rs + lane; e < re; e += 64) {
          const float rv = HAS_VAL ? A.rowval[e] : 1.0f;
          atomicAdd(&aty[A.rowind[e]], v * rv);
        }
      }
    }
    }
    wave_sync<USE_LDS>();

    // -- 3: active list
    int na = 0;
    if (S.nnbrs > 0) {
      // FSLIM (estimate.c:424-431, neighbors.c:16-125): the nnbrs columns most similar to
      // iC among those sharing a user with it; no l1 screen.  The co-rating dot products
      // are the Gram column just computed.  Ties (undefined upstream): lower item id first.
      const float ninf = -__builtin_huge_valf();
      const float cn_c = A.cnorm[iC];
      for (int i = lane; i < ncols; i += 64) {
        const float a = aty[i];
        float sim = ninf;
        if ((a != 0.0f || __builtin_signbit(a)) && i != iC) {
          const float cn_i = A.cnorm[i];
          sim = S.simtype == 0 ? a / cn_i : (S.simtype == 1 ? a / ((cn_i + cn_c) - a) : a);
        }
        aty[i] = sim;
      }
      wave_sync<USE_LDS>();
      int* sel = reinterpret_cast<int*>(x);  // selected ids, in similarity order
      for (int round = 0; round < S.nnbrs; ++round) {
        float bs = ninf;
        int bi = 0x7fffffff;
        for (int i = lane; i < ncols; i += 64) {  // ascending i per lane: first max wins
          const float sv = aty[i];
          if (sv > bs) {
            bs = sv;
            bi = i;
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float os = __shfl_xor(bs, off);
          const int oi = __shfl_xor(bi, off);
          const bool take = os > bs || (os == bs && oi < bi);
          bs = take ? os : bs;
          bi = take ? oi : bi;
        }
        bs = uni(bs);
        bi = uni(bi);
        if (!(bs > ninf)) break;  // fewer candidates than nnbrs (wave-uniform)
        if (lane == 0) {
          sel[na] = bi;
          aty[bi] = ninf;
        }
        ++na;
        wave_sync<USE_LDS>();
      }
      // the active list is kept in ascending id order (output order, warm-start search)
      int myid = 0, rank = 0;
      for (int b = 0; b < na; b += 64) {  // na <= nnbrs entries, 64 per pass
        const int e = b + lane;
        myid = e < na ? sel[e] : 0;
        rank = 0;
        for (int o = 0; o < na; ++o) rank += (sel[o] < myid) ? 1 : 0;
        wave_sync<USE_LDS>();
        if (e < na) ids[rank] = myid;  // ids aliases aty: similarities are dead by now
      }
      wave_sync<USE_LDS>();
      for (int e = lane; e < na; e += 64) x[e] = 0.0f;
    } else
    for (int base = 0; base < ncols; base += 64) {
      const int i = base + lane;
      const float a = i < ncols ? aty[i] : 0.0f;
      const bool act = (i < ncols) && (i != iC) && (a > l1);
      const uint64_t m = __ballot(act);
      wave_sync<USE_LDS>();  // every lane has consumed its aty[i] before slots are rewritten
      if (act) {
        const int slot = na + __popcll(m & lane_lt);
        ids[slot] = i;
        x[slot] = 0.0f;
      }
      na += __popcll(m);
    }
    wave_sync<USE_LDS>();

    // -- 4: warm start.  imodel column iC is ascending; binary-search each of
    //    its ids in the active list, then fold the non-zero x into r
    if (S.nnbrs == 0 && S.icolptr != nullptr && iC < S.incols) {
      const int64_t ws = uni(S.icolptr[iC]), we = uni(S.icolptr[iC + 1]);
      for (int64_t e = ws + lane; e < we; e += 64) {
        const int k = S.icolind[e];
        int lo = 0, hi = na;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (ids[mid] < k) lo = mid + 1; else hi = mid;
        }
        // (estimate.c:456-464: a negative previous value is copied and then reset to 0)
        if (lo < na && ids[lo] == k) x[lo] = fmaxf(S.icolval[e], 0.0f);
      }
      wave_sync<USE_LDS>();
      for (int q = 0; q < na; ++q) {  // cd.c:108-110
        const float xi = uni(x[q]);
        if (xi > kEps || xi < -kEps) {
          const int i = uni(ids[q]);
          const int64_t s = uni(A.colptr[i]), t = uni(A.colptr[i + 1]);
          for (int64_t e = s + lane; e < t; e += 64)
            r[A.colind[e]] -= xi * (HAS_VAL ? A.colval[e] : 1.0f);
          wave_sync<USE_LDS>();
        }
      }
    }

    // -- 5: sweeps
    const int64_t cap = 50 * (ce - cs);  // estimate.c:448-449
    const int maxit = cap < (int64_t)S.maxniters ? (int)cap : S.maxniters;
    int t = 0, conv = 0;
    int64_t D = 0, U = 0;
    for (; t < maxit; ++t) {
      float dlt = 0.0f;
      const PermCtx pc = perm_make((uint32_t)na, perm_key(S.seed, (uint32_t)iC, (uint32_t)t));
      // Software pipeline over the visits of the sweep (every stage is a dependent
      // memory access of the one before it):
      //   stage A, 3 visits ahead: slot q = perm(p+3), item id and x from the work vectors
      //   stage B, 2 visits ahead: column offsets and norms of that item (L2)
      //   stage C, 1 visit  ahead: the first 128 entries of the column (two per lane)
      // Slots are visited once per sweep, so an x read ahead cannot be stale; the
      // pipeline is refilled at every sweep.
      struct StageB { int64_t s, e; float cn, sq; };
      auto stage_a = [&](const int pp, int& q, int& i, float& xv) {
        q = (int)perm_index(pc, (uint32_t)pp);
        i = ids[q];
        xv = x[q];
      };
      auto stage_b = [&](const int i) -> StageB {
        StageB b;
        b.s = A.colptr[i];
        b.e = A.colptr[i + 1];
        b.cn = A.cnorm[i];
        b.sq = A.csq[i];
        return b;
      };
      int qa = 0, ia = 0, qb = 0, ib = 0, qc = 0, ic = 0;
      float xa = 0.f, xb = 0.f, xc = 0.f;
      StageB ba = {0, 0, 0.f, 0.f}, bb = {0, 0, 0.f, 0.f};
      int u0 = 0, u1 = 0;
      float v0 = 0.f, v1 = 0.f;
      auto stage_c = [&](const int64_t s, const int64_t e_end) {
        const int64_t k0 = s + lane, k1 = k0 + 64;
        u0 = k0 < e_end ? A.colind[k0] : 0;
        v0 = k0 < e_end ? (HAS_VAL ? A.colval[k0] : 1.0f) : 0.0f;
        u1 = k1 < e_end ? A.colind[k1] : 0;
        v1 = k1 < e_end ? (HAS_VAL ? A.colval[k1] : 1.0f) : 0.0f;
      };
      if (na > 0) {  // fill: visit 0 fully staged, visit 1 through B, visit 2 through A
        stage_a(0, qa, ia, xa);
        ba = stage_b(uni(ia));
        if (na > 1) {
          stage_a(1, qb, ib, xb);
          bb = stage_b(uni(ib));
        }
        if (na > 2) stage_a(2, qc, ic, xc);
        stage_c(uni(ba.s), uni(ba.e));
      }
      for (int p = 0; p < na; ++p) {
        // the visit being processed
        const int q = uni(qa);
        const float xi = uni(xa);
        const int64_t s = uni(ba.s), e_end = uni(ba.e);
        const float cn = uni(ba.cn), sq = uni(ba.sq);
        const int cu0 = u0, cu1 = u1;
        const float cv0 = v0, cv1 = v1;
        // advance the pipeline before touching r, so the loads fly under this visit
        qa = qb; ia = ib; xa = xb; ba = bb;
        qb = qc; ib = ic; xb = xc;
        if (p + 1 < na) stage_c(uni(ba.s), uni(ba.e));
        if (p + 2 < na) bb = stage_b(uni(ib));
        if (p + 3 < na) stage_a(p + 3, qc, ic, xc);

        float acc = cv0 * r[cu0] + cv1 * r[cu1];
        for (int64_t e = s + 128 + lane; e < e_end; e += 64) {
          const float v = HAS_VAL ? A.colval[e] : 1.0f;
          acc += v * r[A.colind[e]];
        }
        const float dot = wave_sum(acc);

        const float xeff = (xi > kEps || xi < -kEps) ? xi : 0.0f;
        const float num = dot + xeff * sq;
        const float nx = num > l1 ? (num - l1) / (cn * cn + l2) : 0.0f;
        const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
        const float d = neff - xeff;
        D += e_end - s;
        if (d != 0.0f) {
          if (s + lane < e_end) r[cu0] -= d * cv0;
          if (s + 64 + lane < e_end) r[cu1] -= d * cv1;
          for (int64_t e = s + 128 + lane; e < e_end; e += 64)
            r[A.colind[e]] -= d * (HAS_VAL ? A.colval[e] : 1.0f);
          U += e_end - s;
          wave_sync<USE_LDS>();
        }
        if (lane == 0) x[q] = nx;
        dlt += (nx - xi) * (nx - xi);
      }
      wave_sync<USE_LDS>();
      if (dlt < S.opt_tol) {  // cd.c:135-138
        conv = 1;
        break;
      }
    }
    const int niters = t + 1;  // cd.c:140

    // -- 6: loss terms and output
    float e2 = 0.0f;
    for (int u = lane; u < nrows; u += 64) {
      const float rv = r[u];
      e2 += rv * rv;
    }
    float reg = 0.0f;
    int nz = 0;
    for (int base = 0; base < na; base += 64) {
      const int q = base + lane;
      const float xv = q < na ? x[q] : 0.0f;
      reg += 0.5f * l2 * xv * xv + l1 * fabsf(xv);
      nz += __popcll(__ballot(fabsf(xv) > kEps));
    }
    const float err = 0.5f * wave_sum(e2);
    const float obj = err + wave_sum(reg);

    unsigned long long off = 0;
    if (lane == 0) off = atomicAdd(S.out_cursor, (unsigned long long)nz);
    off = (unsigned long long)uni((int64_t)off);
    const bool fits = (int64_t)(off + (unsigned long long)nz) <= S.out_cap;
    if (fits) {
      int wpos = 0;
      for (int base = 0; base < na; base += 64) {
        const int q = base + lane;
        const float xv = q < na ? x[q] : 0.0f;
        const bool keep = fabsf(xv) > kEps;
        const uint64_t m = __ballot(keep);
        if (keep) {
          const int64_t dst = (int64_t)off + wpos + __popcll(m & lane_lt);
          S.out_ind[dst] = ids[q];
          S.out_val[dst] = xv;
        }
        wpos += __popcll(m);
      }
    }
    if (lane == 0) {
      if (!fits) atomicMax(S.overflow, 1);
      S.out_cnt[iC] = fits ? nz : -nz - 1;
      S.out_off[iC] = (int64_t)off;
      S.st_na[iC] = na;
      S.st_sweeps[iC] = niters;
      S.st_conv[iC] = conv;
      S.st_G[iC] = G;
      S.st_D[iC] = D;
      S.st_U[iC] = U;
      S.st_err[iC] = err;
      S.st_obj[iC] = obj;
    }
    wave_sync<USE_LDS>();
  }
}

}  // namespace slimamd
