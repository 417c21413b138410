// cd_tile.hpp -- the CD solver for matrices whose residual does not fit in LDS.
//
// Why a second kernel.  With the residual r (one float per user) in HBM, the
// one-wavefront-per-item kernel of cd_wave.hpp gathers 4 bytes from a random
// sector per nnz touched and re-streams the whole column view per item: at
// 1M users x 100K items it moves ~16x more bytes than it uses.  Here a
// workgroup of 16 wavefronts solves a TILE of P item columns together
// (P = 32 by default, 16 optional):
//
//   * residuals are interleaved, r[user][P].  Measured on MI355X
//     (scripts/micro/gather_bw.hip): random 64-byte granules saturate the chip
//     at 3.2 TB/s, random 128-byte granules at 6.0 TB/s -- the fabric is bound
//     by request count below a full 128-byte line.  With P = 32 the 32
//     problems' values of one user ARE one 128-byte line, so every gather and
//     every write-back is a whole line;
//   * the P problems visit coordinates in the same order, so each column of R
//     (ids + values) is read once per tile instead of once per item, with
//     coalesced loads: wavefront w of the workgroup owns 64 consecutive nnz of
//     each 1024-nnz chunk of the visited column.  The 16-lane rows of a lane group
//     (P lanes = the P problems of one user) load the same 16 entries, and step j
//     of group g reads entry g * P + j with one DPP row broadcast (no LDS staging,
//     no permute, no address register per step);
//   * a wavefront step covers 64/P users x P problems; all steps of a block
//     (P loads per lane) are in flight together, and a slice of several chunks is one
//     stream: two blocks alternate, the ids of chunk k+2 are requested before the
//     gathers of chunk k+1 (loads complete in issue order), and no load of the
//     stream is conditional -- entries past the end of a slice point at a spare
//     line behind the member's user range that holds 0.  The P dot products are
//     reduced with log2(64/P) cross-lane adds + one LDS exchange per visit; every
//     wavefront then evaluates the P soft-threshold updates redundantly (bitwise
//     identical, so control flow stays workgroup-uniform) and applies the residual
//     update to its own nnz, storing whole lines (all P problems of a user,
//     changed or not -- no read-modify-write in L2/HBM);
//   * the last TWO chunks of a slice stay in registers between dot and update
//     (store-only update); earlier chunks of longer slices are read again;
//   * the next visit's scalars (column id, offsets, x row, norms) are loaded one
//     visit ahead with scalar-base addressing; per-problem counters live in LDS;
//   * the screen aTy > l1 (estimate.c:412-444) is one extra pass a_i . y over every
//     column for the P problems at once, each wavefront taking whole columns and
//     skipping the users outside the tile's user set through an LDS bitmap -- no
//     atomics, a fixed summation order (round 1 accumulated the Gram column with
//     device-scope float atomics: twice the time, and order-dependent sums).  The
//     sums depend on R only: they are kept in HBM and read by the next solve of the
//     same columns (S.gram_mode; model-selection grids);
//   * a warm start's coefficients are folded into the residual row by row (FOLD = 2:
//     r[u] = y[u] - sum_j v_uj x_j over the member's users, x lines from ONE copy per
//     cluster) instead of a gather + write-back per column of the union.
//
// Per problem the arithmetic is exactly that of cd_wave.hpp (and of the
// reference, src/libslim/cd.c:101-142): same update rule, same epsilon rule,
// same stopping test, per-problem sweep cap min(50*nnz, maxniters); only the
// visiting order differs -- a keyed permutation (cd_perm.hpp) of the UNION of
// the tile's active sets, each problem skipping coordinates outside its own
// active set.
//
// x is kept dense and interleaved too, x[item][P], with -inf marking "not in
// this problem's active set".
//
// Clusters.  A tile's critical path is (sweeps of its slowest problem) x (time of
// one sweep on one CU); the most popular items need ~4x the median sweeps, so with
// few tiles per CU a launch waits for one workgroup.  K = 1, 2, 4, 8 or 16 workgroups
// (a cluster) can therefore share a tile: the USERS are split into K ranges of equal
// nnz, member k keeps only its range of r and walks only its slice of every column
// (csplit[i][k] .. csplit[i][k+1], precomputed), and the P partial dots are summed
// over the cluster once per visit with a tagged-granule exchange through HBM
// (8-byte {epoch, value} words written by one write-through store each and polled
// until every tag shows the visit's epoch -- placement-independent, no fences; see
// MI355X guide "R2").  Every member then evaluates the same updates on its own
// copy of x, so members stay in lock-step without further communication.  All
// spins are bounded: a member that waits longer than ~10 s raises the abort flag,
// the launch ends, and the host solves what is pending again without clusters.
#pragma once
#include <type_traits>

#include "cd_wave.hpp"

namespace slimamd {

// The chunks of a column slice are aligned to its END: the first chunk is the short one, the last
// two -- the ones that stay in registers for the update -- are full.  (Aligned to the start, a
// slice of 2.5 chunks kept 1.5 chunks and gathered 1 again on an updating visit; now it keeps 2
// and gathers 0.5 again.)  0 restores the start-aligned grid (A/B builds).
#ifndef SLIM_TILE_END_ALIGNED
#define SLIM_TILE_END_ALIGNED 1
#endif

constexpr int kTileNW = 16;  // wavefronts per workgroup (default geometry)
constexpr int kTileKMax = 32;  // largest cluster (workgroups sharing one tile)
constexpr float kInactive = -__builtin_huge_valf();
__device__ __forceinline__ bool tile_active(float xv) { return xv > -3.0e38f; }

__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(int& v) { asm volatile("" : "+v"(v)); }

// Broadcast inside each row of 16 lanes: every lane reads lane N of its own row (DPP
// row_newbcast, one VALU move, no LDS round trip and no address register).  The visit
// distributes the user ids / values of a 64-nnz block with it: the block is loaded so that the
// 16-lane rows of a lane group (P lanes = the P problems of one user) hold the same 16 entries,
// P / 16 registers per block, and step j of group g reads entry g * P + j.
template <int N>
__device__ __forceinline__ int row_bcast_c(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ int row_bcast(int v, int n) {  // n: a constant after unrolling
  switch (n & 15) {
    case 0: return row_bcast_c<0>(v);
    case 1: return row_bcast_c<1>(v);
    case 2: return row_bcast_c<2>(v);
    case 3: return row_bcast_c<3>(v);
    case 4: return row_bcast_c<4>(v);
    case 5: return row_bcast_c<5>(v);
    case 6: return row_bcast_c<6>(v);
    case 7: return row_bcast_c<7>(v);
    case 8: return row_bcast_c<8>(v);
    case 9: return row_bcast_c<9>(v);
    case 10: return row_bcast_c<10>(v);
    case 11: return row_bcast_c<11>(v);
    case 12: return row_bcast_c<12>(v);
    case 13: return row_bcast_c<13>(v);
    case 14: return row_bcast_c<14>(v);
    default: return row_bcast_c<15>(v);
  }
}
__device__ __forceinline__ float row_bcast(float v, int n) {
  return __int_as_float(row_bcast(__float_as_int(v), n));
}

// load from a uniform base + a 32-bit byte offset of the lane (global_load ..., v_off, s[base])
template <class T>
__device__ __forceinline__ T ld_off(const void* base, const uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

typedef unsigned long long tile_gran_t;

__device__ __forceinline__ void gran_store(tile_gran_t* p, uint32_t epoch, float v) {
  __hip_atomic_store(p, ((tile_gran_t)epoch << 32) | (tile_gran_t)__float_as_uint(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ tile_gran_t gran_load(const tile_gran_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One phase of a launch.  HI = the heavy phase: clusters of S.cluster_hi workgroups on the
// first S.nheavy tiles of the work list; otherwise clusters of S.cluster on the rest.  The
// phase is a template parameter so that the cluster geometry stays a function of kernel
// arguments (re-derivable, no live registers across the visit loop).  Returns false when
// the launch was aborted.
//
// FOLD selects how warm-start coefficients enter the residual (cd.c:108-110), 0: not at all (a
// kernel for cold starts only), 1: column by column (one pass over every column of the union
// list: gather + write-back of its users' lines), 2: row by row -- r[u] = y[u] - sum over the
// items j of row u of v_uj x[j], the x lines gathered from ONE copy per cluster (member 0's),
// which the XCD's L2 can hold, and every residual line written exactly once.
//
// bid = the workgroup's position in the launch.  With S.xcd_swizzle the hardware's round-robin
// placement (block b on XCD b % 8, observed) is undone so that consecutive positions -- the
// members of a cluster -- share an XCD and its L2; nothing depends on it for correctness.
__device__ __forceinline__ int tile_block_id(const SolveArgs& S) {
  const int b = (int)blockIdx.x, g = (int)gridDim.x;
  return S.xcd_swizzle ? (b & 7) * (g >> 3) + (b >> 3) : b;
}

template <int P, bool HAS_VAL, bool PROFILE, int NW, bool FSLIM, bool HI, int FOLD>
__device__ __forceinline__ bool tile_phase(const DevMatrix& A, const SolveArgs& S, uint32_t& epoch) {
  constexpr int NT = 64 * NW;  // threads per workgroup
  constexpr int SL = 64 / P;       // users per wavefront step (lane groups)
  constexpr int STEPS = 64 / SL;   // steps per 64-nnz block (== P)
  constexpr int PPW = P / NW;      // problems served per wavefront in the per-problem phases
  constexpr int LOGP = P == 32 ? 5 : 4;
  static_assert(P == 16 || P == 32, "tile width");
  __shared__ float s_part[2][NW][P];
  __shared__ float s_red[2][NW][P];
  __shared__ int s_item[P];
  __shared__ int s_na[P];
  __shared__ int s_maxit[P], s_niters[P], s_conv[P];
  __shared__ unsigned long long s_D[P], s_U[P];
  __shared__ int s_grp, s_nunion;
  __shared__ float s_tot[2][P];
  __shared__ int s_abort;
  extern __shared__ uint32_t s_bits[];  // user bitmap of the screen pass (S.bm_words words)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int q = lane & (P - 1);  // problem handled by this lane
  const int slot = lane >> LOGP; // which user of a wavefront step
  const uint64_t lane_lt = (1ull << lane) - 1ull;

  // cluster geometry: K consecutive workgroups share tiles, member mk owns users
  // [ubase, uend)
  const int K = HI ? S.cluster_hi : S.cluster;
  const int bid = tile_block_id(S);
  const int cid = bid / K, mk = bid % K;
  const int32_t* __restrict__ ubounds = HI ? S.ubounds_hi : S.ubounds;
  const int ubase = ubounds[mk], uend = ubounds[mk + 1];
  tile_gran_t* const mbox = (HI ? S.mailbox_hi : S.mailbox) + (int64_t)cid * (2 * kTileKMax * P + 8);
  // this member's partial aTy of the tile's columns over ITS users, [ncols][P] (screen pass)
  float* const part = S.atypart + (int64_t)bid * S.x_stride;
  const int64_t* __restrict__ csplit = HI ? S.csplit_hi : S.csplit;  // [ncols][K+1] slice boundaries
  const int grp_end = HI ? S.nheavy : S.ngroups;
  if (tid == 0) s_abort = 0;
  float* __restrict__ r = S.slab + (int64_t)bid * S.slab_stride;   // [my users][P]
  float* x = S.xslab + (int64_t)bid * S.x_stride;                 // [ncols][P]
  int* __restrict__ ul = S.ulist + (int64_t)bid * S.u_stride;      // union list
  const int64_t* __restrict__ colptr = A.colptr;
  const int32_t* __restrict__ ci = A.colind;
  const float* __restrict__ cv = A.colval;

  const int ncols = A.ncols;
  const float l1 = S.l1, l2 = S.l2;

  // sum over the cluster of a per-problem value (identical in every lane serving q);
  // called by all threads of all members at the same points of the program.  Wave 0
  // publishes this member's P granules, then every lane polls the granules of "its"
  // members (all loads of a poll round are issued together: one round trip when the
  // partners have already published) until every tag shows the epoch.  The wait is
  // bounded: after ~10 s the abort flag is raised and the launch ends with an error.
  auto cluster_sum = [&](const float v) -> float {
    if (K == 1) return v;
    ++epoch;
    if (epoch == 0) epoch = 1;
    const int par = (int)(epoch & 1u);
    if (wave == 0) {
      if (lane < P) gran_store(mbox + (par * kTileKMax + mk) * P + lane, epoch, v);
      // every lane polls the granules of members slot, slot+SL, ...: four per round trip
      // (one batch covers clusters of up to 4*SL members, larger ones take two batches)
      constexpr int MAXG = 4;
      float tot = 0.0f;
      bool aborted = s_abort != 0;
      for (int kb = 0; kb < K; kb += MAXG * SL) {
        tile_gran_t g[MAXG];
        uint64_t t0 = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < MAXG; ++j) {
            const int kk = kb + slot + j * SL;
            g[j] = kk < K ? gran_load(mbox + (par * kTileKMax + kk) * P + q)
                          : ((tile_gran_t)epoch << 32);
          }
#pragma unroll
          for (int j = 0; j < MAXG; ++j) ok &= (uint32_t)(g[j] >> 32) == epoch;
          if (__all(ok) || aborted) break;
          if (t0 == 0) t0 = wall_clock64();
          // (a partner that is late by more than a few round trips is polled at a lower rate:
          // polls are fabric reads)
          if (wall_clock64() - t0 > 2000ull) __builtin_amdgcn_s_sleep(64);
          else __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > 1000000000ull) {  // 10 s at 100 MHz: give up, loudly
            aborted = true;
            s_abort = 1;
            atomicMax(S.overflow, 2);  // (max: an arena overflow, 1, never downgrades it)
          }
        }
#pragma unroll
        for (int j = 0; j < MAXG; ++j)
          if (kb + slot + j * SL < K) tot += __uint_as_float((uint32_t)g[j]);
      }
      if (SL == 4) tot += __shfl_xor(tot, 16);
      tot += __shfl_xor(tot, 32);
      if (lane < P) s_tot[par][q] = tot;
    }
    __syncthreads();
    return s_tot[par][q];
  };
  // cluster-wide barrier that also publishes this member's plain stores / atomics
  auto cluster_barrier = [&]() {
    if (K == 1) {
      __syncthreads();
      return;
    }
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    (void)cluster_sum(0.0f);
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  };

  for (;;) {
    // member 0 pulls the next tile and tells the others
    if (tid == 0) {
      int gnext = 0;
      if (mk == 0) gnext = HI ? atomicAdd(S.queue_hi, 1) : S.nheavy + atomicAdd(S.queue, 1);
      s_grp = gnext;
    }
    __syncthreads();
    const int grp = (int)(cluster_sum(mk == 0 ? (float)s_grp : 0.0f) + 0.5f);
    if (s_abort) return false;
    if (grp >= grp_end) break;
    const uint64_t t_start = wall_clock64();
    const int base = grp * P;
    // position of this tile in the unsharded work list (== grp for a single shard)
    constexpr int PER = 32 / P;  // tiles per 32-column shard granule
    const uint32_t gkey =
        (uint32_t)(((grp / PER) * S.shard_count + S.shard_index) * PER + grp % PER);
    const int nprob = (S.nwork - base) < P ? (S.nwork - base) : P;
    // Screen sums of this tile kept from an earlier solve of the same columns (a model-selection
    // grid solves every (l1, l2) pair over the same R: a_i . y does not depend on the pair):
    // S.gram_mode 2 = read them instead of running the screen pass, 1 = record them
    // (3 = the sums are the product: rows item_q of G = R^T R are written and the tile is done)
    float* const gram_t =
        (S.gram_mode == 1 || S.gram_mode == 2) ? S.gram + (int64_t)grp * S.x_stride : nullptr;
    const bool cached = S.gram_mode == 2;
    if (tid < P) {
      s_item[tid] = tid < nprob ? S.order[base + tid] : -1;
      s_na[tid] = 0;
    }
    // -- G = R^T R of a binary matrix (S.gram_mode 3 with S.gram_bits): the y of the tile's P items
    //    packed into ONE word per user of this member's range, in LDS (clusters of 32 on a
    //    1M-user matrix: 31K users, 125 KB) -- a column's dot products with all P items are then
    //    P ballots per 64 nnz over words read from LDS: no residual lines, no HBM gathers, only
    //    the column ids are streamed (the line-gathering screen pass below spent 10 s on the 1e9
    //    nnz of C4; this form is bound by the id stream)
    const bool gbits = !HAS_VAL && S.gram_mode == 3 && S.gram_bits != 0;
    if (gbits) {
      const int gst = S.gram_split_stride;  // (K + 1, or the row length of the user passes' table)
      const int nus = uend - ubase;
      for (int k = tid; k < nus; k += NT) s_bits[k] = 0u;
      __syncthreads();
#pragma unroll
      for (int pp = 0; pp < PPW; ++pp) {
        const int pq = wave + pp * NW;
        const int witem = s_item[pq];
        if (witem >= 0) {
          const int64_t cs = uni(csplit[(int64_t)witem * gst + mk]);
          const int64_t ce = uni(csplit[(int64_t)witem * gst + mk + 1]);
          for (int64_t j = cs + lane; j < ce; j += 64) atomicOr(&s_bits[ci[j] - ubase], 1u << pq);
        }
      }
      __syncthreads();
      if (S.gram_bits == 2) {
        // Lanes = COLUMNS (round 5).  The form below gives a wavefront one column at a time and
        // counts with P ballots per 64 nnz: ~2 wavefront instructions per nnz, and a slice is only
        // ~300 nnz long.  Here a wavefront takes 64 columns that are neighbours in the cost-ordered
        // work list (slices of about the same length), every lane walks ITS column's slice and adds
        // the users' words into a bit-sliced counter -- plane p holds bit p of the P running counts,
        // eight words enter through a carry-save tree (Harley-Seal), the carries ripple upward only
        // while some lane still has one -- ~6 lane operations per nnz, i.e. ~0.1 wavefront
        // instructions per nnz, and no cross-lane reduction at all: at the end every lane turns its
        // planes into the P counts of its column and writes one 128-byte line.  Positions at or
        // behind the tile's own only (symmetric fill).
        constexpr int NPL = 16;  // planes: counts below 2^16 per (column slice, item) -- a member's
                                 // user range holds at most 37 888 users (148 KB of LDS)
        const int32_t* __restrict__ ord = S.order;
        for (int p0 = base + wave * 64; p0 < S.nwork; p0 += NW * 64) {
          const int pl = p0 + lane;
          const bool have = pl < S.nwork;
          const int i = have ? ord[pl] : 0;
          int64_t cs = 0, ce = 0;
          if (have) {
            cs = csplit[(int64_t)i * gst + mk];
            ce = csplit[(int64_t)i * gst + mk + 1];
          }
          const int len = (int)(ce - cs);
          uint32_t ones = 0u, twos = 0u, fours = 0u, hi[NPL - 3];
#pragma unroll
          for (int pp = 0; pp < NPL - 3; ++pp) hi[pp] = 0u;
          int nhi = 0;  // planes of hi[] that may be non-zero in some lane (uniform)
#define SLIM_CSA(h, l, a, b, c)      \
  {                                  \
    const uint32_t u_ = (a) ^ (b);   \
    h = ((a) & (b)) | (u_ & (c));    \
    l = u_ ^ (c);                    \
  }
          // a step = 32 ids of every lane's slice, i.e. one whole 128-byte line per lane (eight
          // 16-byte loads, dword-aligned): a lane that took 8 ids per step came back to its line
          // four times, and with 1024 lanes per CU walking different lines the caches did not
          // keep them -- the HBM traffic was several times the column view
          struct __attribute__((packed, aligned(4))) ids4 { int32_t a, b, c, d; };
          for (int t = 0; __ballot(t < len) != 0ull; t += 32) {
            ids4 q4[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              // (a lane whose slice has ended keeps reading behind it, at most 31 entries beyond the
              // column view, which is allocated with that slack; what it reads is dropped below)
              const int64_t at = t < len ? cs + t + 4 * j : cs;
              q4[j] = *reinterpret_cast<const ids4*>(ci + at);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int uu[8] = {q4[2 * r].a, q4[2 * r].b, q4[2 * r].c, q4[2 * r].d,
                                 q4[2 * r + 1].a, q4[2 * r + 1].b, q4[2 * r + 1].c, q4[2 * r + 1].d};
              uint32_t w[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const bool ok = t + 8 * r + j < len;
                w[j] = s_bits[ok ? uu[j] - ubase : 0];
                w[j] = ok ? w[j] : 0u;
              }
              uint32_t ta, tb, fa, fb, e8;
              SLIM_CSA(ta, ones, ones, w[0], w[1]);
              SLIM_CSA(tb, ones, ones, w[2], w[3]);
              SLIM_CSA(fa, twos, twos, ta, tb);
              SLIM_CSA(ta, ones, ones, w[4], w[5]);
              SLIM_CSA(tb, ones, ones, w[6], w[7]);
              SLIM_CSA(fb, twos, twos, ta, tb);
              SLIM_CSA(e8, fours, fours, fa, fb);
              uint32_t carry = e8;  // into plane 3 and upward, while any lane carries
#pragma unroll
              for (int pp = 0; pp < NPL - 3; ++pp) {
                if (__ballot(carry != 0u) == 0ull) break;
                const uint32_t nc = hi[pp] & carry;
                hi[pp] ^= carry;
                carry = nc;
                nhi = nhi > pp + 1 ? nhi : pp + 1;
              }
            }
          }
#undef SLIM_CSA
          // planes -> the P counts of this lane's column
          float cnt[P];
#pragma unroll
          for (int qq = 0; qq < P; ++qq)
            cnt[qq] = (float)(((ones >> qq) & 1u) + (((twos >> qq) & 1u) << 1) + (((fours >> qq) & 1u) << 2));
#pragma unroll
          for (int pp = 0; pp < NPL - 3; ++pp) {
            if (pp < nhi) {
#pragma unroll
              for (int qq = 0; qq < P; ++qq) cnt[qq] += (float)(((hi[pp] >> qq) & 1u) << (pp + 3));
            }
          }
          if (have) {
            float4* const out = reinterpret_cast<float4*>(part + (int64_t)i * P);
#pragma unroll
            for (int j = 0; j < P / 4; ++j)
              out[j] = make_float4(cnt[4 * j], cnt[4 * j + 1], cnt[4 * j + 2], cnt[4 * j + 3]);
          }
        }
      } else {
      // (slices are short here -- a column's nnz over 32 members -- so the slice bounds and the
      // position of the NEXT column are requested while this one is counted)
      int64_t cs_n = 0, ce_n = 0;
      int pos_n = 0;
      if (wave < ncols) {
        cs_n = csplit[(int64_t)wave * gst + mk];
        ce_n = csplit[(int64_t)wave * gst + mk + 1];
        pos_n = S.gram_pos[wave];
      }
      for (int i = wave; i < ncols; i += NW) {
        const int64_t cs = uni(cs_n), ce = uni(ce_n);
        const int pos_i = uni(pos_n);
        if (i + NW < ncols) {
          cs_n = csplit[(int64_t)(i + NW) * gst + mk];
          ce_n = csplit[(int64_t)(i + NW) * gst + mk + 1];
          pos_n = S.gram_pos[i + NW];
        }
        if (pos_i < base) continue;  // (symmetric fill: see the screen pass)
        int cnt[P];
#pragma unroll
        for (int qq = 0; qq < P; ++qq) cnt[qq] = 0;
        for (int64_t jb = cs; jb < ce; jb += 64) {
          const bool ok = jb + lane < ce;
          uint32_t w = 0u;
          if (ok) w = s_bits[ci[jb + lane] - ubase];
          if (__ballot(w != 0u) == 0ull) continue;
#pragma unroll
          for (int qq = 0; qq < P; ++qq) cnt[qq] += __popcll(__ballot((w >> qq) & 1u));
        }
        float v = 0.0f;
#pragma unroll
        for (int qq = 0; qq < P; ++qq) v = lane == qq ? (float)cnt[qq] : v;
        if (lane < P) part[(int64_t)i * P + lane] = v;  // one 128-byte line per column
      }
      }
      cluster_barrier();  // every member's partial sums are published
    }
    // -- clear the interleaved work vectors
    if (!gbits) {
      // (x needs no clearing: the active-set pass below assigns every entry)
      float4* r4 = reinterpret_cast<float4*>(r);
      const int64_t nr4 = (int64_t)(uend - ubase + 1) * (P / 4);  // + the spare line (see visit)
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int64_t k = tid; k < nr4; k += NT) r4[k] = z;
      if (!cached)
        for (int k = tid; k < S.bm_words; k += NT) s_bits[k] = 0u;
    }
    __syncthreads();

    // -- y scatter (estimate.c:406-408): wavefront w serves problems w, w+NW; the users that
    //    hold a rating of any of the tile's items are marked in the LDS bitmap (one bit per
    //    1 << bm_shift users of this member's range)
    const int sh = S.bm_shift;
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
      const int pq = wave + pp * NW;
      const int witem = gbits ? -1 : s_item[pq];
      if (witem >= 0) {
        const int64_t cs = uni(csplit[(int64_t)witem * (K + 1) + mk]);
        const int64_t ce = uni(csplit[(int64_t)witem * (K + 1) + mk + 1]);
        for (int64_t j = cs + lane; j < ce; j += 64) {
          const int u = ci[j] - ubase;
          r[(int64_t)u * P + pq] = HAS_VAL ? cv[j] : 1.0f;
          const uint32_t bit = (uint32_t)u >> sh;
          if (!cached) atomicOr(&s_bits[bit >> 5], 1u << (bit & 31));
        }
      }
    }
    __syncthreads();

    // -- screen pass: aTy_i = a_i . y for EVERY column i, the P problems at once -- what the
    //    reference computes by scanning the whole column view per item (estimate.c:412-421).
    //    Each wavefront takes whole columns (no barrier, no exchange per column): it reads its
    //    slice of the column in 64-nnz blocks, drops the users outside the tile's user set
    //    (bitmap), compacts the rest through the cross-lane network and gathers their residual
    //    lines, SL users x P problems per step.  No atomics: the sum of a column is formed in a
    //    fixed order, so the screen aTy > l1 is reproducible for any rating values.  (The
    //    Gram-column form, sum over the item's users of their rows, needs one device-scope
    //    float atomic per touched (item, problem): 1.2 s of a 12.9 s median tile on C4, 7 s of
    //    the 27 s heaviest tile.)
    if (!cached && !gbits) {
      constexpr int GS = 8;  // gather steps in flight per lane (16: no gain, measured)
      for (int i = wave; i < ncols; i += NW) {
        // building G = R^T R (gram_mode 3): G is symmetric, so a tile only forms the sums of the
        // columns at or behind its own position in the work list (which then holds every
        // column); the mirror entries are written below
        if (S.gram_mode == 3 && uni(S.gram_pos[i]) < base) continue;
        const int64_t cs = uni(csplit[(int64_t)i * (K + 1) + mk]);
        const int64_t ce = uni(csplit[(int64_t)i * (K + 1) + mk + 1]);
        float acc = 0.0f;
        bool co = false;  // FSLIM: a user of the problem's item rated column i (neighbors.c:46-60)
        // ids / values of the next block are requested before the current one is consumed
        int u_n = 0;
        float v_n = 0.0f;
        if (cs + lane < ce) {
          u_n = ci[cs + lane] - ubase;
          v_n = HAS_VAL ? cv[cs + lane] : 1.0f;
        }
        for (int64_t jb = cs; jb < ce; jb += 64) {
          const bool ok = jb + lane < ce;
          const int u = u_n;
          const float v = v_n;
          const int64_t jn = jb + 64 + lane;
          if (jn < ce) {
            u_n = ci[jn] - ubase;
            v_n = HAS_VAL ? cv[jn] : 1.0f;
          }
          const uint32_t bit = (uint32_t)u >> sh;
          const bool f = ok && ((s_bits[bit >> 5] >> (bit & 31)) & 1u);
          const uint64_t m = __ballot(f);
          const int cnt = __popcll(m);
          if (cnt == 0) continue;
          // compaction: the cnt marked entries move to lanes 0 .. cnt-1 (a full permutation,
          // so every lane is written exactly once)
          const int rank = __popcll(m & lane_lt);
          const int dst = (f ? rank : cnt + (lane - rank)) << 2;
          const int cu = __builtin_amdgcn_ds_permute(dst, u);
          const float cw = HAS_VAL ? __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(v)))
                                   : 1.0f;
          for (int t0 = 0; t0 < cnt; t0 += SL * GS) {
            float rr[GS];
#pragma unroll
            for (int g = 0; g < GS; ++g) {
              const int src = t0 + g * SL + slot;
              const int uu = __shfl(cu, src & 63);
              rr[g] = 0.0f;
              if (src < cnt)
                rr[g] = *reinterpret_cast<const float*>(
                    reinterpret_cast<const char*>(r) + (((uint32_t)uu * (uint32_t)(4 * P)) | ((uint32_t)q << 2)));
            }
#pragma unroll
            for (int g = 0; g < GS; ++g) {
              const int src = t0 + g * SL + slot;
              const float w = HAS_VAL ? __shfl(cw, src & 63) : 1.0f;
              acc += w * rr[g];
              if (FSLIM && HAS_VAL) co = co || rr[g] != 0.0f;
            }
          }
        }
        if (SL == 4) acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (FSLIM && HAS_VAL) {
          // ratings that cancel: a co-rated column whose sum is 0 stays a candidate -- the sign
          // bit of the zero carries it through the partial buffer
          if (SL == 4) co = co || (__shfl_xor((int)co, 16) != 0);
          co = co || (__shfl_xor((int)co, 32) != 0);
          if (acc == 0.0f && co) acc = -0.0f;
        }
        if (slot == 0) part[(int64_t)i * P + q] = acc;  // one 128-byte line per column
      }
    }
    if (!cached && !gbits) cluster_barrier();  // every member's partial sums are published
    else __syncthreads();

    // -- active sets: x = 0 for active, -inf for inactive
    if (FSLIM) {
      // FSLIM (estimate.c:424-431, neighbors.c:16-125): the nnbrs columns most similar to the
      // item among those sharing a user with it, no l1 screen.  The co-rating dot products ARE
      // the screen sums.  Selection per problem: a 4-pass radix select over the sortable bits
      // of the similarity finds the nnbrs-th largest value; ties at that value go to the
      // lower item ids (upstream leaves them undefined; the oracle uses the same rule).
      uint32_t* const hist = s_bits;       // [P][256]: the user bitmap is dead by now
      __shared__ uint32_t s_prefix[P], s_want[P];
      __shared__ float s_cn[P];
      if (tid < P) {
        s_prefix[tid] = 0u;
        s_want[tid] = (uint32_t)S.nnbrs;
        s_cn[tid] = s_item[tid] >= 0 ? A.cnorm[s_item[tid]] : 0.0f;
      }
      __syncthreads();
      const int64_t n = (int64_t)ncols * P;
      // similarities (neighbors.c:82-83 cos, :107-109 jac, dotp), -inf for non-candidates
      for (int64_t idx = tid; idx < n; idx += NT) {
        const int i = (int)(idx >> LOGP), qq = (int)(idx & (P - 1));
        const int it = s_item[qq];
        float a = 0.0f;
        bool co = false;
        if (cached) {
          a = gram_t[idx];
          co = a != 0.0f || __builtin_signbit(a);
        } else {
          for (int k = 0; k < K; ++k) {
            const float pk = S.atypart[(int64_t)(cid * K + k) * S.x_stride + idx];
            a += pk;
            co = co || pk != 0.0f || __builtin_signbit(pk);
          }
          if (a == 0.0f) a = co ? -0.0f : 0.0f;  // (-0 + 0 = +0: restore the mark)
          if (gram_t != nullptr && mk == 0) gram_t[idx] = a;
        }
        float sim = kInactive;
        // candidates: the co-rated columns (neighbors.c:46-60), also when the sum cancelled
        if (it >= 0 && i != it && co) {
          const float cn_i = A.cnorm[i];
          sim = S.simtype == 0 ? a / cn_i : (S.simtype == 1 ? a / ((cn_i + s_cn[qq]) - a) : a);
        }
        x[idx] = sim;
      }
      // order-preserving map float -> uint32 (-inf maps to 0x007FFFFF: below every finite value)
      auto keyof = [](const float f) -> uint32_t {
        const uint32_t b = __float_as_uint(f);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      };
      const uint32_t key_ninf = keyof(kInactive);
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int k = tid; k < P * 256; k += NT) hist[k] = 0u;
        __syncthreads();
        for (int64_t idx = tid; idx < n; idx += NT) {
          const int qq = (int)(idx & (P - 1));
          const uint32_t key = keyof(x[idx]);
          if (key == key_ninf) continue;
          const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
          if ((key & hi_mask) == s_prefix[qq]) atomicAdd(&hist[qq * 256 + ((key >> shift) & 255u)], 1u);
        }
        __syncthreads();
        if (tid < P) {  // walk the digits from the top until `want` entries are covered
          uint32_t want = s_want[tid], d = 255u;
          for (;; --d) {
            const uint32_t c = hist[tid * 256 + d];
            if (c >= want || d == 0u) break;
            want -= c;
          }
          // fewer candidates than wanted: the walk ends at digit 0 and everything is taken
          s_prefix[tid] |= d << shift;
          s_want[tid] = want;
        }
        __syncthreads();
      }
      // s_prefix = the threshold key T, s_want = how many entries equal to T to keep (ascending
      // ids).  Wavefront w marks problems w, w + NW, walking the items in order.
#pragma unroll
      for (int pp = 0; pp < PPW; ++pp) {
        const int pq = wave + pp * NW;
        const uint32_t T = s_prefix[pq];
        int quota = (int)s_want[pq], na = 0;
        for (int ib = 0; ib < ncols; ib += 64) {
          const int i = ib + lane;
          const uint32_t key = i < ncols ? keyof(x[(int64_t)i * P + pq]) : key_ninf;
          const bool cand = key != key_ninf;
          const bool tie = cand && key == T;
          const uint64_t mt = __ballot(tie);
          const bool act = cand && (key > T || (tie && __popcll(mt & lane_lt) < quota));
          quota -= __popcll(mt) < quota ? __popcll(mt) : quota;
          if (i < ncols) x[(int64_t)i * P + pq] = act ? 0.0f : kInactive;
          na += __popcll(__ballot(act));
        }
        if (lane == 0) s_na[pq] = na;
      }
    } else if (S.gram_mode == 3) {
      // rows item_q of G and, for the columns of later tiles, their mirror entries (the tile of
      // column i will skip this tile's columns); columns of earlier tiles were skipped.  The K
      // partial sums of an entry are added by ONE member: the members share the entries
      const int64_t n = (int64_t)ncols * P;
      const int64_t per = (n + K - 1) / K;
      const int64_t hi = (mk + 1) * per < n ? (mk + 1) * per : n;
      for (int64_t idx = mk * per + tid; idx < hi; idx += NT) {
        const int i = (int)(idx >> LOGP), qq = (int)(idx & (P - 1));
        const int it = s_item[qq];
        const int pi = S.gram_pos[i];
        if (pi >= base && it >= 0) {
          float a = 0.0f;  // members in rank order
          for (int k = 0; k < K; ++k) a += S.atypart[(int64_t)(cid * K + k) * S.x_stride + idx];
          // (user passes: every pass holds the sums over ITS users; counts are integers below 2^24,
          // so the float additions are exact in any order)
          if (S.gram_accum) a += S.G[(int64_t)it * S.G_ld + i];
          S.G[(int64_t)it * S.G_ld + i] = a;
          if (pi >= base + P) S.G[(int64_t)i * S.G_ld + it] = a;
        }
      }
    } else {  // estimate.c:433-444: aTy > l1 (strict), the item itself excluded
      const int64_t n = (int64_t)ncols * P;
      for (int64_t idx = tid; idx < n; idx += NT) {
        const int i = (int)(idx >> LOGP), qq = (int)(idx & (P - 1));
        const int it = s_item[qq];
        float a = 0.0f;  // members in rank order: the same sum on every member
        if (cached) {
          a = gram_t[idx];
        } else {
          for (int k = 0; k < K; ++k)
            a += S.atypart[(int64_t)(cid * K + k) * S.x_stride + idx];
          if (gram_t != nullptr && mk == 0) gram_t[idx] = a;
        }
        const bool act = it >= 0 && i != it && a > l1;
        x[idx] = act ? 0.0f : kInactive;
        if (act) atomicAdd(&s_na[qq], 1);
      }
    }
    __syncthreads();
    if (S.gram_mode == 3) continue;  // (the queue pull of the next round is the cluster's sync point)

    // -- warm start (estimate.c:453-464): previous coefficients of active coordinates
    // (in the FSLIM branch the reference never sets its warm-start flags: a no-op there)
    const bool warm = FOLD != 0 && S.icolptr != nullptr && !FSLIM;
    if (warm) {
#pragma unroll
      for (int pp = 0; pp < PPW; ++pp) {
        const int pq = wave + pp * NW;
        const int witem = s_item[pq];
        if (witem >= 0 && witem < S.incols) {
          const int64_t ws = uni(S.icolptr[witem]), we = uni(S.icolptr[witem + 1]);
          for (int64_t e = ws + lane; e < we; e += 64) {
            const int k = S.icolind[e];
            if (k < ncols) {
              const int64_t a = (int64_t)k * P + pq;
              // (estimate.c:456-464: a negative previous value is copied, then reset to 0)
              if (tile_active(x[a])) x[a] = fmaxf(S.icolval[e], 0.0f);
            }
          }
        }
      }
      __syncthreads();
    }

    // -- union of the tile's active sets, ascending (wavefront 0)
    if (wave == 0) {
      int nu = 0;
      for (int ib = 0; ib < ncols; ib += 64) {
        const int i = ib + lane;
        bool any = false;
        if (i < ncols) {
          const float4* row = reinterpret_cast<const float4*>(x + (int64_t)i * P);
#pragma unroll
          for (int c = 0; c < P / 4; ++c) {
            const float4 f = row[c];
            any = any || tile_active(f.x) || tile_active(f.y) || tile_active(f.z) || tile_active(f.w);
          }
        }
        const uint64_t m = __ballot(any);
        if (any) ul[nu + __popcll(m & lane_lt)] = i;
        nu += __popcll(m);
      }
      if (lane == 0) s_nunion = nu;
    }
    __syncthreads();
    const int nunion = s_nunion;

    // -- per-problem state: "done" replicated in every lane that serves problem q; what the
    //    visit loop only accumulates or rarely reads (traffic counters, sweep cap and count)
    //    lives in LDS, kept by the lanes 0 .. P-1 of wavefront 0 -- registers that stay live
    //    across the visit loop are what the two gather buffers of a visit compete with
    const int item_q = s_item[q];
    bool done_q = item_q < 0;
    if (tid < P) {
      int maxit = 0;
      if (item_q >= 0) {
        const int64_t cap = 50 * (colptr[item_q + 1] - colptr[item_q]);  // estimate.c:448-449
        maxit = cap < (int64_t)S.maxniters ? (int)cap : S.maxniters;
      }
      s_maxit[tid] = maxit;
      s_niters[tid] = 0;
      s_conv[tid] = 0;
      s_D[tid] = 0;
      s_U[tid] = 0;
    }
    __syncthreads();
    int buf = 0;
    // PROFILE: [0] loads of the dot, [1] reduce + barrier, [2] update math, [3] stores issued,
    // [4] closing barrier, [5] visits, [6] visits with an update
    uint64_t prof[7] = {0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> uint64_t {
      if (!PROFILE) return 0;
      __builtin_amdgcn_s_waitcnt(0);
      return clock64();
    };

    const char* __restrict__ rb = reinterpret_cast<const char*>(r);
    char* __restrict__ rbw = reinterpret_cast<char*>(r);
    const uint32_t qoff = (uint32_t)q << 2;

    // One coordinate: dot for the P problems, update, residual axpy.
    // mode 0: CD visit; mode 1: fold the warm-start coefficients into r (cd.c:108-110)
    // [s, e) is this member's slice of column i, len the length of the whole column
    // HI (latency-bound: slices of a few hundred nnz): the ids of the next visit's first
    // block are requested while this visit waits for the cluster, [sn_v, sn_v + nn_v) = that
    // slice
    constexpr int NR = P / 16;  // id / value registers per 64-nnz block (see row_bcast)
    const int ent0 = slot * P + (lane & 15);  // block entry this lane loads into register 0
    const int udummy = uend - ubase;  // the spare line behind this member's user range (always 0)
    int pf_id[NR];
    float pf_v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      pf_id[k] = 0;
      pf_v[k] = 0.0f;
    }
    int64_t pf_at = -1;  // slice start the prefetched block belongs to (-1: none)
    auto visit = [&](const int i, const int64_t s, const int64_t e, const int64_t len,
                     const float xi, const float cn, const float sq, const bool live, float& dlt,
                     const int mode, const int64_t sn_v, const int nn_v) {
      int64_t pf_here = pf_at;  // valid for the first load of the first block only
      pf_at = -1;
      const bool part = live && tile_active(xi);
      if (!__any(part)) return;
      constexpr int64_t CH = 64 * NW;  // nnz per workgroup chunk
      // One 64-nnz block of this wavefront: ids / values (row_bcast layout), the gathered
      // residuals, nh = valid nnz of the block.  Two
      // blocks alternate: the gathers of the next block of a slice are issued before the
      // current one is consumed, so that a slice of several chunks is one stream of requests
      // instead of one memory round trip per chunk (ids, then lines).
      struct Blk {
        int id[NR];
        float v[NR];
        float r[STEPS];
        // entries [lo, nh) of the block lie inside the slice (ids) / [lov, nhv) (values)
        int nh, nhv, lo, lov;
      };
      // entries of a 64-nnz block starting at b0 that lie inside [s, e): [lo, nh)
      auto span_of = [&](const int64_t b0, int& lo, int& nh) {
        const int64_t left = e - b0, under = s - b0;
        nh = left <= 0 ? 0 : (left < 64 ? (int)left : 64);
        lo = under <= 0 ? 0 : (under < 64 ? (int)under : 64);
      };
      // uniform base + lane offset of entry `ent`, clamped into the slice (a block that misses the
      // slice altogether reads one valid element of the array)
      auto base_of = [&](const int64_t b0, const int lo, const int nh) -> int64_t {
        if (nh > lo) return b0;
        return b0 < 0 ? 0 : (b0 < S.nnz_last ? b0 : S.nnz_last);
      };
      auto off_of = [&](const int ent, const int lo, const int nh) -> uint32_t {
        if (nh <= lo) return 0u;
        const int c = ent < lo ? lo : (ent < nh ? ent : nh - 1);
        return (uint32_t)c;
      };
      auto inside = [&](const int ent, const int lo, const int nh) -> bool {
        return (uint32_t)(ent - lo) < (uint32_t)(nh - lo);  // (nh <= lo: never)
      };
      Blk A, B;
      // this wavefront's 64 consecutive nnz of chunk c0: one coalesced request per array and
      // register (the rows of a lane group load the same 16 entries).  Entries past the end of
      // the slice become (user = the spare line behind this member's range, which holds 0 and
      // stays 0; value 0): gathers, dot and write-back then need no predication at all.
      // (ids and values are requested separately: the ids of a block are free to be replaced
      // once its gathers are issued, its values only after it has been summed)
      auto load_idx = [&](Blk& b, const int64_t c0) {
        const int64_t b0 = c0 + 64 * wave;
        span_of(b0, b.lo, b.nh);
        // unconditional loads of the raw entries (a load under a condition makes the number
        // of requests in flight path-dependent, and the compiler then waits for ALL of them
        // where it only needs the oldest): uniform base + 32-bit lane offset, clamped into the
        // slice; gather() turns them into line numbers when it needs them
        const int32_t* __restrict__ cb = ci + base_of(b0, b.lo, b.nh);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const uint32_t ec = off_of(ent0 + 16 * k, b.lo, b.nh);
          b.id[k] = cb[ec];  // (raw user id: no arithmetic on it here, that would wait)
        }
      };
      auto load_val = [&](Blk& b, const int64_t c0) {  // after load_idx of the same block
        const int64_t b0 = c0 + 64 * wave;
        span_of(b0, b.lov, b.nhv);
        const float* __restrict__ vb = cv + base_of(b0, b.lov, b.nhv);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const uint32_t ec = off_of(ent0 + 16 * k, b.lov, b.nhv);
          b.v[k] = HAS_VAL ? vb[ec] : 1.0f;
        }
      };
      auto load_ids = [&](Blk& b, const int64_t c0) {
        if (HI && c0 == pf_here) {  // requested during the previous visit
          span_of(c0 + 64 * wave, b.lo, b.nh);
          b.nhv = b.nh;
          b.lov = b.lo;
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            b.id[k] = pf_id[k];
            b.v[k] = pf_v[k];
          }
          pf_here = -1;  // (pf_id is reused for the next visit before this block is re-read)
        } else {
          load_idx(b, c0);
          load_val(b, c0);
        }
      };
      // values of the entries past the slice are 0 (idempotent)
      auto mask_val = [&](Blk& b) {
#pragma unroll
        for (int k = 0; k < NR; ++k) b.v[k] = inside(ent0 + 16 * k, b.lov, b.nhv) ? b.v[k] : 0.0f;
      };
      // raw user ids -> line numbers; entries past the slice -> the spare line (binary
      // matrices: the values are 1 for the entries of the slice)
      auto fix_ids = [&](Blk& b) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          b.id[k] = inside(ent0 + 16 * k, b.lo, b.nh) ? b.id[k] - ubase : udummy;
          if (!HAS_VAL) b.v[k] = 1.0f;
        }
        if (!HAS_VAL) {
          b.nhv = b.nh;
          b.lov = b.lo;
        }
      };
      // gather the residual lines of the block: STEPS loads per lane in flight
      auto gather = [&](Blk& b) {
        fix_ids(b);
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
          const int u = row_bcast(b.id[j >> 4], j & 15);
          b.r[j] = *reinterpret_cast<const float*>(rb + (((uint32_t)u * (uint32_t)(4 * P)) | qoff));
          // (address -> load, one step at a time: hoisting the 32 address computations of a
          // block above its loads costs 32 registers the second block needs.  No use of the
          // values here: the block is consumed after the NEXT block's loads have been issued.)
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto dot_block = [&](Blk& b) -> float {
        float a = 0.0f;
        if (HAS_VAL) mask_val(b);
        {
#pragma unroll
          for (int j = 0; j < STEPS; ++j)  // (entries past the slice gathered the spare line: 0)
            a += HAS_VAL ? row_bcast(b.v[j >> 4], j & 15) * b.r[j] : b.r[j];
        }
        return a;
      };
      // whole lines: every problem's value of the user is written back (entries past the slice
      // write 0 - d * 0 to the spare line)
      auto scatter = [&](Blk& b, const float d) {
        mask_val(b);
        {
#pragma unroll
          for (int j = 0; j < STEPS; ++j) {
            const int u = row_bcast(b.id[j >> 4], j & 15);
            const float v = row_bcast(b.v[j >> 4], j & 15);
            *reinterpret_cast<float*>(rbw + (((uint32_t)u * (uint32_t)(4 * P)) | qoff)) =
                b.r[j] - d * v;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };

      if (FOLD == 1 && mode == 1) {
        // fold of a warm-start coefficient: r -= x_i a_i, one pass per chunk (ids, gather,
        // write-back); no dot, no exchange.  Two folds can touch the same user line: the
        // closing barrier stays.
        const float d = (part && (xi > kEps || xi < -kEps)) ? xi : 0.0f;
        if (!__any(d != 0.0f)) return;
        for (int64_t c = s; c < e; c += CH) {
          load_ids(A, c);
          gather(A);
          scatter(A, d);
        }
        __syncthreads();
        return;
      }

      const uint64_t p0 = tick();
      float acc = 0.0f;
      // The chunks of the slice alternate between the two blocks so that the last one lands in
      // A (it stays in registers for the update): with an even number of chunks the first one
      // is consumed on its own, the others pairwise -- the gathers of chunk k+1 are in flight
      // before chunk k is summed.
      const int64_t nchunks = e > s ? (e - s + CH - 1) / CH : 1;
      // the chunk grid: [sg, e) in nchunks chunks of CH (end-aligned: sg <= s, the first chunk
      // holds the slice's remainder)
      const int64_t sg = SLIM_TILE_END_ALIGNED ? e - nchunks * CH : s;
      int64_t c0 = sg;
      // a wavefront whose block of a one-chunk slice misses the slice has nothing to do (one
      // branch around the whole stream; inside it every load is unconditional)
      const bool mine = nchunks > 1 || (sg + 64 * wave < e && sg + 64 * wave + 64 > s);
      if (mine && HAS_VAL) {
        // valued matrices: one block at a time (the values of two blocks on top of their
        // residual lines do not fit the register budget: measured as 40 spills in this loop)
        for (; c0 + CH < e; c0 += CH) {
          load_ids(A, c0);
          gather(A);
          acc += dot_block(A);
        }
        load_ids(A, c0);
        gather(A);
      }
      // Binary matrices: two blocks in flight.  Chunk k of the slice lands in A when
      // (nchunks - 1 - k) is even, else in B, so that the LAST chunk ends up in A and the lines
      // of the one before it in B: both stay in registers for the update (a slice of one or two
      // chunks is never gathered twice).  Loads complete in issue order, so the ids of a chunk
      // are requested BEFORE the gathers of the chunk ahead of it (a block's id registers are
      // free as soon as its gathers are issued) -- waiting for them then does not drain those
      // gathers:   ids(k+2), lines(k+1), [sum k], ids(k+3), lines(k+2), [sum k+1], ...
      if (mine && !HAS_VAL) {
        if (nchunks & 1) {  // b0 -> A
          load_ids(A, c0);
          load_idx(B, c0 + CH);
          gather(A);
          c0 += CH;
        } else {  // b0 -> B, b1 -> A
          load_ids(B, c0);
          load_idx(A, c0 + CH);
          gather(B);
          load_idx(B, c0 + 2 * CH);
          gather(A);
          acc += dot_block(B);
          c0 += 2 * CH;
        }
        // here: A in flight = the chunk at c0 - CH, B's ids = those of the chunk at c0
        while (c0 < e) {
          load_idx(A, c0 + CH);
          gather(B);
          acc += dot_block(A);
          load_idx(B, c0 + 2 * CH);
          gather(A);
          acc += dot_block(B);
          c0 += 2 * CH;
        }
        c0 -= CH;  // start of the last chunk (in A)
      }
      // (sn_v / nn_v are still in flight when the visit starts: made uniform only here.)  The
      // loads are unconditional instructions with a clamped address -- lanes past the slice hold
      // garbage that load_ids never looks at -- so that exactly one (two with values) load is in
      // flight behind the gather on every path and the dot below waits for the gather only
      // (s_waitcnt vmcnt(1|2)); a load under a condition makes the count path-dependent and the
      // compiler drains the queue instead.
      if (HI) {
        const int64_t sn = uni(sn_v);
        const int nn = uni(nn_v);
        // (the first chunk of the next visit's grid)
        const int64_t sgn = SLIM_TILE_END_ALIGNED ? sn + nn - ((nn + CH - 1) / CH) * CH : sn;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          int64_t jj = sgn + 64 * wave + ent0 + 16 * k;
          jj = jj < S.nnz_last ? jj : S.nnz_last;
          jj = jj < 0 ? 0 : jj;
          pf_id[k] = ci[jj];
          pf_v[k] = HAS_VAL ? cv[jj] : 1.0f;
        }
        pf_at = (nn > 0 && S.hi_prefetch) ? sgn : -1;
      }
      const uint64_t p1 = tick();

      float d = 0.0f, nx = xi;
      uint64_t p2 = p1;
      {
        if (mine) acc += dot_block(A);
        if (SL == 4) acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (slot == 0) s_part[buf][wave][q] = acc;
        __syncthreads();
        float dot = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) dot += s_part[buf][w][q];
        buf ^= 1;
        dot = cluster_sum(dot);
        p2 = tick();
        const float xeff = (xi > kEps || xi < -kEps) ? xi : 0.0f;
        const float num = dot + xeff * sq;
        nx = num > l1 ? (num - l1) / (cn * cn + l2) : 0.0f;
        const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
        d = neff - xeff;
        if (!part) {
          d = 0.0f;
          nx = xi;
        } else {
          dlt += (nx - xi) * (nx - xi);
          if (tid < P) {  // traffic counters of SURVEY.md 8(d)
            s_D[q] += (unsigned long long)len;
            if (d != 0.0f) s_U[q] += (unsigned long long)len;
          }
        }
      }
      const bool upd = __any(d != 0.0f);
      const bool xch = __any(part && nx != xi);
      const uint64_t p3 = tick();
      if (upd && mine) {
        scatter(A, d);  // the last chunk: still in registers
        int64_t stop = c0;  // chunks [s, stop) are read again (L2 / Infinity Cache)
        if (!HAS_VAL && nchunks >= 2) {
          // binary: so are the LINES of the chunk before it; its ids are read again (4 bytes
          // per nnz from L2 instead of a 128-byte line per nnz)
          stop = c0 - CH;
          load_idx(B, stop);
          fix_ids(B);
          scatter(B, d);
        }
        // the ids of the next chunk are requested before the lines of the current one are
        // waited for
        if (sg < stop) {
          load_ids(A, sg);
          for (int64_t c = sg; c < stop; c += CH) {
            gather(A);
            if (c + CH < stop) load_ids(B, c + CH);
            scatter(A, d);
            if (c + CH < stop) {
#pragma unroll
              for (int k = 0; k < NR; ++k) {
                A.id[k] = B.id[k];
                A.v[k] = B.v[k];
              }
              A.nh = B.nh;
              A.nhv = B.nhv;
              A.lo = B.lo;
              A.lov = B.lov;
            }
          }
        }
      }
      if (xch && wave == 0 && slot == 0 && part && nx != xi) x[(int64_t)i * P + q] = nx;
      const uint64_t p4 = PROFILE ? clock64() : 0;
      if (upd || xch) __syncthreads();
      if (PROFILE) {
        const uint64_t p5 = clock64();
        prof[0] += p1 - p0;
        prof[1] += p2 - p1;
        prof[2] += p3 - p2;
        prof[3] += p4 - p3;
        prof[4] += p5 - p4;
        prof[5] += 1;
        prof[6] += upd ? 1 : 0;
      }
    };

    const uint64_t t_setup = wall_clock64();
    if (FOLD == 1 && warm) {
      float unused = 0.0f;
      for (int p = 0; p < nunion; ++p) {
        const int i = uni(ul[p]);
        const int64_t* sp = csplit + (int64_t)i * (K + 1);
        visit(i, uni(sp[mk]), uni(sp[mk + 1]), uni(sp[K]) - uni(sp[0]), x[(int64_t)i * P + q],
              0.0f, 0.0f, !done_q, unused, 1, 0, 0);
      }
    }
    if (FOLD == 2 && warm) {
      // Row-wise fold: r[u][q] = y[u][q] - sum_{j in row u} v_uj x_j[q] over this member's users.
      // Every member wrote the same warm-start values into its own x; all of them gather from
      // member 0's copy, so that a cluster keeps ONE [ncols][P] array hot (C5: 2.5 MB; the K
      // private copies of a cluster would be 20 MB).  One user per wavefront at a time, the 64
      // ids of a block of its row spread over the lanes, every lane group gathering the lines
      // of its share (STEPS loads in flight per lane), the ids of the next block -- of this row
      // or of the wavefront's next user -- requested before the gathers of the current one.
      // The residual line of a user is read (it holds y) and written once.
      cluster_barrier();  // member 0's x is complete and visible
      {
        const float* __restrict__ xs = S.xslab + (int64_t)(bid - mk) * S.x_stride;
        const int64_t* __restrict__ rp = A.rowptr + ubase;
        const int32_t* __restrict__ ri = A.rowind;
        const float* __restrict__ rv = A.rowval;
        const int nu = uend - ubase;
        int u = wave;
        if (u < nu) {
          int64_t b = uni(rp[u]), re = uni(rp[u + 1]);
          int64_t rs_n = 0, re_n = 0;  // row of the wavefront's next user
          if (u + NW < nu) {
            rs_n = rp[u + NW];
            re_n = rp[u + NW + 1];
          }
          int id_n[NR];
          float v_n[NR];
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            int64_t jj = b + ent0 + 16 * k;
            jj = jj < S.nnz_last ? jj : S.nnz_last;
            id_n[k] = ri[jj];
            v_n[k] = HAS_VAL ? rv[jj] : 1.0f;
          }
          float acc = 0.0f;
          float yv = r[(int64_t)u * P + q];
          for (;;) {
            const int64_t left = re - b;
            const int nh = left <= 0 ? 0 : (left < 64 ? (int)left : 64);
            int id[NR];
            float v[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) {  // entries past the row: item 0 with value 0
              const bool ok = ent0 + 16 * k < nh;
              id[k] = ok ? id_n[k] : 0;
              v[k] = ok ? v_n[k] : 0.0f;
            }
            // what comes after this block
            const bool row_end = left <= 64;
            const int un = u + NW;
            int64_t nb = b + 64;
            if (row_end) nb = uni(rs_n);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
              int64_t jj = nb + ent0 + 16 * k;
              jj = jj < S.nnz_last ? jj : S.nnz_last;
              jj = jj < 0 ? 0 : jj;
              id_n[k] = ri[jj];
              v_n[k] = HAS_VAL ? rv[jj] : 1.0f;
            }
            float xg[STEPS];
#pragma unroll
            for (int j = 0; j < STEPS; ++j) {
              const int jj = row_bcast(id[j >> 4], j & 15);
              xg[j] = xs[(uint32_t)jj * (uint32_t)P + (uint32_t)q];
            }
#pragma unroll
            for (int j = 0; j < STEPS; ++j) {
              const float xv = xg[j];
              // the coefficients that enter the residual (cd.c:27), inactive = -inf = none
              const float xe = (xv > kEps || (xv < -kEps && tile_active(xv))) ? xv : 0.0f;
              acc += row_bcast(v[j >> 4], j & 15) * xe;
            }
            if (row_end) {
              if (SL == 4) acc += __shfl_xor(acc, 16);
              acc += __shfl_xor(acc, 32);
              if (slot == 0) r[(int64_t)u * P + q] = yv - acc;
              acc = 0.0f;
              u = un;
              if (u >= nu) break;
              b = nb;
              re = uni(re_n);
              yv = r[(int64_t)u * P + q];
              if (u + NW < nu) {
                rs_n = rp[u + NW];
                re_n = rp[u + NW + 1];
              }
            } else {
              b = nb;
            }
          }
        }
      }
      cluster_barrier();  // nobody reads member 0's x any more: the sweeps may change it
    }
    const uint64_t t_fold = wall_clock64();

    // -- sweeps (cd.c:112-139)
    for (int t = 0;; ++t) {
      if (!done_q && t >= s_maxit[q]) {  // loop exhausted without convergence: niters = t + 1
        done_q = true;
        if (tid < P) s_niters[q] = s_maxit[q] + 1;
      }
      const bool live = !done_q;
      if (!__any(live) || s_abort) break;
      float dlt = 0.0f;
      const PermCtx pc = perm_make((uint32_t)nunion, perm_key(S.seed, gkey, (uint32_t)t));
      if (nunion > 0) {
        // software pipeline on the visit scalars: the column id is read two visits ahead, the
        // slice offsets / x row / norms one visit ahead (values stay in VGPRs until consumed)
        // (every address below = uniform base + 32-bit lane offset: the scalar-base form of
        // the load, no 64-bit per-lane pointers kept across the visit)
        const uint32_t sp_stride = (uint32_t)(K + 1) * 8u;
        auto meta = [&](const int i1, int64_t& s_o, int& n_o, int& l_o, float& xi_o, float& cn_o,
                        float& sq_o) {
          const uint32_t so = (uint32_t)i1 * sp_stride;
          const int64_t a = ld_off<int64_t>(csplit, so + (uint32_t)mk * 8u);
          const int64_t b = ld_off<int64_t>(csplit, so + (uint32_t)mk * 8u + 8u);
          const int64_t c = ld_off<int64_t>(csplit, so);
          const int64_t d = ld_off<int64_t>(csplit, so + (uint32_t)K * 8u);
          s_o = a;
          n_o = (int)(b - a);
          l_o = (int)(d - c);
          xi_o = ld_off<float>(x, (uint32_t)i1 * (uint32_t)(4 * P) + qoff);
          cn_o = ld_off<float>(A.cnorm, (uint32_t)i1 * 4u);
          sq_o = ld_off<float>(A.csq, (uint32_t)i1 * 4u);
        };
        int i_n1 = ld_off<int>(ul, perm_index(pc, 0u) * 4u);
        int i_n2 = nunion > 1 ? ld_off<int>(ul, perm_index(pc, 1u) * 4u) : 0;
        int64_t s_n;
        int n_n, l_n;
        float xi_n, cn_n, sq_n;
        meta(i_n1, s_n, n_n, l_n, xi_n, cn_n, sq_n);
        for (int p = 0; p < nunion; ++p) {
          const int i = uni(i_n1);
          const int64_t s = uni(s_n), e = s + uni(n_n), len = uni(l_n);
          const float xi = xi_n, cn = uni(cn_n), sq = uni(sq_n);
          if (p + 1 < nunion) {
            i_n1 = i_n2;
            meta(i_n1, s_n, n_n, l_n, xi_n, cn_n, sq_n);
            if (p + 2 < nunion) i_n2 = ld_off<int>(ul, perm_index(pc, (uint32_t)(p + 2)) * 4u);
          }
          const bool more = p + 1 < nunion;
          visit(i, s, e, len, xi, cn, sq, live, dlt, 0, s_n, more ? n_n : 0);
        }
      }
      if (live && dlt < S.opt_tol) {  // cd.c:135-138
        done_q = true;
        if (tid < P) {
          s_conv[q] = 1;
          s_niters[q] = t + 1;
        }
      }
    }

    const uint64_t t_sweeps = wall_clock64();
    if (s_abort) return false;  // aborted launch: this tile's columns are not reported
    // -- 1/2 ||r||^2 and the objective, per problem (estimate.c:477-489)
    {
      float e2 = 0.0f, reg = 0.0f;
      const int g = wave * SL + slot;
      for (int u = g; u < uend - ubase; u += NW * SL) {
        const float rv = r[(int64_t)u * P + q];
        e2 += rv * rv;
      }
      for (int i = g; i < ncols; i += NW * SL) {
        const float xv = x[(int64_t)i * P + q];
        if (tile_active(xv)) reg += 0.5f * l2 * xv * xv + l1 * fabsf(xv);
      }
      if (SL == 4) {
        e2 += __shfl_xor(e2, 16);
        reg += __shfl_xor(reg, 16);
      }
      e2 += __shfl_xor(e2, 32);
      reg += __shfl_xor(reg, 32);
      if (slot == 0) {
        s_red[0][wave][q] = e2;
        s_red[1][wave][q] = reg;
      }
    }
    __syncthreads();
    float err_q = 0.0f;  // 1/2 ||r||^2 of problem q over all members' users
    for (int w = 0; w < NW; ++w) err_q += s_red[0][w][q];
    err_q = 0.5f * cluster_sum(err_q);

    // -- output: wavefront w of member 0 compacts problems w, w+16 (estimate.c:492-505)
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
      const int pq = wave + pp * NW;
      const int witem = s_item[pq];
      if (witem < 0 || mk != 0) continue;
      float reg = 0.0f;
      for (int w = 0; w < NW; ++w) reg += s_red[1][w][pq];
      const float err = lane_bcast(err_q, pq);
      int nz = 0;
      for (int ib = 0; ib < ncols; ib += 64) {
        const int i = ib + lane;
        const float xv = i < ncols ? x[(int64_t)i * P + pq] : kInactive;
        nz += __popcll(__ballot(tile_active(xv) && fabsf(xv) > kEps));
      }
      unsigned long long off = 0;
      if (lane == 0) off = atomicAdd(S.out_cursor, (unsigned long long)nz);
      off = (unsigned long long)uni((int64_t)off);
      const bool fits = (int64_t)(off + (unsigned long long)nz) <= S.out_cap;
      if (fits) {
        int wpos = 0;
        for (int ib = 0; ib < ncols; ib += 64) {
          const int i = ib + lane;
          const float xv = i < ncols ? x[(int64_t)i * P + pq] : kInactive;
          const bool keep = tile_active(xv) && fabsf(xv) > kEps;
          const uint64_t m = __ballot(keep);
          if (keep) {
            const int64_t dst = (int64_t)off + wpos + __popcll(m & lane_lt);
            S.out_ind[dst] = i;
            S.out_val[dst] = xv;
          }
          wpos += __popcll(m);
        }
      }
      const int niters = s_niters[pq];
      const int conv = s_conv[pq];
      const int64_t Dw = (int64_t)s_D[pq], Uw = (int64_t)s_U[pq];
      if (lane == 0) {
        if (!fits) atomicMax(S.overflow, 1);
        S.out_cnt[witem] = fits ? nz : -nz - 1;
        S.out_off[witem] = (int64_t)off;
        S.st_na[witem] = s_na[pq];
        S.st_sweeps[witem] = niters;
        S.st_conv[witem] = conv;
        S.st_D[witem] = Dw;
        S.st_U[witem] = Uw;
        S.st_err[witem] = err;
        S.st_obj[witem] = err + reg;
      }
    }
    if (S.trace != nullptr && tid == 0 && mk == 0) {
      uint64_t* tr = S.trace + (int64_t)grp * 8;
      tr[0] = t_start;
      tr[1] = t_setup;
      tr[2] = t_sweeps;
      tr[3] = wall_clock64();
      tr[4] = (uint64_t)bid;
      tr[5] = (uint64_t)nunion;
      tr[6] = (uint64_t)K;
      tr[7] = t_fold;
      if (PROFILE) {
        uint64_t* pr = S.trace + (int64_t)S.ngroups * 8 + (int64_t)grp * 8;
        for (int k = 0; k < 7; ++k) pr[k] = prof[k];
      }
    }
    __syncthreads();
  }
  return true;
}

// PROFILE adds s_memtime stamps around the phases of a visit (SLIM_GPU_TRACE=2); the
// waits it needs perturb the schedule a little, so it is a separate instantiation.
// NW = wavefronts per workgroup: 16 (one workgroup per CU) or 8 (two per CU, whose phases
// -- gather / barrier / cluster exchange / write-back -- then overlap).
// (16 wavefronts per CU either way: the second launch bound, 4 waves per SIMD, caps the
// kernel at 128 VGPRs)
// FSLIM (neighbour selection instead of the l1 screen) is a separate instantiation: it needs
// 32 KB of LDS for its select histograms whatever the matrix, and the visit loop sits at the
// 128-VGPR cap, where more code in the kernel is not free.  (Same-box A/B runs,
// profiles/r02/ab_variants.txt, rejected parking a chunk of a visit in LDS, pipelining the id
// loads of a visit's chunks, an 8-wavefront / 256-VGPR form of the workgroup and a row-wise
// warm-start fold.)
template <int P, bool HAS_VAL, bool PROFILE, int NW, bool FSLIM = false, int FOLD = 1>
__global__ __launch_bounds__(64 * NW, 4) void cd_tile_kernel(const DevMatrix A, const SolveArgs S) {
  uint32_t epoch = 0;
  // heavy phase first: whole clusters of S.cluster_hi only (S.cluster divides S.cluster_hi,
  // so the workgroups of a big cluster regroup into whole small ones afterwards)
  if (S.nheavy > 0 && tile_block_id(S) < ((int)gridDim.x / S.cluster_hi) * S.cluster_hi) {
    if (!tile_phase<P, HAS_VAL, PROFILE, NW, FSLIM, true, FOLD>(A, S, epoch)) return;
    __syncthreads();
  }
  tile_phase<P, HAS_VAL, PROFILE, NW, FSLIM, false, FOLD>(A, S, epoch);
}

}  // namespace slimamd
