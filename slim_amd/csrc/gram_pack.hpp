// gram_pack.hpp -- G = R^T R as byte planes (the form the item-space kernel of cd_gramr.hpp streams).
//
// An update of item-space CD reads one row of G (cd_gram.hpp): 4 ncols bytes as floats.  For a
// binary or small-integer rating matrix every entry of G is a non-negative integer below 2^24
// (a co-rating count, or a sum of products of small integers) and most of them are small:
// G_ij ~ nnz_i nnz_j / nrows, so in a row only the columns of the most popular items reach 256
// (C4: ~10 % of an average row; nothing reaches 65 536 outside the 160 x 160 block of the top
// items).  The packed form stores a row as
//
//     lo [ncols]      bits 0..7 of every entry, columns in POPULARITY order (rank 0 = the item
//                     with the most ratings, ties by id)
//     hi [hi_k 8192]  bits 8..15 of the first hi_k groups of 8192 ranks -- behind the last rank of
//                     the row that needs them the plane simply ends
//     hi2[hi2_k 8192] bits 16..23, likewise
//
// Behind that prefix a 16-rank chunk is stored RELATIVE to a per-chunk base: adjacent ranks are
// about equally popular, so the 16 entries of a chunk spread by a few standard deviations of a
// count around a common level (C4: ~270 +- 50) -- `lo` holds entry - 16 b, `base` the byte b
// (floor(min / 16), at most 255); a chunk whose entries do not fit that (spread above 255, level
// above 4080) extends the prefix.  Decoded as (float)(lo + 16 b + 256 hi + 65536 hi2), exactly.
//
// The DIAGONAL entry G_ii = |a_i|^2 is not a co-rating count of two different items: it sits at
// the row's own rank and is as large as the row's largest entries (the first version kept it in the
// planes, and the hi plane of an item of rank r then reached to r: 54-63 % of a row on C4 instead of
// its popular head).  The planes hold a filler there (the chunk's base level) and the kernel puts
// `diag[i]` in its place when it decodes that chunk.
//
// i.e. ~1.1-1.2 bytes per entry instead of 4, decoded exactly: (float)(lo + 256 hi + 65536 hi2)
// is the float the unpacked G holds, so a kernel that reads the planes performs the very fmaf
// sequence a kernel that reads the floats performs.  A matrix whose G holds anything else
// (fractional or negative ratings) is not packed and stays with the float kernels.
//
// Ranks (not ids) index the planes' columns and the kernel's g vector; rank_of / item_of
// translate.  Groups of 8192 ranks = 512 threads x 16 bytes: the unit in which cd_gramr.hpp's
// workgroup walks a row.
#pragma once
#include <cstdint>

#include "cd_wave.hpp"

namespace slimamd {

constexpr int kGramrNT = 512;     // threads of cd_gramr.hpp's workgroup
constexpr int kPackGroup = 8192;  // ranks per group: 512 threads x one 16-byte load
constexpr int kGramrMaxGroups = 13;  // groups of the largest instantiation, cd_gramr_kernel<10, 3>: 106 496 items
constexpr int kGramrCarryMaxGroups = 6;  // instantiations that can carry g from solve to solve (cd_gramr.hpp, g_save / g_load)

struct GramPacked {
  const uint8_t* lo;       // [ncols][ldb]
  int64_t ldb;             // bytes per row of lo: ncols rounded up to 16
  const uint8_t* hi;       // pool of hi planes, row i at hi_off[i], hi_k[i] * 8192 bytes
  const int64_t* hi_off;
  const int32_t* hi_k;
  const int64_t* hi2_off;  // bits 16-23 of the first hi2_k[i] groups: in the same pool, right behind the row's hi
  const int32_t* hi2_k;    // groups (hi2_off[i] = hi_off[i] + hi_k[i] * 8192: the solver needs no second offset)
  const uint8_t* base;     // [ncols][8192]: byte [t * 16 + k] = b of chunk t + 512 k (0 inside the hi prefix)
  const float* diag;       // [ncols]: G_ii, kept out of the planes (see below)
  const uint4* meta;       // [ncols]: what the solver needs of a row in ONE 16-byte record, so that a lane
                           // can load it with its batch header: {rank | hi_k << 17 | hi2_k << 21,
                           // hi_off / 8192, nnz of the column, bits of |a_i|^2 = G_ii}
  const int32_t* rank_of;  // [ncols]
  const int32_t* item_of;  // [nchunks * 16]; -1 behind ncols
  int32_t nchunks;         // 16-rank chunks of a row: ceil(ncols / 16)
};

#ifdef SLIM_GRAM_PACK_KERNELS  // (defined by the one translation unit that owns the two kernels)
// Pass 1, one workgroup per row: how many groups need the hi / hi2 plane (use_base: chunks behind
// the prefix may be stored relative to a base byte), and whether the row can be packed at all
// (flags[0] |= 1 otherwise).
__global__ __launch_bounds__(256) void gram_pack_scan(const float* __restrict__ G, int64_t ld, int ncols,
                                                      const int32_t* __restrict__ item_of, int nchunks,
                                                      int use_base, int32_t* __restrict__ hi_k,
                                                      int32_t* __restrict__ hi2_k, int32_t* __restrict__ flags) {
  const int row = blockIdx.x;
  const float* __restrict__ g = G + (int64_t)row * ld;
  int last1 = -1, last2 = -1;  // last chunk that needs the hi plane, last chunk with an entry >= 65536
  bool bad = false;
  for (int c = threadIdx.x; c < nchunks; c += 256) {
    int mn = 0x7fffffff, mx = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int it = item_of[c * 16 + e];
      if (it >= 0) {
        const float v = g[it];
        const int iv = (int)v;
        if (!(v >= 0.0f && v < 16777216.0f) || (float)iv != v) bad = true;
        if (it != row) {  // (the diagonal is kept apart)
          mn = iv < mn ? iv : mn;
          mx = iv > mx ? iv : mx;
        }
      }
    }
    int b = (use_base && mn != 0x7fffffff) ? (mn >> 4) : 0;
    b = b > 255 ? 255 : b;
    if (mx - 16 * b > 255) last1 = c;
    if (mx >= 65536) last2 = c;
  }
  __shared__ int s1, s2, sb;
  if (threadIdx.x == 0) {
    s1 = -1;
    s2 = -1;
    sb = 0;
  }
  __syncthreads();
  atomicMax(&s1, last1);
  atomicMax(&s2, last2);
  if (bad) atomicOr(&sb, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    constexpr int CG = kPackGroup / 16;  // chunks per group
    hi_k[row] = (s1 + CG) / CG;          // ceil((s1 + 1) / 512); 0 when s1 == -1
    hi2_k[row] = (s2 + CG) / CG;
    if (sb) atomicOr(flags, 1);
  }
}

// Pass 2, one workgroup per row: write the planes (16 ranks per thread and step).
__global__ __launch_bounds__(256) void gram_pack_write(const float* __restrict__ G, int64_t ld, int ncols,
                                                       const int32_t* __restrict__ item_of, int nchunks,
                                                       uint8_t* __restrict__ lo, int64_t ldb,
                                                       uint8_t* __restrict__ hi, const int64_t* __restrict__ hi_off,
                                                       const int32_t* __restrict__ hi_k,
                                                       uint8_t* __restrict__ hi2, const int64_t* __restrict__ hi2_off,
                                                       const int32_t* __restrict__ hi2_k,
                                                       uint8_t* __restrict__ base, float* __restrict__ diag) {
  const int row = blockIdx.x;
  const float* __restrict__ g = G + (int64_t)row * ld;
  const int n1 = hi_k[row] * (kPackGroup / 16), n2 = hi2_k[row] * (kPackGroup / 16);
  uint8_t* __restrict__ plo = lo + (int64_t)row * ldb;
  uint8_t* __restrict__ phi = hi + hi_off[row];
  uint8_t* __restrict__ ph2 = hi2 + hi2_off[row];
  uint8_t* __restrict__ pb = base + (int64_t)row * kPackGroup;
  const int nmax = max(nchunks, max(n1, n2));
  for (int c = threadIdx.x; c < nmax; c += 256) {
    uint32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};
    if (c < nchunks) {
      uint32_t iv[16];
      uint32_t mn = 0xffffffffu;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int it = item_of[c * 16 + e];
        iv[e] = it >= 0 ? (uint32_t)(int)g[it] : 0u;
        if (it == row) diag[row] = g[it];
        if (it >= 0 && it != row) mn = iv[e] < mn ? iv[e] : mn;
      }
      uint32_t b = 0;
      if (c >= n1) {  // behind the prefix: relative to the chunk's base (pass 1 made sure it fits)
        b = mn == 0xffffffffu ? 0u : (mn >> 4);
        b = b > 255u ? 255u : b;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int it = item_of[c * 16 + e];
        const uint32_t v = (it >= 0 && it != row) ? iv[e] - 16u * b : 0u;  // (diagonal: a filler)
        w0[e >> 2] |= (v & 255u) << (8 * (e & 3));
        w1[e >> 2] |= ((v >> 8) & 255u) << (8 * (e & 3));
        w2[e >> 2] |= ((v >> 16) & 255u) << (8 * (e & 3));
      }
      *reinterpret_cast<uint4*>(plo + 16 * (int64_t)c) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
      pb[(c & (kGramrNT - 1)) * 16 + (c / kGramrNT)] = (uint8_t)b;
    }
    if (c < n1) *reinterpret_cast<uint4*>(phi + 16 * (int64_t)c) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
    if (c < n2) *reinterpret_cast<uint4*>(ph2 + 16 * (int64_t)c) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
  }
}

// the row records (after pass 2).  The solver takes |a_i|^2 and its root from the record instead
// of csq / cnorm: flags[0] |= 2 (and the planes are not used) unless G_ii == csq[i] and
// sqrtf(csq[i]) == cnorm[i] bit for bit -- true for the integer-valued matrices that can be packed.
__global__ void gram_pack_meta(int ncols, const int32_t* __restrict__ rank_of, const int32_t* __restrict__ hi_k,
                               const int32_t* __restrict__ hi2_k, const int64_t* __restrict__ hi_off,
                               const float* __restrict__ diag, const int64_t* __restrict__ colptr,
                               const float* __restrict__ csq, const float* __restrict__ cnorm,
                               uint4* __restrict__ meta, int32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncols) {
    if (diag[i] != csq[i] || sqrtf(csq[i]) != cnorm[i]) atomicOr(flags, 2);
    meta[i] = make_uint4((uint32_t)rank_of[i] | ((uint32_t)hi_k[i] << 17) | ((uint32_t)hi2_k[i] << 21),
                         (uint32_t)(hi_off[i] / kPackGroup), (uint32_t)(colptr[i + 1] - colptr[i]),
                         __float_as_uint(csq[i]));
  }
}

#endif  // SLIM_GRAM_PACK_KERNELS

// the 16 floats of one chunk: lo + 256 hi + 65536 hi2, exact (integers below 2^24)
__device__ __forceinline__ void unpack16(const uint4 l, float (&f)[16]) {
  const uint32_t w[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[4 * j + 0] = (float)(w[j] & 255u);
    f[4 * j + 1] = (float)((w[j] >> 8) & 255u);
    f[4 * j + 2] = (float)((w[j] >> 16) & 255u);
    f[4 * j + 3] = (float)(w[j] >> 24);
  }
}
__device__ __forceinline__ void unpack16_add(const uint4 h, const float scale, float (&f)[16]) {
  const uint32_t w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[4 * j + 0] = fmaf(scale, (float)(w[j] & 255u), f[4 * j + 0]);
    f[4 * j + 1] = fmaf(scale, (float)((w[j] >> 8) & 255u), f[4 * j + 1]);
    f[4 * j + 2] = fmaf(scale, (float)((w[j] >> 16) & 255u), f[4 * j + 2]);
    f[4 * j + 3] = fmaf(scale, (float)(w[j] >> 24), f[4 * j + 3]);
  }
}

}  // namespace slimamd
