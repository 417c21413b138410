// cd_gramr.hpp -- item-space coordinate descent with g ON CHIP and G streamed as byte planes.
//
// cd_gram.hpp carries g_i = a_i . r over the items and pays one row of G per update.  Measured on
// the 1M x 100K matrix (profiles/r05/support_overlap_c4.txt): a problem makes ~14 300 updates
// (2 600 coefficients x 5.5 sweeps), the supports of the 32 problems of a tile are as good as
// independent (32 supports of 2.6 % cover 45 % of the items: sharing factor 1.8), and every item is
// in every active set -- so neither sharing rows among problems nor skipping coordinates removes
// work, and the lever that is left is BYTES PER ENTRY OF G and what else an update moves:
//
//   * the row comes as byte planes in popularity order (gram_pack.hpp): ~1.15 bytes per entry
//     instead of 4, decoded exactly, so the fmaf sequence is the float kernel's;
//   * g never leaves the compute unit.  100 000 floats do not fit the LDS (160 KB), but LDS +
//     registers hold them: a workgroup of 512 threads (two wavefronts per SIMD: 256 VGPRs each)
//     owns g in 16-entry chunks, thread t the ranks [16 (t + 512 k), +16) for k = 0 .. K-1 -- the
//     first KR groups in REGISTERS (16 KR VGPRs per lane), the last KL groups in LDS.  An update is
//     then one 16-byte load per thread and group plus FMAs on values the thread already holds: no
//     pass over g in HBM (cd_gram_kernel<8,0>: 0.8 MB per 512-visit batch) and no batch of deferred
//     updates kept current through 512 single-element gathers per update;
//   * a visit needs g_i of ITS coordinate, which sits in some thread's register: before a batch of
//     64 visits every wavefront exports the entries it owns (a wave-uniform register select -- the
//     visit's coordinate is known to all) to a 64-float LDS line, entries of the LDS groups are read
//     in place.  Then the batch runs like cd_gram.hpp's LDS form: every wavefront evaluates the 64
//     visits redundantly, the first lane whose coefficient moves is the next change of the
//     sequential algorithm, its row is streamed and applied by all threads to what they own, and
//     every lane corrects the g of its own visit with ONE entry of that row.
//
// Same update rule (cd.c:121-128), same epsilon rule (cd.c:27), same cap (estimate.c:448-449), same
// stop rule (cd.c:135), same visiting order (cd_perm.hpp over the tile's union list) and -- the
// planes decode exactly -- the same float arithmetic as cd_gram.hpp; checked against the oracle's
// tile walk and against that kernel.
//
// Byte model: per update hi_k(row) and hi2_k(row) groups of 8192 bytes on top of the row's ncols
// bytes of `lo` (counted per problem on the device: SolveArgs.st_B), nothing else of size.
#pragma once
#include <utility>

#include "cd_tile.hpp"
#include "gram_pack.hpp"

#ifndef SLIM_GRAMR_AUX
#define SLIM_GRAMR_AUX 0
#endif
#ifndef SLIM_GRAMR_PROF
#define SLIM_GRAMR_PROF 0
#endif
#ifndef SLIM_GRAMR_FROM_PLANES  // (A/B: 0 = aTy of a problem from the float G, as in round 5)
#define SLIM_GRAMR_FROM_PLANES 1
#endif
// (A/B: 1 = the head of the likely next mover's row is pulled into L2 behind every row's head.  Measured
// round 6, profiles/r06/prefetch_ab.txt: 2.93 s against 2.79 s without -- the row start is not waiting
// for a cold L2 line, and the extra request and the candidate's ballot / readlane cost more than they
// bring.  Off.)
#ifndef SLIM_GRAMR_PREFETCH
#define SLIM_GRAMR_PREFETCH 0
#endif
#ifndef SLIM_GRAMR_LATE_GATHERS  // (A/B: 0 = the gathers in front of the row's head, as in round 5)
#define SLIM_GRAMR_LATE_GATHERS 1
#endif
// bisect switches of scripts/gramr_k13_sweep.py (A/B builds only; all 0 in the product):
#ifndef SLIM_GRAMR_DRAIN     // every ring wait drains everything outstanding
#define SLIM_GRAMR_DRAIN 0
#endif
#ifndef SLIM_GRAMR_SYNCROW   // a workgroup barrier behind every row
#define SLIM_GRAMR_SYNCROW 0
#endif

namespace slimamd {

// one group of this thread's g: 16 consecutive registers, so that entry e of a group can be read
// with a wave-uniform e by ONE v_movrels (register-indirect move) -- a switch over 160 constant
// indices made the register allocator spill a hundred live ranges around it
typedef float gramr_v16 __attribute__((ext_vector_type(16)));

// The groups are members of a struct reached by compile-time indices only (an array indexed by
// the variable of an unrolled loop stays in scratch memory: the promotion to registers runs
// before the loops are unrolled).
template <int N>
struct GramrRegs {
  gramr_v16 v;
  GramrRegs<N - 1> rest;
};
template <>
struct GramrRegs<0> {};
template <int I, int N>
__device__ __forceinline__ gramr_v16& gramr_reg(GramrRegs<N>& r) {
  if constexpr (I == 0) return r.v;
  else return gramr_reg<I - 1>(r.rest);
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// entry e of group k for wave-uniform k, e: a chain of uniform branches down to the group, then one
// register-indirect move.  (The group is also an input of an empty asm: a vector with a single use
// -- "load the group, take element e" -- is folded into a scalar load at a variable address and the
// whole struct stays in memory; with a second use it is the group's registers that are indexed.  The
// first form of this, an in-out operand, cost a copy of the 16 registers per read.)
template <int N, int KRA>
__device__ __forceinline__ float gramr_sel_from(GramrRegs<KRA>& gr, const int k, const int e) {
  if (k == N) {
    const gramr_v16 t = gramr_reg<N>(gr);
    asm volatile("" ::"v"(t));
    return t[e];
  }
  if constexpr (N + 1 < KRA) return gramr_sel_from<N + 1, KRA>(gr, k, e);
  return 0.0f;
}
template <int KRA>
__device__ __forceinline__ float gramr_sel(GramrRegs<KRA>& gr, const int k, const int e) {
  return gramr_sel_from<0, KRA>(gr, k, e);
}

// The export of a batch in ONE loop of 14 instructions per entry, written by hand (round 6).  The
// compiler's form of `gramr_sel` above is a tree of uniform branches down to the group (the
// register-indirect move needs a static base register) plus a lane-masked LDS write per entry: ~40
// instructions with ten branches, 388 cycles per entry at two wavefronts per SIMD
// (profiles/r05/gramr_cycle_profile_c4.txt) -- a quarter of the kernel's cycles with the barrier
// that waits for the slowest wavefront's export.  Here every group is an input operand PINNED to
// fixed registers (group k in v[16 + 16 k : 31 + 16 k]: the allocator keeps the groups there for
// the whole kernel, the ISA is checked for copies by scripts/isa_metadata.sh), so entry (k, e) is
// v[16 + 16 k + e] = ONE indexed move with no branch; the value goes from its owner's lane to an
// SGPR (v_readlane) and from there into lane b of a result register (a move under EXEC = 1 << b:
// v_writelane with an SGPR value AND an SGPR lane select breaks gfx9's one-scalar-operand rule) --
// no branch and no LDS access inside the loop; the caller stores the result register once.
//   mine: the lanes (visits) whose coordinate this wavefront holds in registers; r: the lanes' ranks.
// Wait states (gfx9 ISA, "manually inserted wait states": the assembler adds none inside an asm
// block): the lane selects of both v_readlane are written by SALU instructions (s_ff1, s_bfe), not by
// a VALU -- no wait states; the SGPR written by v_readlane is read by SALU instructions and as a
// VALU source operand (v_writelane's data), both interlocked; M0 is saved and restored around the
// loop (s_set_gpr_idx_on writes it).
#ifndef SLIM_GRAMR_ASMEXPORT
#define SLIM_GRAMR_ASMEXPORT 1
#endif
#define SLIM_GRAMR_EXPORT_ASM                                   \
  "s_mov_b32 %[sm0], m0\n\t"                                   \
  "s_mov_b64 %[sx], exec\n"                                     \
  "1:\n\t"                                                      \
  "s_ff1_i32_b64 %[sb], %[mine]\n\t"                            \
  "v_readlane_b32 %[sr], %[r], %[sb]\n\t"                       \
  "s_bitset0_b64 %[mine], %[sb]\n\t"                            \
  "s_bfe_u32 %[sl], %[sr], 0x60004\n\t"                         \
  "s_lshr_b32 %[sidx], %[sr], 13\n\t"                           \
  "s_and_b32 %[sr], %[sr], 15\n\t"                              \
  "s_lshl4_add_u32 %[sidx], %[sidx], %[sr]\n\t"                 \
  "s_set_gpr_idx_on %[sidx], gpr_idx(SRC0)\n\t"                 \
  "v_mov_b32 %[vt], v16\n\t"                                    \
  "s_set_gpr_idx_off\n\t"                                       \
  "v_readlane_b32 %[sv], %[vt], %[sl]\n\t"                      \
  "s_lshl_b64 exec, 1, %[sb]\n\t"                               \
  "v_mov_b32 %[res], %[sv]\n\t"                                 \
  "s_mov_b64 exec, %[sx]\n\t"                                   \
  "s_cmp_lg_u64 %[mine], 0\n\t"                                 \
  "s_cbranch_scc1 1b\n\t"                                       \
  "s_mov_b32 m0, %[sm0]\n"
#define SLIM_GRAMR_EXPORT_OUT                                                                           \
  [res] "+v"(res), [mine] "+s"(mine), [sb] "=&s"(sb), [sr] "=&s"(sr), [sl] "=&s"(sl), [sidx] "=&s"(sidx), \
      [sv] "=&s"(sv), [vt] "=&v"(vt), [sm0] "=&s"(sm0), [sx] "=&s"(sx)
template <int KRA>
__device__ __forceinline__ float gramr_export(GramrRegs<KRA>& gr, const int r, uint64_t mine) {
  float res = 0.0f, vt;
  int sb, sr, sl, sidx, sm0;
  float sv;
  uint64_t sx;
  static_assert(KRA == 1 || KRA == 3 || KRA == 6 || KRA == 10, "one operand list per instantiation");
  if constexpr (KRA == 1) {
    asm volatile(SLIM_GRAMR_EXPORT_ASM : SLIM_GRAMR_EXPORT_OUT : [r] "v"(r), "{v[16:31]}"(gramr_reg<0>(gr)) : "scc");
  } else if constexpr (KRA == 3) {
    asm volatile(SLIM_GRAMR_EXPORT_ASM
                 : SLIM_GRAMR_EXPORT_OUT
                 : [r] "v"(r), "{v[16:31]}"(gramr_reg<0>(gr)), "{v[32:47]}"(gramr_reg<1>(gr)),
                   "{v[48:63]}"(gramr_reg<2>(gr))
                 : "scc");
  } else if constexpr (KRA == 6) {
    asm volatile(SLIM_GRAMR_EXPORT_ASM
                 : SLIM_GRAMR_EXPORT_OUT
                 : [r] "v"(r), "{v[16:31]}"(gramr_reg<0>(gr)), "{v[32:47]}"(gramr_reg<1>(gr)),
                   "{v[48:63]}"(gramr_reg<2>(gr)), "{v[64:79]}"(gramr_reg<3>(gr)), "{v[80:95]}"(gramr_reg<4>(gr)),
                   "{v[96:111]}"(gramr_reg<5>(gr))
                 : "scc");
  } else {
    asm volatile(SLIM_GRAMR_EXPORT_ASM
                 : SLIM_GRAMR_EXPORT_OUT
                 : [r] "v"(r), "{v[16:31]}"(gramr_reg<0>(gr)), "{v[32:47]}"(gramr_reg<1>(gr)),
                   "{v[48:63]}"(gramr_reg<2>(gr)), "{v[64:79]}"(gramr_reg<3>(gr)), "{v[80:95]}"(gramr_reg<4>(gr)),
                   "{v[96:111]}"(gramr_reg<5>(gr)), "{v[112:127]}"(gramr_reg<6>(gr)),
                   "{v[128:143]}"(gramr_reg<7>(gr)), "{v[144:159]}"(gramr_reg<8>(gr)),
                   "{v[160:175]}"(gramr_reg<9>(gr))
                 : "scc");
  }
  return res;
}

// KR groups of 8192 ranks in registers, KL groups in LDS (dynamic: KL * 32 KB).
// DMA: the row is streamed into a per-wavefront LDS ring by `global_load_lds_dwordx4` (LDS-DMA: no
// VGPR holds data in flight) kGramrAhead groups ahead of the one being decoded, instead of GRP
// register loads per round trip.
// AH = groups requested ahead of the one consumed; the ring has AH + 1 slots of 1 KB per wavefront,
// and one more slot takes the row's base bytes.
constexpr int gramr_ring_bytes(int ah) { return (kGramrNT / 64) * (ah + 2) * 1024; }
// Behind the ring, per wavefront: two buffers of 1280 bytes for the batch headers (x and the row
// record of the 64 items of a batch), filled by LDS-DMA one batch ahead.
constexpr int kGramrHdr = 1280;
// (+ 256 bytes per wavefront: where the prefetch of the likely next row's head lands -- never read)
constexpr int kGramrDump = SLIM_GRAMR_PREFETCH ? 256 : 0;
constexpr int gramr_hdr_bytes() { return (kGramrNT / 64) * (2 * kGramrHdr + kGramrDump); }

template <int KR, int KL, bool DMA = false, int WPS = ((KR <= 2 && KL == 0) ? 4 : 2), int AH = 2>
__global__ __launch_bounds__(kGramrNT, WPS) void cd_gramr_kernel(
    const DevMatrix A, const SolveArgs S, const GramPacked P) {
  constexpr int NT = kGramrNT, K = KR + KL;
  constexpr int kGramrAhead = AH, kGramrSlots = AH + 2;  // (ring + one slot for the base bytes)
  constexpr int KRA = KR > 0 ? KR : 1;
  constexpr int R0 = KR * kPackGroup;  // first rank held in LDS
  static_assert(KR <= 12, "register select covers 12 groups");
  static_assert(KR + KL <= 16, "one base byte per group in a 16-byte load");
  extern __shared__ __attribute__((aligned(16))) float g_lds[];  // [KL][4][NT] float4: conflict-free
  __shared__ float s_gB[2][64];  // the batch's g, double-buffered: one barrier per batch (fetch_g)
  __shared__ int s_p, s_na;
  __shared__ unsigned long long s_D;   // sum of nnz(col i) over the active set: D of one sweep
  __shared__ double s_e2[64], s_reg[64];  // the output pass's sums, per lane of wavefront 0
  __shared__ unsigned long long s_off;
  __shared__ int s_nz;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int ncols = A.ncols;
#if !SLIM_GRAMR_FROM_PLANES
  const int n4 = S.ncols_pad >> 2;
#endif
  const int nchunks = P.nchunks;
  const float l1 = S.l1, l2 = S.l2;
#if !SLIM_GRAMR_FROM_PLANES
  const float* __restrict__ Gm = S.G;  // floats: aTy of a problem (x init, active set)
  const int64_t ld = S.G_ld;
#endif
  float* const x = S.xslab + (int64_t)blockIdx.x * S.x_stride;  // [ncols_pad], item ids, -inf = inactive
#if !SLIM_GRAMR_FROM_PLANES
  float4* const x4 = reinterpret_cast<float4*>(x);
#endif
  float4* const gl4 = reinterpret_cast<float4*>(g_lds);
  const int64_t* __restrict__ colptr = A.colptr;

  GramrRegs<KRA> gr;

  // g += nd * G[row, :] on what this thread owns.  (plo, phi, ph2, hk, h2k): the row's planes.
  // Every load is `global_load_dwordx4 v, v_off, s[base]` with ONE 32-bit offset register per
  // group, clamped to the row's last chunk (threads behind the end of the row read that chunk again
  // and update entries of g no visit ever reads) -- no exec masking, no 64-bit address per group.
  const uint32_t voff0 = 16u * (uint32_t)tid;
  const uint32_t vlast = 16u * (uint32_t)(nchunks - 1);
  // s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#if SLIM_GRAMR_DRAIN
#define SLIM_VMCNT(n) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8))
#else
#define SLIM_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#endif
  char* const ring_w = reinterpret_cast<char*>(g_lds) + KL * kPackGroup * 4 + wave * (kGramrSlots * 1024);
#if SLIM_GRAMR_PROF
  unsigned long long pt_first_v = 0, pt_mark_v = 0, pt_export_v = 0, pt_mark = 0;
  unsigned long long* const pt_first_p = &pt_first_v;
  unsigned long long* const pt_mark_p = &pt_mark_v;
#endif
  char* const hdr_w = reinterpret_cast<char*>(g_lds) + KL * kPackGroup * 4 + (DMA ? gramr_ring_bytes(AH) : 0) +
                      wave * (2 * kGramrHdr + kGramrDump);
  char* const dump_w = hdr_w + 2 * kGramrHdr;
  // (rg: the rank whose entry of this row every lane gathers -- its own visit's; gq: the four bytes)
  auto apply = [&](const uint8_t* __restrict__ plo, const uint8_t* __restrict__ phi,
                   const uint8_t* __restrict__ ph2, const uint8_t* __restrict__ pbase, const int hk,
                   const int h2k, const int cdiag, const int ediag, const float vdiag,
                   const float nd, const int rg, uint32_t (&gq)[4],
                   const uint8_t* __restrict__ plo_next) __attribute__((always_inline)) {
    const bool gin1 = rg < hk * kPackGroup, gin2 = rg < h2k * kPackGroup;
    auto gathers = [&]() __attribute__((always_inline)) {
      // relaxed atomic loads of wavefront scope: ordinary global_load_ubyte in the ISA, but they stay
      // where they are written -- the ring's wait counts below include them
      gq[0] = __hip_atomic_load(plo + rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      gq[1] = __hip_atomic_load(phi + (gin1 ? rg : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      gq[2] = __hip_atomic_load(ph2 + (gin2 ? rg : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      gq[3] = __hip_atomic_load(pbase + (((rg >> 4) & (NT - 1)) * 16 + (rg >> 13)), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    if constexpr (!(DMA && SLIM_GRAMR_LATE_GATHERS)) gathers();
    // the base bytes of this thread's chunks (byte k: chunk tid + 512 k; 0 inside the hi prefix)
    const int kdiag = cdiag / NT, tdiag = cdiag % NT;  // (uniform: the group test is a scalar branch)
    // (with the ring they come by LDS-DMA too, ahead of the row's first group: as a register load
    // the scheduler started on group 0's base byte between the ring's first requests, and the wait it
    // put there drained the requests already made -- an exposed latency per row before the ring was
    // even full)
    uint32_t bwords[4];
    if constexpr (DMA) {
      uint32_t vzb = voff0;
      asm volatile("" : "+v"(vzb));
      __builtin_amdgcn_global_load_lds(pbase + vzb, ring_w + (kGramrAhead + 1) * 1024, 16, 0, 0);
    } else {
      const uint4 bw = ld_off<uint4>(pbase, voff0);
      bwords[0] = bw.x;
      bwords[1] = bw.y;
      bwords[2] = bw.z;
      bwords[3] = bw.w;
    }
    // one group: decoded and added to what the thread owns
    // (the hi chunk is fetched by `hi_of` INSIDE the branch that uses it: with the load in one block
    // and the use in another, the compiler's wait-count state at the join says "maybe pending", and
    // before the next write of those registers it waits for EVERYTHING outstanding -- the whole ring,
    // on every group of every row whether it has a hi chunk or not)
    auto consume = [&](auto kc, const uint4 lo, auto&& hi_of) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      float f[16];
      unpack16(lo, f);
      {
        // (exact: integers below 2^24.  The pairs pass through an empty asm: left alone, the optimizer
        // adds the base in integer and converts afterwards -- 16 byte-select adds + 16 conversions
        // instead of 16 byte conversions + 8 packed float adds)
        typedef float gramr_v2 __attribute__((ext_vector_type(2)));
        const float bf = 16.0f * (float)((bwords[(k >> 2) & 3] >> (8 * (k & 3))) & 255u);
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          gramr_v2 v = {f[e], f[e + 1]};
          asm volatile("" : "+v"(v));
          v += (gramr_v2)(bf);
          f[e] = v.x;
          f[e + 1] = v.y;
        }
      }
      if (k < hk) {
        unpack16_add(hi_of(), 256.0f, f);
        if (k < h2k) {
          uint32_t vz = voff0;
          asm volatile("" : "+v"(vz));
          const uint32_t vo = min(vz + (uint32_t)(kPackGroup * k), vlast);
          unpack16_add(ld_off<uint4>(ph2, vo), 65536.0f, f);
        }
      }
      if (k == kdiag && tid == tdiag) {  // the one chunk of the row that holds its diagonal entry
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = e == ediag ? vdiag : f[e];
      }
      if constexpr (k < KR) {
        gramr_v16& g = gramr_reg<k>(gr);
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = fmaf(nd, f[e], g[e]);
        // (pins the group's FMAs HERE: left alone, the optimizer sinks the FMAs of all groups
        // behind the last `k < hk` join and keeps 16 decoded floats per group alive until then)
        asm volatile("" : "+v"(g));
      } else {
        float4* const gp = gl4 + (k - KR) * 4 * NT + tid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = gp[j * NT];
          v.x = fmaf(nd, f[4 * j + 0], v.x);
          v.y = fmaf(nd, f[4 * j + 1], v.y);
          v.z = fmaf(nd, f[4 * j + 2], v.z);
          v.w = fmaf(nd, f[4 * j + 3], v.w);
          gp[j * NT] = v;
        }
      }
    };
    if constexpr (DMA) {
      // group k's `lo` lands in slot k % S of this wavefront's ring (S = AH + 1 slots of 1 KB);
      // lane L's 16 bytes at L * 16 of the slot -- every thread reads back what it asked for.
      // `hi` is rare behind the row's popular head (C4: half a group per row on average): the hi
      // chunks of the first two groups are register loads issued BEFORE the ring's requests (older
      // than all of them, so waiting for them drains nothing), later ones -- rows of the few very
      // popular items -- are loaded when their group is decoded.
      constexpr int S = kGramrAhead + 1;
      uint4 h01[2];
      static_for<2>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < K) {
          if (k < hk) h01[k] = ld_off<uint4>(phi, min(voff0 + (uint32_t)(kPackGroup * k), vlast));
        }
      });
      // (the byte offset of a request is recomputed from an opaque copy of the thread's offset: left
      // alone the optimizer hoists the thirteen clamped offsets out of the row loop and keeps them,
      // zero-extended to 64 bits for the request's address, in 26 VGPRs -- round 6, seen in the ISA)
      auto request = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        uint32_t vz = voff0;
        asm volatile("" : "+v"(vz));
        const uint32_t vo = min(vz + (uint32_t)(kPackGroup * k), vlast);
        // (aux = SLIM_GRAMR_AUX: 2 = nt, a row is read once by one CU -- MI355X guide, "nt-weights")
        __builtin_amdgcn_global_load_lds(plo + vo, ring_w + (k % S) * 1024, 16, 0, SLIM_GRAMR_AUX);
      };
      static_for<(kGramrAhead < K ? kGramrAhead : K)>([&](auto kc) __attribute__((always_inline)) { request(kc); });
      // (round 6) the lanes' four byte gathers BEHIND the ring's first requests: in front of them they
      // were older than group 0, whose wait then waited for four scattered single-byte loads too
      constexpr int kLate = (SLIM_GRAMR_LATE_GATHERS ? 4 : 0) + (SLIM_GRAMR_PREFETCH ? 1 : 0);
      if constexpr (SLIM_GRAMR_LATE_GATHERS) gathers();
      if constexpr (SLIM_GRAMR_PREFETCH) {
        // (round 6) the first AH groups of the row the NEXT update will most likely stream -- the next
        // pending visit whose coefficient is not 0 -- are pulled into L2 now, in the shadow of this row:
        // what this wavefront will ask for there is AH x 1 KB = AH x 8 lines, one line per lane of one
        // 4-byte LDS-DMA request into a slot nobody reads.  One operation, always issued (no candidate:
        // this row's own head, lines already in flight) -- the wait counts below include it.
        const int ln = lane < 8 * kGramrAhead ? lane : 0;
        const uint32_t po = min((uint32_t)(kPackGroup * (ln >> 3)) + 1024u * (uint32_t)wave + 128u * (uint32_t)(ln & 7), vlast);
        __builtin_amdgcn_global_load_lds(plo_next + po, dump_w, 4, 0, 0);
      }
      static_for<K>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + kGramrAhead < K) request(std::integral_constant<int, k + kGramrAhead>{});
        // loads complete in order: once at most `ahead` requests (the groups behind this one) are
        // outstanding, group k has landed
        constexpr int ahead = (K - 1 - k) < kGramrAhead ? (K - 1 - k) : kGramrAhead;
        // (the late gathers sit between request AH - 1 and request AH: younger than the first AH groups)
        SLIM_VMCNT(ahead + (k < kGramrAhead ? kLate : 0));
        asm volatile("" ::: "memory");
        if constexpr (k == 0) {  // (requested before group 0: landed with it)
          const uint4 bw = *reinterpret_cast<const uint4*>(ring_w + (kGramrAhead + 1) * 1024 + lane * 16);
          bwords[0] = bw.x;
          bwords[1] = bw.y;
          bwords[2] = bw.z;
          bwords[3] = bw.w;
        }
#if SLIM_GRAMR_PROF
        if constexpr (k == 0) *pt_first_p += __builtin_readcyclecounter() - *pt_mark_p;
#endif
        const uint4 lo = *reinterpret_cast<const uint4*>(ring_w + (k % S) * 1024 + lane * 16);
        consume(kc, lo, [&]() __attribute__((always_inline)) -> uint4 {
          if constexpr (k < 2) return h01[k];
          else {
            uint32_t vz = voff0;
            asm volatile("" : "+v"(vz));
            return ld_off<uint4>(phi, min(vz + (uint32_t)(kPackGroup * k), vlast));
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      constexpr int GRP = 4;
      static_for<(K + GRP - 1) / GRP>([&](auto k0c) __attribute__((always_inline)) {
        constexpr int k0 = decltype(k0c)::value * GRP;
        uint4 l[GRP], h[GRP];
        static_for<GRP>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value, k = k0 + u;
          if constexpr (k < K) {
            const uint32_t vo = min(voff0 + (uint32_t)(kPackGroup * k), vlast);
            l[u] = ld_off<uint4>(plo, vo);
            if (k < hk) h[u] = ld_off<uint4>(phi, vo);
          }
        });
        __builtin_amdgcn_sched_barrier(0);  // (the loads of ONE group of GRP chunks in flight, not of all)
        static_for<GRP>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value, k = k0 + u;
          if constexpr (k < K)
            consume(std::integral_constant<int, k>{}, l[u], [&]() __attribute__((always_inline)) -> uint4 { return h[u]; });
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
  };
#undef SLIM_VMCNT
  int nrows_read = 0, nhi16_read = 0;  // rows applied (fold + updates); 16-byte chunks of hi / hi2 planes among them
  // float index in g_lds of rank r >= R0
  auto lds_index = [&](const int r) __attribute__((always_inline)) -> int {
    const int rr = r - R0;
    const int kk = rr >> 13, t = (rr >> 4) & (NT - 1), e = rr & 15;
    return ((kk * 4 + (e >> 2)) * NT + t) * 4 + (e & 3);
  };

  // g of the lanes' coordinates (rank r, wanted where `want`): every wavefront exports the entries
  // ITS threads hold -- in registers (a wave-uniform register select per entry) or in the LDS groups
  // (program order inside a wavefront: no other wavefront's updates are read) -- to a 64-float LDS
  // line.  Called by all threads; ONE barrier: the line is double-buffered, and a buffer written in
  // call n was last read right behind the barrier of call n - 2, i.e. before its readers reached the
  // barrier of call n - 1.
  int gb_par = 0;
  auto fetch_g = [&](const bool want, const int r) __attribute__((always_inline)) -> float {
    float* const gb = s_gB[gb_par];
    gb_par ^= 1;
    const bool in_lds = r >= R0;
    const bool my_wave = (((r >> 4) & (NT - 1)) >> 6) == wave;
    if (KR > 0) {
      uint64_t mine = __ballot(want && !in_lds && my_wave);
#if SLIM_GRAMR_ASMEXPORT
      if (mine) {
        const bool own = want && !in_lds && my_wave;
        const float res = gramr_export<KRA>(gr, r, mine);
        if (own) gb[lane] = res;
      }
#else
      while (mine) {
        const int b = __builtin_ctzll(mine);
        mine &= mine - 1ull;
        const int rb = lane_bcast(r, b);
        const float v = gramr_sel<KRA>(gr, rb >> 13, rb & 15);
        if (lane == ((rb >> 4) & 63)) gb[b] = v;
      }
#endif
    }
    if (KL > 0 && want && in_lds && my_wave) gb[lane] = g_lds[lds_index(r)];
#if SLIM_GRAMR_PROF
    pt_export_v += __builtin_readcyclecounter() - pt_mark;  // (of `fetch`: until the barrier)
#endif
    // (a bare barrier behind a wait for this wavefront's LDS writes: __syncthreads() adds a
    // workgroup fence, and with the header's LDS-DMA requests in flight that fence waits for them
    // -- a memory latency in front of every batch's barrier)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    return gb[lane];
  };

  for (;;) {
    if (tid == 0) {
      s_p = atomicAdd(S.queue, 1);
      s_na = 0;
      s_D = 0;
    }
    __syncthreads();
    const int p = s_p;
    if (p >= S.nwork) break;
    const int item = uni(S.order[p]);
    const int grp = p >> 5;
    const uint32_t gkey = (uint32_t)(grp * S.shard_count + S.shard_index);
    const int* __restrict__ ul = S.ulist + (int64_t)grp * S.u_stride;
    const int nunion = uni(S.tile_nunion[grp]);
#if SLIM_GRAMR_FROM_PLANES
    uint4 irec;
    {
      const uint4 mr = P.meta[item];
      irec = make_uint4(uni(mr.x), uni(mr.y), uni(mr.z), uni(mr.w));
    }
#endif
#if !SLIM_GRAMR_FROM_PLANES
    const float* __restrict__ arow = Gm + (int64_t)item * ld;  // aTy of this problem (floats, item ids)
#endif

#if SLIM_GRAMR_FROM_PLANES
    // -- x = 0 on the active set {i != iC : aTy_i > l1} (estimate.c:433-444), -inf elsewhere.  aTy is
    //    row iC of G: decoded here from the byte planes (the floats the unpacked G holds, exactly),
    //    rank by rank, and stored by item id (item_of: ranks -> ids).  The float G is not read by this
    //    kernel any more (round 6): a handle that can pack G drops its 4 ncols^2 bytes of floats.
    {
      const int hk0 = (int)((irec.x >> 17) & 15u), h2k0 = (int)((irec.x >> 21) & 15u);
      const uint8_t* __restrict__ plo0 = P.lo + (int64_t)item * P.ldb;
      const uint8_t* __restrict__ phi0 = P.hi + (int64_t)irec.y * kPackGroup;
      const uint8_t* __restrict__ ph20 = phi0 + (int64_t)hk0 * kPackGroup;
      const uint8_t* __restrict__ pb0 = P.base + (int64_t)item * kPackGroup;
      int na = 0;
      int64_t dact = 0;
      for (int c = tid; c < nchunks; c += NT) {
        const int kg = c / NT;
        float f[16];
        unpack16(*reinterpret_cast<const uint4*>(plo0 + 16 * (int64_t)c), f);
        const float bf = 16.0f * (float)pb0[(c & (NT - 1)) * 16 + kg];
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] += bf;
        if (kg < hk0) unpack16_add(*reinterpret_cast<const uint4*>(phi0 + 16 * (int64_t)c), 256.0f, f);
        if (kg < h2k0) unpack16_add(*reinterpret_cast<const uint4*>(ph20 + 16 * (int64_t)c), 65536.0f, f);
        const int4* __restrict__ io = reinterpret_cast<const int4*>(P.item_of + 16 * (int64_t)c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int4 q = io[j];
          const int it[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (it[u] >= 0) {
              const bool act = it[u] != item && f[4 * j + u] > l1;   // (the diagonal's filler is never looked at)
              x[it[u]] = act ? 0.0f : kInactive;
              if (act) {
                ++na;
                dact += (int64_t)P.meta[it[u]].z;  // (SURVEY 8(d)'s D: every sweep visits the whole active set)
              }
            }
          }
        }
      }
      na = (int)wave_sum((float)na);  // (< 2^24: exact)
      if (lane == 0 && na) atomicAdd(&s_na, na);
      if (dact) atomicAdd(&s_D, (unsigned long long)dact);
    }
#else
    // -- x = 0 on the active set {i != iC : aTy_i > l1} (estimate.c:433-444), -inf elsewhere
    {
      const float4* __restrict__ a4 = reinterpret_cast<const float4*>(arow);
      int na = 0;
      int64_t dact = 0;
      for (int c = tid; c < n4; c += NT) {
        const float4 a = a4[c];
        const int i0 = c << 2;
        float4 xs;
        const bool a0 = i0 + 0 < ncols && i0 + 0 != item && a.x > l1;
        const bool a1 = i0 + 1 < ncols && i0 + 1 != item && a.y > l1;
        const bool a2 = i0 + 2 < ncols && i0 + 2 != item && a.z > l1;
        const bool a3 = i0 + 3 < ncols && i0 + 3 != item && a.w > l1;
        xs.x = a0 ? 0.0f : kInactive;
        xs.y = a1 ? 0.0f : kInactive;
        xs.z = a2 ? 0.0f : kInactive;
        xs.w = a3 ? 0.0f : kInactive;
        na += (int)a0 + (int)a1 + (int)a2 + (int)a3;
        x4[c] = xs;
        // (SURVEY 8(d)'s D: every sweep visits the whole active set, so D = sweeps x this sum)
        if (a0 | a1 | a2 | a3) {
          const int64_t c0 = colptr[i0], c1 = colptr[i0 + 1];
          const int64_t c2 = i0 + 2 <= ncols ? colptr[i0 + 2] : c1, c3 = i0 + 3 <= ncols ? colptr[i0 + 3] : c2;
          const int64_t c4 = i0 + 4 <= ncols ? colptr[i0 + 4] : c3;
          dact += (a0 ? c1 - c0 : 0) + (a1 ? c2 - c1 : 0) + (a2 ? c3 - c2 : 0) + (a3 ? c4 - c3 : 0);
        }
      }
      na = (int)wave_sum((float)na);  // (< 2^24: exact)
      if (lane == 0 && na) atomicAdd(&s_na, na);
      if (dact) atomicAdd(&s_D, (unsigned long long)dact);
    }
#endif
    // -- warm start (estimate.c:453-464): previous coefficients of the coordinates active now (a
    //    negative value ends up 0 there: the flag-clearing loop resets every x < 0)
    int64_t fe = 0, we = 0;  // entries of the previous column still to fold
    auto warm_x = [&]() __attribute__((always_inline)) {
      if (S.icolptr != nullptr && item < S.incols) {
        __syncthreads();
        fe = uni(S.icolptr[item]);
        we = uni(S.icolptr[item + 1]);
        for (int64_t e = fe + tid; e < we; e += NT) {
          const int k = S.icolind[e];
          if (k < ncols && tile_active(x[k])) {
            const float v = S.icolval[e];
            x[k] = v < 0.0f ? 0.0f : v;
          }
        }
      }
    };
    warm_x();
    // -- g carried over from the previous solve of this problem (a grid step that only moved l2:
    //    same active set, x starts at that solve's model): what the set-up row and the fold below
    //    would recompute -- aTy - sum_j x_j G[j, :] over exactly the kept coefficients (the epsilon
    //    rule of cd.c:27 governs both) -- is what that solve held on chip when it stopped.  Saved and
    //    reloaded in the threads' own layout (16 floats per thread and group), slot `item`.
    // (instantiations of up to six groups -- the grids' shapes; the 13-group kernel keeps its register budget)
    constexpr bool kCarry = K <= kGramrCarryMaxGroups;
    const bool carried = kCarry && S.g_load != nullptr;
    if (carried) {
      const float4* __restrict__ gs = reinterpret_cast<const float4*>(S.g_load + (int64_t)item * S.g_stride);
      static_for<KRA>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        gramr_v16 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 q = gs[(k * NT + tid) * 4 + j];
          v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
        gramr_reg<k>(gr) = v;
      });
#pragma unroll
      for (int k = 0; k < KL; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) gl4[(k * 4 + j) * NT + tid] = gs[((KRA + k) * NT + tid) * 4 + j];
    } else {
    static_for<KRA>([&](auto kc) __attribute__((always_inline)) {
      gramr_reg<decltype(kc)::value>(gr) = (gramr_v16)(0.0f);
    });
#pragma unroll
    for (int k = 0; k < KL; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) gl4[(k * 4 + j) * NT + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    int maxit = 0;
    {
      const int64_t cap = 50 * (uni(colptr[item + 1]) - uni(colptr[item]));  // estimate.c:448-449
      maxit = cap < (int64_t)S.maxniters ? (int)cap : S.maxniters;
    }
    int niters = 0, conv = 0;
    nrows_read = 0;
    nhi16_read = 0;

    // ONE loop carries the whole problem, so that the registers holding g see one site that
    // updates them (apply) and one that reads them out (fetch_g) -- with a site per phase the
    // register allocator splits the 160 live ranges around each and spills hundreds of them.
    //   phase 0  set-up: g = aTy (0 + 1 * G[iC, :] from the planes: the same floats) and the
    //            warm-start fold g -= x_j G[j, :] (cd.c:108-110 in item space)
    //   phase 1  sweeps (cd.c:112-139): a pass = one batch of 64 visits
    //   phase 2  output (estimate.c:477-505): a pass = 64 item ids, ascending
    if (wave == 0) {
      s_e2[lane] = 0.0;
      s_reg[lane] = 0.0;
    }
    int phase = 0;
    bool init_row = !carried;
    if (carried) fe = we;  // (nothing to fold: phase 0 applies no row)
    int t = 0, p0 = 0;      // sweep, first position of the batch
    float dlt = 0.0f;
    PermCtx pc = perm_make(1u, 0u);
    // header of the coming batches, in two stages so that no load waits for another inside a batch:
    // stage A (two batches ahead) the lane's item from the tile's list, stage B (one batch ahead) its
    // x and its row record {rank | hi_k | hi2_k, hi group, nnz, |a|^2}
    int i_a = 0;
    int i_n = 0;
    int hw = 0;  // the header buffer the next header_b fills
    auto header_a = [&](const int q0) __attribute__((always_inline)) {
      const int pos = q0 + lane;
      i_a = pos < nunion ? ul[perm_index(pc, (uint32_t)pos)] : -1;
    };
    // (stage B by LDS-DMA into this wavefront's header buffer: as register loads the record was
    // spilled, and the spill put a wait for the loads just issued in front of the batch's barrier)
    auto header_b = [&]() __attribute__((always_inline)) {
      i_n = i_a < 0 ? 0 : i_a;
      char* const hb = hdr_w + hw * kGramrHdr;
      __builtin_amdgcn_global_load_lds(x + i_n, hb, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(P.meta + i_n, hb + 256, 16, 0, 0);
      hw ^= 1;
    };
    int ib = 0, wpos = 0, nz = 0;  // output pass
    unsigned long long off = 0;
    bool fits = false;
    unsigned long long Uu = 0;  // SURVEY 8(d)'s U: nnz of the columns whose coefficient moved (uniform)
#if SLIM_GRAMR_PROF
    // (a build for scripts/gramr_prof.py only: where a problem's cycles go, reported in place of D / U / bytes / rows)
    unsigned long long pt_fetch = 0, pt_dec = 0, pt_apply = 0;
    pt_first_v = 0;
    pt_export_v = 0;
#define SLIM_PT(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - pt_mark; pt_mark = now_; }
#else
#define SLIM_PT(acc)
#endif

    for (;;) {  // passes
      // -- what the lanes of this pass are about
      bool want = false;
      int r = 0, i = 0;
      uint4 mrow = make_uint4(0u, 0u, 0u, 0u);  // row record of the lane's item: read by readlane at an update
      float xi = kInactive;
      bool part = false, keep = false;
      uint64_t mkeep = 0;
      if (phase == 1) {
        i = i_n;
        // the header of this batch: requested one batch ago, nothing else is outstanding
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // vmcnt(0)
        asm volatile("" ::: "memory");
        {
          const char* const hcur = hdr_w + (hw ^ 1) * kGramrHdr;
          const bool valid = p0 + lane < nunion;
          int lz = lane;  // (through an empty asm: hoisted out of the pass loop, lane * 4 and lane * 16 were spilled)
          asm volatile("" : "+v"(lz));
          const float xv = *reinterpret_cast<const float*>(hcur + lz * 4);
          mrow = *reinterpret_cast<const uint4*>(hcur + 256 + lz * 16);
          r = (int)(mrow.x & 0x1FFFFu);
          xi = valid ? xv : kInactive;
        }
        asm volatile("" ::: "memory");
        part = tile_active(xi);
        header_b();             // (batch p0 + 64: its items came with the previous pass)
        header_a(p0 + 128);
        want = part;
      } else if (phase == 2) {
        i = ib + lane;
        xi = i < ncols ? x[i] : kInactive;
        const bool act = tile_active(xi);
        keep = act && fabsf(xi) > kEps;
        if (act && wave == 0)
          s_reg[lane] += 0.5 * (double)l2 * (double)xi * (double)xi + (double)l1 * (double)fabsf(xi);
        mkeep = __ballot(keep);
        if (mkeep == 0) {  // nothing kept among these 64 ids
          ib += 64;
          if (ib >= ncols) break;
          continue;
        }
        r = keep ? (int)(P.meta[i].x & 0x1FFFFu) : 0;
        want = keep;
      }
#if SLIM_GRAMR_PROF
      pt_mark = __builtin_readcyclecounter();
#endif
      const float g0 = fetch_g(want, r);  // (the one site that reads g out)
      SLIM_PT(pt_fetch)
      float gi = g0;
      uint64_t pend = 0;
      if (phase == 1) {
        pend = __ballot(part);
      }
      // -- the rows this pass applies
      for (;;) {
        int row = 0;
        float nd = 0.0f;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);  // {rank | hi_k << 17 | hi2_k << 21, hi group, hi2 group, G_ii}
        if (phase == 0) {
          bool have = false;
          if (init_row) {
            row = item;
            nd = 1.0f;
            have = true;
          } else {
            while (fe < we) {
              const int kk = uni(S.icolind[fe]);
              ++fe;
              if (kk < ncols) {
                const float xk = uni(x[kk]);
                if (xk > kEps) {
                  row = kk;
                  nd = -xk;
                  have = true;
                  break;
                }
              }
            }
          }
          if (!have) break;
        } else if (phase == 1) {
          if (pend == 0) break;
          const float xeff = (xi > kEps || xi < -kEps) ? xi : 0.0f;
          // |a_i|^2 from the record (the planes are only built when it equals csq[i] and its root
          // equals cnorm[i], gram_pack_meta) -- setup.c:130's rounded norm, squared again (cd.c:127)
          const float sq = __uint_as_float(mrow.w);
          const float cn = sqrtf(sq);
          const float num = cd_num(gi, xeff, sq);
          const float nx = num > l1 ? (num - l1) / cd_den(cn, l2) : 0.0f;
          const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
          const float d = neff - xeff;
          const uint64_t m = __ballot(part && nx != xi) & pend;
          if (m == 0) break;  // nothing else in the batch moves
          const int f = __builtin_ctzll(m);
          const float d_f = lane_bcast(d, f);
          const float nx_f = lane_bcast(nx, f), xi_f = lane_bcast(xi, f);
          dlt = fmaf(nx_f - xi_f, nx_f - xi_f, dlt);
          row = lane_bcast(i, f);
          // (addressed by the uniform `row`: a per-lane &x[i] kept for this store was spilled, and its
          // reload put a scratch latency and a full drain in front of every update of wavefront 0)
          if (wave == 0 && lane == f) x[row] = nx;
          pend = f == 63 ? 0ull : (pend & ~((2ull << f) - 1ull));
          if (d_f == 0.0f) continue;  // (a change below the epsilon of cd.c:27 moves no g)
          nd = -d_f;
          // (the row's record came with the batch header: no dependent load between the decision
          // and the first request of the row)
          rec.x = (uint32_t)lane_bcast((int)mrow.x, f);
          rec.y = (uint32_t)lane_bcast((int)mrow.y, f);
          rec.z = (uint32_t)lane_bcast((int)mrow.z, f);
          rec.w = (uint32_t)lane_bcast((int)mrow.w, f);
          Uu += (unsigned long long)rec.z;
        } else {
          break;
        }
        if (phase == 0) {
          const uint4 mr = P.meta[row];
          rec.x = uni(mr.x);
          rec.y = uni(mr.y);
          rec.z = uni(mr.z);
          rec.w = uni(mr.w);
        }
        const int hk = (int)((rec.x >> 17) & 15u), h2k = (int)((rec.x >> 21) & 15u);
        const uint8_t* __restrict__ plo = P.lo + (int64_t)row * P.ldb;
        const uint8_t* __restrict__ phi = P.hi + (int64_t)rec.y * kPackGroup;
        const uint8_t* __restrict__ ph2 = phi + (int64_t)hk * kPackGroup;  // (one pool: a row's hi2 groups follow its hi groups)
        const uint8_t* __restrict__ pbase = P.base + (int64_t)row * kPackGroup;
        // a visit's lane: the entry of the row its own coordinate needs (four byte loads issued
        // together, ahead of the row; behind a plane's prefix the lane reads byte 0 and drops it)
        const bool in1 = r < hk * kPackGroup, in2 = r < h2k * kPackGroup;
        uint32_t gq[4];
        const int rdiag = (int)(rec.x & 0x1FFFFu);
        SLIM_PT(pt_dec)
#if SLIM_GRAMR_PROF
        pt_mark_v = pt_mark;
#endif
        // the likely next mover: the next pending visit with a coefficient (a zero one rarely moves)
        const uint8_t* __restrict__ plo_next = plo;
        if (SLIM_GRAMR_PREFETCH && phase == 1) {
          const uint64_t cand = __ballot(part && xi != 0.0f) & pend;
          if (cand) plo_next = P.lo + (int64_t)lane_bcast(i, __builtin_ctzll(cand)) * P.ldb;
        }
        apply(plo, phi, ph2, pbase, hk, h2k, rdiag >> 4, rdiag & 15, __uint_as_float(rec.w), nd, r, gq, plo_next);  // (the one site that updates g)
        float gsel = (float)gq[0] + 16.0f * (float)gq[3];
        gsel = in1 ? fmaf(256.0f, (float)gq[1], gsel) : gsel;
        gsel = in2 ? fmaf(65536.0f, (float)gq[2], gsel) : gsel;
        if (init_row) {  // (the byte model counts updates and folds; this row was the set-up)
          init_row = false;
        } else {
          ++nrows_read;
          nhi16_read += min(hk * (kPackGroup / 16), nchunks) + min(h2k * (kPackGroup / 16), nchunks);
        }
        gi = fmaf(nd, gsel, gi);
#if SLIM_GRAMR_SYNCROW
        __syncthreads();
#endif
        SLIM_PT(pt_apply)
      }
      // -- what comes next
      if (phase == 1) {
        p0 += 64;
        if (p0 < nunion) continue;
        if (dlt < S.opt_tol) {  // cd.c:135-138
          conv = 1;
          niters = t + 1;
        } else {
          ++t;
        }
      } else if (phase == 2) {
        if (keep) {
#if SLIM_GRAMR_FROM_PLANES
          float aty = 0.0f;
          if (wave == 0) {  // (row iC's entry at the lane's rank: the four bytes of a gather)
            const int hk0 = (int)((irec.x >> 17) & 15u), h2k0 = (int)((irec.x >> 21) & 15u);
            const uint8_t* __restrict__ plo0 = P.lo + (int64_t)item * P.ldb;
            const uint8_t* __restrict__ phi0 = P.hi + (int64_t)irec.y * kPackGroup;
            const uint8_t* __restrict__ ph20 = phi0 + (int64_t)hk0 * kPackGroup;
            const uint8_t* __restrict__ pb0 = P.base + (int64_t)item * kPackGroup;
            const bool j1 = r < hk0 * kPackGroup, j2 = r < h2k0 * kPackGroup;
            aty = (float)plo0[r] + 16.0f * (float)pb0[((r >> 4) & (NT - 1)) * 16 + (r >> 13)];
            if (j1) aty = fmaf(256.0f, (float)phi0[r], aty);
            if (j2) aty = fmaf(65536.0f, (float)ph20[r], aty);
          }
          if (wave == 0) s_e2[lane] += (double)xi * ((double)aty + (double)g0);
#else
          if (wave == 0) s_e2[lane] += (double)xi * ((double)arow[i] + (double)g0);
#endif
          if (fits && wave == 0) {
            const int64_t dst = (int64_t)off + wpos + __popcll(mkeep & ((1ull << lane) - 1ull));
            S.out_ind[dst] = i;
            S.out_val[dst] = xi;
          }
        }
        wpos += __popcll(mkeep);
        ib += 64;
        if (ib >= ncols) break;
        continue;
      }
      // here: the set-up is done, or a sweep has ended -- start the next sweep or the output
      bool sweep = false;
      if (!conv) {
        if (t >= maxit) {  // loop exhausted without convergence: niters = t + 1 (cd.c:140)
          niters = maxit + 1;
        } else if (nunion == 0) {  // (an empty active set: one sweep that changes nothing)
          conv = 1;
          niters = t + 1;
        } else {
          sweep = true;
        }
      }
      if (sweep) {
        // (the first batch of a sweep may hold coordinates of the LAST batch of the sweep before,
        // whose x wavefront 0 may still be writing: the header loads wait for it.  Inside a sweep
        // the header of the next batch is loaded ahead without one -- its coordinates are others.)
        __syncthreads();
        phase = 1;
        p0 = 0;
        dlt = 0.0f;
        pc = perm_make((uint32_t)nunion, perm_key(S.seed, gkey, (uint32_t)t));
        header_a(0);
        header_b();
        header_a(64);
        continue;
      }
      // wavefront 0 counts the kept coefficients and claims the arena space
      if (wave == 0) {
        int cnt = 0;
        for (int jb = 0; jb < ncols; jb += 64) {
          const int j = jb + lane;
          const float xv = j < ncols ? x[j] : kInactive;
          cnt += __popcll(__ballot(tile_active(xv) && fabsf(xv) > kEps));
        }
        if (lane == 0) {
          s_off = atomicAdd(S.out_cursor, (unsigned long long)cnt);
          s_nz = cnt;
        }
      }
      __syncthreads();
      off = s_off;
      nz = s_nz;
      fits = (int64_t)(off + (unsigned long long)nz) <= S.out_cap;
      phase = 2;
      ib = 0;
      wpos = 0;
      if (ncols <= 0) break;
    }
    // g as this solve leaves it, for the next solve of the problem to start from (see above)
    if (kCarry && S.g_save != nullptr) {
      float4* __restrict__ gs = reinterpret_cast<float4*>(S.g_save + (int64_t)item * S.g_stride);
      static_for<KRA>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const gramr_v16 v = gramr_reg<k>(gr);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gs[(k * NT + tid) * 4 + j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
      });
#pragma unroll
      for (int k = 0; k < KL; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) gs[((KRA + k) * NT + tid) * 4 + j] = gl4[(k * 4 + j) * NT + tid];
    }
    // ||y - A x||^2 = |a_iC|^2 - sum_i x_i (aTy_i + g_i) over the kept coefficients (every
    // wavefront formed the same sums; wavefront 0 reports)
    if (wave == 0) {
      double e2 = s_e2[lane], reg = s_reg[lane];
      for (int o = 32; o > 0; o >>= 1) {
        e2 += __shfl_xor(e2, o);
        reg += __shfl_xor(reg, o);
      }
      if (lane == 0) {
        const float err = (float)(0.5 * ((double)A.csq[item] - e2));
        if (!fits) atomicMax(S.overflow, 1);
        S.out_cnt[item] = fits ? nz : -nz - 1;
        S.out_off[item] = (int64_t)off;
        S.st_na[item] = s_na;
        S.st_sweeps[item] = niters;
        S.st_conv[item] = conv;
#if SLIM_GRAMR_PROF
        S.st_D[item] = (int64_t)pt_fetch;
        S.st_U[item] = (int64_t)pt_dec;
        S.st_B[item] = (int64_t)pt_apply;
        S.st_G[item] = (int)(pt_first_v >> 8);
        S.st_na[item] = (int)(pt_export_v >> 8);
#else
        S.st_D[item] = (int64_t)s_D * (int64_t)(conv ? niters : maxit);  // (sweeps that ran)
        S.st_U[item] = (int64_t)Uu;
        S.st_G[item] = nrows_read;  // (the engine reports the staging pass's G for the column)
        S.st_B[item] = (int64_t)nrows_read * (P.ldb + 16 * (int64_t)(nchunks < NT ? nchunks : NT)) + (int64_t)nhi16_read * 16 +
                       ((carried ? 4 : 0) + (kCarry && S.g_save != nullptr ? 4 : 0)) * S.g_stride;  // (+ g carried in / out)
#endif
        S.st_err[item] = err;
        S.st_obj[item] = err + (float)reg;
      }
    }
    __syncthreads();
  }
}

// Union of the active sets of every tile (32 consecutive entries of the work list), ascending by
// item id: cd_gram.hpp's gram_union_kernel read off the BYTE PLANES (round 6: the floats of G are
// dropped once the planes stand).  One workgroup per tile: every row iC of the tile is decoded rank
// by rank with coalesced 16-byte loads (the floats the unpacked G holds, exactly -- the same set as
// the float kernel's), an active entry sets the bit of its ITEM in an LDS bitmap, and the bitmap is
// written out in ascending order.  3.7 MB of planes per tile on the 100 000-item matrix instead of
// 12.8 MB of floats.
constexpr int kGramrUnionNT = 256;
#ifdef SLIM_GRAM_PACK_KERNELS  // (defined by the one translation unit that owns the non-template kernels)
__global__ __launch_bounds__(kGramrUnionNT) void gramr_union_kernel(const DevMatrix A, const SolveArgs S,
                                                                     const GramPacked P) {
  constexpr int NT = kGramrUnionNT;
  constexpr int kWords = kGramrMaxGroups * kPackGroup / 32;  // items the largest instantiation holds
  __shared__ uint32_t s_bits[kWords];
  __shared__ int s_scan[NT];
  const int tid = threadIdx.x;
  const int grp = blockIdx.x;
  const int base = grp * 32;
  const int nprob = (S.nwork - base) < 32 ? (S.nwork - base) : 32;
  const int ncols = A.ncols;
  const int nwords = (ncols + 31) >> 5;
  const int nchunks = P.nchunks;
  int* __restrict__ ul = S.ulist + (int64_t)grp * S.u_stride;
  for (int w = tid; w < nwords; w += NT) s_bits[w] = 0u;
  __syncthreads();
  for (int q = 0; q < nprob; ++q) {
    const int it0 = uni(S.order[base + q]);
    const uint4 mr = P.meta[it0];
    const uint32_t rx = uni(mr.x), ry = uni(mr.y);
    const int hk = (int)((rx >> 17) & 15u), h2k = (int)((rx >> 21) & 15u);
    const uint8_t* __restrict__ plo = P.lo + (int64_t)it0 * P.ldb;
    const uint8_t* __restrict__ phi = P.hi + (int64_t)ry * kPackGroup;
    const uint8_t* __restrict__ ph2 = phi + (int64_t)hk * kPackGroup;
    const uint8_t* __restrict__ pb = P.base + (int64_t)it0 * kPackGroup;
    for (int c = tid; c < nchunks; c += NT) {
      const int kg = c / kGramrNT;
      float f[16];
      unpack16(*reinterpret_cast<const uint4*>(plo + 16 * (int64_t)c), f);
      const float bf = 16.0f * (float)pb[(c & (kGramrNT - 1)) * 16 + kg];
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] += bf;
      if (kg < hk) unpack16_add(*reinterpret_cast<const uint4*>(phi + 16 * (int64_t)c), 256.0f, f);
      if (kg < h2k) unpack16_add(*reinterpret_cast<const uint4*>(ph2 + 16 * (int64_t)c), 65536.0f, f);
      const int4* __restrict__ io = reinterpret_cast<const int4*>(P.item_of + 16 * (int64_t)c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int4 qv = io[j];
        const int it[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (it[u] >= 0 && it[u] != it0 && f[4 * j + u] > S.l1) atomicOr(&s_bits[it[u] >> 5], 1u << (it[u] & 31));
      }
    }
  }
  __syncthreads();
  // ascending: thread t owns the words [t W, (t + 1) W)
  const int W = (nwords + NT - 1) / NT;
  const int w0 = tid * W, w1 = min(nwords, w0 + W);
  int cnt = 0;
  for (int w = w0; w < w1; ++w) cnt += __popc(s_bits[w]);
  s_scan[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < NT; o <<= 1) {  // (inclusive scan over the 256 counts)
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int pos = s_scan[tid] - cnt;
  for (int w = w0; w < w1; ++w) {
    uint32_t b = s_bits[w];
    while (b) {
      const int j = __builtin_ctz(b);
      b &= b - 1u;
      ul[pos++] = (w << 5) + j;
    }
  }
  if (tid == NT - 1) S.tile_nunion[grp] = s_scan[tid];
}
#endif  // SLIM_GRAM_PACK_KERNELS

}  // namespace slimamd
