// STATUS (round 6): AN EXPERIMENT, NOT PART OF libslim.so.  Measured on the 1M x 100K matrix (8192
// columns; cd_gramr_kernel<10,3>: 2.78 s):
//   * rows chained inside a batch only (-DSLIM_GRAMRP_NOCHAINT: a batch ends with nothing in
//     flight): models bit-equal to the float kernel, 4.46 s -- the barrier + export at every ROW
//     boundary (14 400 per problem instead of 8 600 per batch), the second set of gathers and 280
//     SGPR spills cost far more than the row starts they hide;
//   * batches ending with a row in flight (the replay below): wrong models -- the replayed value
//     does not equal what the registers hold one boundary later although ranks, gathers and the
//     barrier count check out (-DSLIM_GRAMRP_CHECK); not found in the time given.
// What the attempt taught, and what cd_gramr.hpp took from it: the hand-written export loop, the
// explicit arithmetic of the update rule (cd_wave.hpp: cd_num), offsets recomputed per request
// instead of 26 VGPRs of hoisted ones, and three facts about the compiler recorded in DESIGN 4.2e
// (LDS reads behind an LDS-DMA, loads carried across a back edge, stores pending beside loads).
// Build it with: hipcc ... -I.. experimental/gramrp_inst.hip (gramr_inst.hpp no longer declares it).
//
// cd_gramrp.hpp -- cd_gramr.hpp's descent with the DECISIONS ONE ROW AHEAD OF THE STREAM (round 6).
//
// cd_gramr_kernel alternates: decide the next mover (every wavefront, redundantly) -> request its row
// of G -> wait a memory latency for the first group -> stream 13 groups -> decide ...; and once per
// batch of 64 visits: export the batch's g out of registers -> barrier -> read.  The cycle profile of
// wavefront 0 on the 1M x 100K matrix (profiles/r06/gramr_cycle_profile_c4_asmexport.txt): 60 %
// streaming, 15 % waiting for a row's first group, 16 % export + barrier, 9 % deciding -- and all
// eight wavefronts of the one workgroup a compute unit holds are in the same phase at the same time,
// so nothing fills the gaps.  Here a wavefront never stops streaming inside a sweep:
//
//   * the next mover is decided WHILE the current row streams.  What the decision needs of the
//     current row is one entry per lane (the row's effect on the g of the lane's own visit); that
//     byte is gathered with the row's head, long before the row's last group arrives.  The head of
//     the next row (gathers, base bytes, first ring requests) is issued in the shadow of the
//     current row's last groups: the ring never drains between rows.
//   * across batches the same: the g of batch n+1's visits is exported at every row boundary of
//     batch n (registers are consistent there, the next row's first groups are already in flight
//     while the export runs), and the one row decided since -- the row in flight -- is replayed on
//     the exported value with its own gathered entry: fmaf in the order the registers see, so the
//     value is the float the register will hold.  Every head therefore gathers for the lanes of
//     two batches.
//   * one barrier per ROW boundary (not per batch), with loads in flight across it.
//   * every wait is a hand-counted `s_waitcnt vmcnt(n)`: loads and stores complete in order on
//     gfx9, n = the operations issued behind the group that is awaited.  Everything that is counted
//     is issued unconditionally (a plane chunk that is not needed is requested from a line that
//     stays in the caches); an operation the count does not know (a compiler spill, a rare
//     epsilon-sized update's store) only makes a wait longer, never shorter.
//
// Same update rule (cd.c:121-128), epsilon rule (cd.c:27), cap (estimate.c:448-449), stop rule
// (cd.c:135), visiting order (cd_perm.hpp) and fmaf sequence as cd_gramr.hpp / cd_gram.hpp: the
// models are bit-identical (tests/test_gpu_parity.py, scripts/gramr_k13_sweep.py).
//
// g: KR groups of 8192 ranks in registers, KLF full groups in LDS, and a TAIL group of TLT threads x 16
// ranks (the 1M x 100K matrix: 100 000 = 12 x 8192 + 1696 ranks -- a full thirteenth group would
// leave 24 KB of LDS empty, which the three header buffers per wavefront below need).
#pragma once
#include "cd_gramr.hpp"

namespace slimamd {

// (the host pass of the compiler parses the kernel too and rejects the 16-byte form of the LDS-DMA
// builtin outside a gfx950 device compilation)
#if defined(__HIP_DEVICE_COMPILE__)
#define SLIM_LDS_DMA(src, dst, size, off, aux) __builtin_amdgcn_global_load_lds(src, dst, size, off, aux)
#else
#define SLIM_LDS_DMA(src, dst, size, off, aux) ((void)0)
#endif

// per wavefront behind the ring: three header buffers {x[64], record[64], item[64]} (the decision
// batch, the next one -- its ranks are what gets exported and gathered for -- and the one in
// flight) and one item buffer (the tile's list entries, requested one batch before their header)
constexpr int kGramrpHdr = 1536;
constexpr int kGramrpHdrWave = 3 * kGramrpHdr + 256;
// per wavefront: AH + 1 ring slots of 1 KB and TWO slots for base bytes (the next row's arrive while
// the current row's last groups still read theirs: in registers they cost four VGPRs the stream
// does not have)
constexpr int gramrp_ring_wave(int ah) { return (ah + 3) * 1024; }
constexpr int gramrp_lds_bytes(int klf, int tt, int ah) {
  return klf * kPackGroup * 4 + tt * 64 + (kGramrNT / 64) * (gramrp_ring_wave(ah) + kGramrpHdrWave);
}

constexpr int gramrp_group_at(int p, int ah) { return p < ah ? p + 1 : (p == ah ? 0 : p); }

template <int KR, int KLF, int TLT, int AH>
__global__ __launch_bounds__(kGramrNT, 2) void cd_gramrp_kernel(const DevMatrix A, const SolveArgs S,
                                                                const GramPacked P) {
  constexpr int NT = kGramrNT;
  constexpr int KL = KLF + (TLT > 0 ? 1 : 0), K = KR + KL, NS = AH + 1;
  constexpr int CP = K - 1 - AH;  // the last group behind which the row still has requests to make
  constexpr int R0 = KR * kPackGroup;
  constexpr int GLDS = KLF * kPackGroup * 4 + TLT * 64;  // bytes of g in LDS
  static_assert(KR == 1 || KR == 3 || KR == 6 || KR == 10, "gramr_export has one operand list per KR");
  static_assert(CP + 1 >= KR, "the register groups are updated by the first part of a row only");
  static_assert(K > AH + 1 && K <= 16 && TLT % 64 == 0 && TLT < NT, "geometry");
  extern __shared__ __attribute__((aligned(16))) float g_lds[];
  __shared__ float s_gB[2][64];  // the exported g of 64 visits, double-buffered (one barrier per post)
  __shared__ int s_p, s_na;
  __shared__ unsigned long long s_D;
  __shared__ unsigned long long s_off;
  __shared__ int s_nz;
  // The coefficients that moved since the last point with nothing in flight: {item, value}, written to
  // x in HBM there (flush_x).  No store is issued while rows stream: on gfx9 a store may complete out
  // of order with the loads around it, so it can neither be counted in a `vmcnt` wait nor left out
  // without cost -- and the compiler, seeing loads and stores pending together, answers every wait of
  // its own with vmcnt(0).  Nothing reads x inside a sweep but the headers of batches still to come,
  // whose coordinates moved last in the sweep before.
  constexpr int kUpd = 128;
  __shared__ int s_urow[kUpd];
  __shared__ float s_unx[kUpd];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int ncols = A.ncols;
  const int n4 = S.ncols_pad >> 2;
  const int nchunks = P.nchunks;
  const float l1 = S.l1, l2 = S.l2;
  const float* __restrict__ Gm = S.G;
  const int64_t ld = S.G_ld;
  float* const x_slab = S.xslab + (int64_t)blockIdx.x * S.x_stride;
  float4* const gl4 = reinterpret_cast<float4*>(g_lds);
  const int64_t* __restrict__ colptr = A.colptr;
  char* const lds0 = reinterpret_cast<char*>(g_lds);
  char* const ring_w = lds0 + GLDS + wave * gramrp_ring_wave(AH);  // NS slots + two for base bytes
  char* const hdr_w = lds0 + GLDS + (NT / 64) * gramrp_ring_wave(AH) + wave * kGramrpHdrWave;
  char* const itm_w = hdr_w + 3 * kGramrpHdr;
  // the output pass's sums, per lane of wavefront 0: in wavefront 0's ring slots (nothing streams then)
  double* const s_e2 = reinterpret_cast<double*>(lds0 + GLDS);
  double* const s_reg = s_e2 + 64;

  GramrRegs<KR> gr;

  // Everything below runs inside ONE lambda whose parameters are __restrict__: the pointers every
  // LDS-DMA request reads from.  Inlined, each request carries the alias scope of its source and every
  // other access of the body -- the LDS reads in particular -- is marked as not aliasing those scopes.
  // That is what keeps the compiler from putting `s_waitcnt vmcnt(0)` in front of EVERY LDS read while
  // an LDS-DMA is in flight (it cannot see which LDS bytes a request writes and assumes all of them;
  // round 5's kernel had this property by accident -- its `apply` lambda took restrict parameters --
  // and loses its ring without it: seen in the ISA of the first build of this file).  The ring's own
  // reads are ordered behind their requests by the hand-counted waits.
  auto body = [&](const uint8_t* __restrict__ LO, const uint8_t* __restrict__ HI, const uint8_t* __restrict__ BS,
                  const uint4* __restrict__ META, float* __restrict__ x,
                  const int32_t* __restrict__ UL0) __attribute__((always_inline)) {
  float4* const x4 = reinterpret_cast<float4*>(x);
  const uint32_t voff0 = 16u * (uint32_t)tid;
  const uint32_t vlast = 16u * (uint32_t)(nchunks - 1);
#if defined(SLIM_GRAMRP_DRAIN)  // (bisect build: every wait drains)
#define SLIM_VMCNT(n) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8))
#else
#define SLIM_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#endif

  // float index in g_lds of rank r >= R0
  auto lds_index = [&](const int r) __attribute__((always_inline)) -> int {
    const int rr = r - R0;
    const int kk = rr >> 13, t = (rr >> 4) & (NT - 1), e = rr & 15;
    if (TLT > 0 && kk >= KLF) return ((KLF * 4 * NT + (e >> 2) * TLT + t) * 4) + (e & 3);
    return ((kk * 4 + (e >> 2)) * NT + t) * 4 + (e & 3);
  };

  // ---- a row of G as the stream sees it (all wave-uniform)
  struct Row {
    bool valid;
    int row;
    float nd;
    const uint8_t* plo;
    const uint8_t* phi;
    const uint8_t* ph2;
    int hk, h2k, kdiag, tdiag, ediag;
    float vdiag;
    int base;  // ring slot of its group 0
    int bs;    // which of the two base-byte slots
  };
  auto make_row = [&](const int row, const float nd, const uint32_t rx, const uint32_t ry, const uint32_t rw,
                      const int base, const int bs) __attribute__((always_inline)) -> Row {
    Row q;
    q.valid = true;
    q.row = row;
    q.nd = nd;
    q.hk = (int)((rx >> 17) & 15u);
    q.h2k = (int)((rx >> 21) & 15u);
    q.plo = LO + (int64_t)row * P.ldb;
    q.phi = HI + (int64_t)ry * kPackGroup;
    q.ph2 = q.phi + (int64_t)q.hk * kPackGroup;
    const int rdiag = (int)(rx & 0x1FFFFu);
    const int cdiag = rdiag >> 4;
    q.kdiag = cdiag / NT;
    q.tdiag = cdiag % NT;
    q.ediag = rdiag & 15;
    q.vdiag = __uint_as_float(rw);
    q.base = base;
    q.bs = bs;
    return q;
  };
  auto slot_of = [&](const int base, const int k) __attribute__((always_inline)) -> int {
    int s = base + (k % NS);
    return s >= NS ? s - NS : s;
  };

  // Loads and stores that the wait counts below include must stay where they are written: a relaxed
  // atomic access of wavefront scope is an ordinary global_load / global_store in the ISA (no cache
  // bypass), but neither the optimizer nor the scheduler moves it across the waits (a plain load
  // used much later may be sunk to its use -- behind a wait that counted it).  (Loads written in
  // assembly were tried and are wrong: the compiler takes an asm's output as ready and copies it into
  // the loop-carried register before the data has landed.)
  // A gathered byte comes as the aligned dword that holds it -- planes and rows are 16-byte aligned
  // -- and is picked out of it where it is used: a byte load's zero extension would be scheduled
  // right behind the load and wait for it.
  auto ld_w = [](const uint8_t* base, const uint32_t off) __attribute__((always_inline)) -> uint32_t {
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(base + (off & ~3u)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WAVEFRONT);
  };
  auto ld_u64 = [](const uint8_t* base, const uint32_t off) __attribute__((always_inline)) -> uint64_t {
    return __hip_atomic_load(reinterpret_cast<const uint64_t*>(base + off), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WAVEFRONT);
  };
  // the gathered bytes of a head: the row's entry at the lanes' ranks, for two batches of lanes
  uint32_t gb0 = 0, gb1 = 0, gb3 = 0, gc0 = 0, gc1 = 0, gc3 = 0;
  uint64_t h0a = 0, h0b = 0;  // hi chunk of group 0 (the one most rows have)
  auto gsel_of = [&](const Row& q, const int r, uint32_t w0, uint32_t w1, uint32_t w3) __attribute__((always_inline)) -> float {
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w3));  // (nothing of this moves up to the loads)
    const bool in1 = r < q.hk * kPackGroup, in2 = r < q.h2k * kPackGroup;
    const uint32_t b0 = (w0 >> (8 * (r & 3))) & 255u;
    const uint32_t b1 = (w1 >> (8 * (in1 ? r & 3 : 0))) & 255u;
    const uint32_t b3 = (w3 >> (8 * ((r >> 13) & 3))) & 255u;
    float g = (float)b0 + 16.0f * (float)b3;
    g = in1 ? fmaf(256.0f, (float)b1, g) : g;
    // (bits 16-23: only the rows of the ~160 most popular items have them, and a row of those is
    // rarely a mover's -- fetched here, with the compiler's own wait, instead of two more gathers
    // held in registers through every row)
    if (q.h2k > 0) {
      const uint32_t b2 = q.ph2[in2 ? r : 0];
      g = in2 ? fmaf(65536.0f, (float)b2, g) : g;
    }
    return g;
  };
  // The register loads of a row -- six byte gathers (the row's entry at the lanes' ranks, for two
  // batches of lanes) and the hi chunk of group 0 (two 8-byte loads) -- are issued at the START OF THE
  // ROW'S OWN TURN, behind the first AH ring requests (which the previous turn made) and in front of
  // the rest, and used later in the same turn: the waits in between are written out, so the compiler
  // knows they have landed.  (Issued with the head in the previous turn they were correct but useless:
  // across the loop's back edge the compiler does not trust in-order completion while an LDS-DMA is
  // pending and drains everything in front of their first use.)  kRowLoads = 8 operations, always
  // (what a row does not need comes from the first bytes of a plane: lines that stay cached).
  constexpr int kRowLoads = 8;
  auto row_loads = [&](const Row& q, const int rD, const int rN) __attribute__((always_inline)) {
    const uint8_t* __restrict__ pbase = BS + (int64_t)q.row * kPackGroup;
    {
      const bool in1 = rD < q.hk * kPackGroup;
      gb0 = ld_w(q.plo, (uint32_t)rD);
      gb1 = ld_w(q.phi, in1 ? (uint32_t)rD : 0u);
      gb3 = ld_w(pbase, (uint32_t)(((rD >> 4) & (NT - 1)) * 16 + (rD >> 13)));
    }
    {
      const bool in1 = rN < q.hk * kPackGroup;
      gc0 = ld_w(q.plo, (uint32_t)rN);
      gc1 = ld_w(q.phi, in1 ? (uint32_t)rN : 0u);
      gc3 = ld_w(pbase, (uint32_t)(((rN >> 4) & (NT - 1)) * 16 + (rN >> 13)));
    }
    {
      const uint32_t ho = 0 < q.hk ? min(voff0, vlast) : 16u * (uint32_t)lane;
      h0a = ld_u64(q.phi, ho);
      h0b = ld_u64(q.phi, ho + 8u);
    }
  };
  // the head of a row, made in the shadow of the row before it: the base bytes, then (request) the
  // first AH groups of the stream -- kHeadOps = 1 operation besides the ring's
  constexpr int kHeadOps = 1;
  auto head_base = [&](const Row& q) __attribute__((always_inline)) {
    const uint8_t* __restrict__ pbase = BS + (int64_t)q.row * kPackGroup;
    uint32_t vz = voff0;
    asm volatile("" : "+v"(vz));
    SLIM_LDS_DMA(pbase + vz, ring_w + (NS + q.bs) * 1024, 16, 0, 0);
  };
  // A row streams its groups in the order 1, 2, .., AH, 0, AH + 1, .. K - 1: group 0 is the one with a
  // hi chunk in most rows, and that chunk (a register load of this turn) has landed once the ring has
  // turned over -- position p holds group kGroupAt(p).
  // (the byte offset of a request is recomputed from an opaque copy of the thread's offset: left
  // alone the optimizer hoists the thirteen clamped offsets out of the loops and keeps them, zero-
  // extended to 64 bits for the request's address, in 26 VGPRs -- the stream's whole spill budget)
  auto request = [&](const Row& q, const int p) __attribute__((always_inline)) {  // the group at position p
    const int k = gramrp_group_at(p, AH);
    uint32_t vz = voff0;
    asm volatile("" : "+v"(vz));
    const uint32_t vo = min(vz + (uint32_t)(kPackGroup * k), vlast);
    // (aux = SLIM_GRAMR_AUX: 2 = nt, a row is read once by one CU -- MI355X guide, "nt-weights")
    SLIM_LDS_DMA(q.plo + vo, ring_w + slot_of(q.base, p) * 1024, 16, 0, SLIM_GRAMR_AUX);
  };
  // group k of row q: read from its ring slot, decoded, added to what this thread owns
  auto consume = [&](const Row& q, auto pc_) __attribute__((always_inline)) {
    constexpr int pos = decltype(pc_)::value, k = gramrp_group_at(pos, AH);
    const uint4 lo = *reinterpret_cast<const uint4*>(ring_w + slot_of(q.base, pos) * 1024 + lane * 16);
    float f[16];
    unpack16(lo, f);
    {
      typedef float gramr_v2 __attribute__((ext_vector_type(2)));
      // the chunk's base byte: byte k of this thread's 16 in the row's base slot
      const uint32_t bw = *reinterpret_cast<const uint32_t*>(ring_w + (NS + q.bs) * 1024 + lane * 16 + (k >> 2) * 4);
      const float bf = 16.0f * (float)((bw >> (8 * (k & 3))) & 255u);
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        gramr_v2 v = {f[e], f[e + 1]};
        asm volatile("" : "+v"(v));
        v += (gramr_v2)(bf);
        f[e] = v.x;
        f[e + 1] = v.y;
      }
    }
    if (k < q.hk) {
      // (group 0's chunk came with the head; later groups -- rows of the few very popular items --
      // are fetched here: the compiler's own wait drains the ring for them)
      uint32_t vz = voff0;
      asm volatile("" : "+v"(vz));  // (see request(): not hoisted, not kept)
      const uint32_t vo = min(vz + (uint32_t)(kPackGroup * k), vlast);
      if constexpr (k == 0) {
        uint64_t a = h0a, b = h0b;
        asm volatile("" : "+v"(a), "+v"(b));
        unpack16_add(make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)), 256.0f, f);
      }
      else unpack16_add(ld_off<uint4>(q.phi, vo), 256.0f, f);
      if (k < q.h2k) unpack16_add(ld_off<uint4>(q.ph2, vo), 65536.0f, f);
    }
    if (k == q.kdiag && tid == q.tdiag) {
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] = e == q.ediag ? q.vdiag : f[e];
    }
    const float nd = q.nd;
    if constexpr (k < KR) {
      gramr_v16& g = gramr_reg<k>(gr);
#pragma unroll
      for (int e = 0; e < 16; ++e) g[e] = fmaf(nd, f[e], g[e]);
      asm volatile("" : "+v"(g));
    } else if constexpr (TLT > 0 && k == K - 1) {
      if (tid < TLT) {
        float4* const gp = gl4 + KLF * 4 * NT + tid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = gp[j * TLT];
          v.x = fmaf(nd, f[4 * j + 0], v.x);
          v.y = fmaf(nd, f[4 * j + 1], v.y);
          v.z = fmaf(nd, f[4 * j + 2], v.z);
          v.w = fmaf(nd, f[4 * j + 3], v.w);
          gp[j * TLT] = v;
        }
      }
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

  // ---- the export of 64 entries of g (ONE site): the registers' part by gramr_export, the LDS
  //      groups in place; every wavefront writes what its own threads hold.  The caller's barrier
  //      publishes the line.
#if defined(SLIM_GRAMRP_CHECK)
  __shared__ int s_turn[8];
  int my_posts = 0, misaligned = 0;
#endif
  int par = 0;  // the line buffer the next post writes
  auto post = [&](const bool want, const int r) __attribute__((always_inline)) {
    float* const gb = s_gB[par];
    par ^= 1;
    const bool in_lds = r >= R0;
    const bool my_wave = (((r >> 4) & (NT - 1)) >> 6) == wave;
    {
      const bool own = want && !in_lds && my_wave;
      uint64_t mine = __ballot(own);
      if (mine) {
        const float res = gramr_export<KR>(gr, r, mine);
        if (own) gb[lane] = res;
      }
    }
    if (KL > 0 && want && in_lds && my_wave) gb[lane] = g_lds[lds_index(r)];
#if defined(SLIM_GRAMRP_CHECK)
    ++my_posts;
    if (lane == 0) s_turn[wave] = my_posts;
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));  // lgkmcnt(0): this wavefront's LDS writes
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#if defined(SLIM_GRAMRP_CHECK)
    if (s_turn[lane & 7] != my_posts) misaligned = 1;
#endif
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
    const int* __restrict__ ul = UL0 + (int64_t)grp * S.u_stride;
    const int nunion = uni(S.tile_nunion[grp]);
    const float* __restrict__ arow = Gm + (int64_t)item * ld;

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
        if (a0 | a1 | a2 | a3) {
          const int64_t c0 = colptr[i0], c1 = colptr[i0 + 1];
          const int64_t c2 = i0 + 2 <= ncols ? colptr[i0 + 2] : c1, c3 = i0 + 3 <= ncols ? colptr[i0 + 3] : c2;
          const int64_t c4 = i0 + 4 <= ncols ? colptr[i0 + 4] : c3;
          dact += (a0 ? c1 - c0 : 0) + (a1 ? c2 - c1 : 0) + (a2 ? c3 - c2 : 0) + (a3 ? c4 - c3 : 0);
        }
      }
      na = (int)wave_sum((float)na);
      if (lane == 0 && na) atomicAdd(&s_na, na);
      if (dact) atomicAdd(&s_D, (unsigned long long)dact);
    }
    // -- warm start (estimate.c:453-464)
    int64_t fe = 0, we = 0;
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
    static_for<KR>([&](auto kc) __attribute__((always_inline)) {
      gramr_reg<decltype(kc)::value>(gr) = (gramr_v16)(0.0f);
    });
#pragma unroll
    for (int k = 0; k < KLF; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) gl4[(k * 4 + j) * NT + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TLT > 0 && tid < TLT)
#pragma unroll
      for (int j = 0; j < 4; ++j) gl4[KLF * 4 * NT + j * TLT + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    int maxit = 0;
    {
      const int64_t cap = 50 * (uni(colptr[item + 1]) - uni(colptr[item]));  // estimate.c:448-449
      maxit = cap < (int64_t)S.maxniters ? (int)cap : S.maxniters;
    }
    int niters = 0, conv = 0;
    int nrows_read = 0, nhi16_read = 0;
    unsigned long long Uu = 0;

    // ---- state of the descent
    //   phase 0  set-up rows (g = aTy from the planes, then the warm-start fold)
    //   phase 1  sweeps; 3 = a sweep is about to start (its headers are loaded with nothing in flight)
    //   phase 2  output; 4 = the output is about to start
    int phase = 0;
    bool init_row = true, first_row = true;
    int t = 0, p0 = 0;
    float dlt = 0.0f;
    PermCtx pc = perm_make(1u, 0u);
    // lanes of the decision batch (first position p0) ...
    int i = 0, r = 0;
    float xi = kInactive, gi = 0.0f, sq = 0.0f;
    bool part = false, have_gi = false, posted_next = false;
    uint64_t pend = 0;
    // ... and of the batch behind it: its ranks are exported and gathered for
    int rN = 0;
    bool wantN = false;
    int hb = 0;  // header buffer of the decision batch (the next batch's: hb + 1, in flight: hb + 2; mod 3)
    // output pass
    int ib = 0, wpos = 0, nz = 0;
    unsigned long long off = 0;
    bool fits = false, keep = false, out_posted = false;
    uint64_t mkeep = 0;
    int ring_base = 0;

    auto hdr_buf = [&](const int j) __attribute__((always_inline)) -> char* {  // j = 0, 1, 2 behind hb
      int b = hb + j;
      b = b >= 3 ? b - 3 : b;
      return hdr_w + b * kGramrpHdr;
    };
    // items of the batch at q0 -> item buffer `which` (LDS-DMA; a lane behind the list asks for entry 0)
    auto request_items = [&](const int q0) __attribute__((always_inline)) {
      const int pos = q0 + lane;
      const int idx = pos < nunion ? (int)perm_index(pc, (uint32_t)pos) : 0;
      SLIM_LDS_DMA(ul + idx, itm_w, 4, 0, 0);
    };
    // x and row record of the items in the item buffer -> header buffer j (two LDS-DMA requests; the
    // item ids are copied beside them -- the item buffer is free again: its next request follows)
    auto request_header = [&](const int j) __attribute__((always_inline)) {
      int lz = lane;
      asm volatile("" : "+v"(lz));
      const int it = *reinterpret_cast<const int*>(itm_w + lz * 4);
      char* const hbp = hdr_buf(j);
      SLIM_LDS_DMA(x + it, hbp, 4, 0, 0);
      SLIM_LDS_DMA(META + it, hbp + 256, 16, 0, 0);
      *reinterpret_cast<int*>(hbp + 1280 + lz * 4) = it;
    };
    // the lanes of the decision batch out of header buffer 0, the ranks of the next batch out of buffer 1
    auto load_lanes = [&]() __attribute__((always_inline)) {
      int lz = lane;
      asm volatile("" : "+v"(lz));
      const char* const h0p = hdr_buf(0);
      const char* const h1p = hdr_buf(1);
      const bool valid = p0 + lane < nunion;
      const float xv = *reinterpret_cast<const float*>(h0p + lz * 4);
      const uint4 m = *reinterpret_cast<const uint4*>(h0p + 256 + lz * 16);
      r = (int)(m.x & 0x1FFFFu);
      sq = __uint_as_float(m.w);  // |a_i|^2 (setup.c:130's rounded norm, squared again: cd.c:127)
      xi = valid ? xv : kInactive;
      part = tile_active(xi);
      const bool validN = p0 + 64 + lane < nunion;
      const float xn = *reinterpret_cast<const float*>(h1p + lz * 4);
      const uint32_t mn = *reinterpret_cast<const uint32_t*>(h1p + 256 + lz * 16);
      wantN = validN && tile_active(xn);
      rN = wantN ? (int)(mn & 0x1FFFFu) : 0;
      have_gi = false;
    };

    auto refresh_next = [&]() __attribute__((always_inline)) {
      int lz = lane;
      asm volatile("" : "+v"(lz));
      const char* const h1p = hdr_buf(1);
      const bool validN = p0 + 64 + lane < nunion;
      const float xn = *reinterpret_cast<const float*>(h1p + lz * 4);
      const uint32_t mn = *reinterpret_cast<const uint32_t*>(h1p + 256 + lz * 16);
      wantN = validN && tile_active(xn);
      rN = wantN ? (int)(mn & 0x1FFFFu) : 0;
    };
#if defined(SLIM_GRAMRP_CHECK)
    float chk_gi = 0.0f, chk_L = 0.0f;
    int chk_rN = 0, chk_r = 0;
    int chk_grp[4] = {0, 0, 0, 0};
    float chk_nd = 0.0f, chk_first_nd = 0.0f, chk_first_a = 0.0f, chk_first_b = 0.0f;
    bool chk_part = false;
    bool chk_on = false;
    int chk_bad = 0, chk_n = 0, chk_same = 0;   // (chk_same: the line already held the final value)
#endif
    int ucount = 0;
    auto flush_x = [&]() __attribute__((always_inline)) {
      if (wave == 0)
        for (int k = lane; k < ucount; k += 64) x[s_urow[k]] = s_unx[k];
      // (nothing else is in flight here: the wait costs a store's latency, and it takes the pending
      // store out of the compiler's book-keeping -- with loads and a store pending together it assumes
      // they complete out of order and turns every later wait of its own into vmcnt(0))
      SLIM_VMCNT(0);
      asm volatile("" ::: "memory");
      ucount = 0;
    };
    Row cur, nxt;
    cur.valid = false;
    nxt.valid = false;
#if SLIM_GRAMR_PROF
    // (a build for scripts/gramrp_prof.py: what the turns of the sweeps were, in place of D / U / bytes / rows)
    long long c_turns = 0, c_chained = 0, c_drain = 0, c_unchained = 0, c_sync = 0;
    unsigned long long t_part1 = 0, t_adv = 0, t_part2 = 0, t_post = 0, t_mark = 0;
#define SLIM_PT(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - t_mark; t_mark = now_; }
#else
#define SLIM_PT(acc)
#endif

    for (;;) {  // one row slot per turn: [groups 0..CP of cur] [advance] [groups CP+1.. of cur + head of nxt] [boundary]
#if SLIM_GRAMR_PROF
      t_mark = __builtin_readcyclecounter();
#endif
      // ---- part 1: positions 0 .. CP (the register groups)
      if (cur.valid) {
        row_loads(cur, r, rN);
#if defined(SLIM_GRAMRP_CHECK)
        chk_rN = rN;
#endif
        static_for<CP + 1>([&](auto pc_) __attribute__((always_inline)) {
          constexpr int p = decltype(pc_)::value;
          request(cur, p + AH);
          // behind the group at position p: AH ring requests -- and, for the first AH positions (whose
          // requests the previous turn made), this turn's register loads
          SLIM_VMCNT(p < AH ? AH + kRowLoads : AH);
          asm volatile("" ::: "memory");
          consume(cur, pc_);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      SLIM_PT(t_part1)
      // ---- advance: what streams next (ONE site)
      nxt.valid = false;
      int ntrans = 0;
      bool done = false;
      const int nbase = cur.valid ? slot_of(cur.base, K % NS) : ring_base;  // ring slot of the next row's group 0
      const int nbs = cur.valid ? (cur.bs ^ 1) : 0;
      if (phase == 0) {
        if (cur.valid) {
          if (first_row) first_row = false;
          else {
            ++nrows_read;
            nhi16_read += min(cur.hk * (kPackGroup / 16), nchunks) + min(cur.h2k * (kPackGroup / 16), nchunks);
          }
        }
        int row = -1;
        float nd = 0.0f;
        if (init_row) {
          init_row = false;
          row = item;
          nd = 1.0f;
        } else {
          while (fe < we) {
            const int kk = uni(S.icolind[fe]);
            ++fe;
            if (kk < ncols) {
              const float xk = uni(x[kk]);
              if (xk > kEps) {
                row = kk;
                nd = -xk;
                break;
              }
            }
          }
        }
        if (row >= 0) {
          const uint4 mr = META[row];
          nxt = make_row(row, nd, uni(mr.x), uni(mr.y), uni(mr.w), nbase, nbs);
        } else {
          phase = 3;  // the set-up is done: the first sweep starts with nothing in flight
        }
      } else if (phase == 3) {
        // here: no row in flight (cur is invalid).  A sweep starts, or the descent is over.
        flush_x();
        bool sweep = false;
        if (!conv) {
          if (t >= maxit) niters = maxit + 1;  // cd.c:140
          else if (nunion == 0) {
            conv = 1;
            niters = t + 1;
          } else sweep = true;
        }
        if (sweep) {
          // (the first batch of a sweep may hold coordinates of the last batch of the sweep before:
          // every wavefront's stores of x are behind this barrier)
          __syncthreads();
          phase = 1;
          p0 = 0;
          dlt = 0.0f;
          pc = perm_make((uint32_t)nunion, perm_key(S.seed, gkey, (uint32_t)t));
          hb = 0;
          request_items(0);
          SLIM_VMCNT(0);
          asm volatile("" ::: "memory");
          request_header(0);
          request_items(64);
          SLIM_VMCNT(0);
          asm volatile("" ::: "memory");
          request_header(1);
          request_items(128);
          SLIM_VMCNT(0);
          asm volatile("" ::: "memory");
          request_header(2);
          request_items(192);  // (the next transition's header request reads them)
          SLIM_VMCNT(0);
          asm volatile("" ::: "memory");
          load_lanes();
          posted_next = false;
        } else {
          phase = 4;
        }
      } else if (phase == 1) {
        // -- the g of the decision batch's lanes, current up to and including the row in flight.
        //    `ahead`: which batch the line posted at the last boundary belongs to, counted from the
        //    decision batch (1: the batch behind it -- posted while this one was being decided;
        //    0: this batch -- posted because it had not been entered yet)
        int ahead = posted_next ? 1 : 0;
        if (!cur.valid) {
          // nothing streams: every header request has to land now (a batch that ended twice in one
          // turn read the ranks of the batch behind it out of a buffer still in flight)
          SLIM_VMCNT(0);
          asm volatile("" ::: "memory");
          refresh_next();
          flush_x();
        }
        if (have_gi && cur.valid) gi = fmaf(cur.nd, gsel_of(cur, r, gb0, gb1, gb3), gi);
        if (cur.valid) {
          ++nrows_read;
          nhi16_read += min(cur.hk * (kPackGroup / 16), nchunks) + min(cur.h2k * (kPackGroup / 16), nchunks);
        }
        while (!done) {
#if defined(SLIM_GRAMRP_NOCHAIN)  // (bisect build: a mover is decided only with nothing in flight)
          if (cur.valid) break;
#endif
          if (ucount > kUpd - 64) break;  // (the update buffer could overflow: the next turn, with nothing in flight, empties it)
          if (!have_gi) {
            if (ahead != 0) break;  // its g is not exported yet: the next boundary does that
            gi = s_gB[par ^ 1][lane];  // rows before `cur` ...
#if defined(SLIM_GRAMRP_CHECK)
            chk_L = gi;
#endif
            // ... and the row in flight, replayed in the registers' order (r was rN when cur's head was made)
            if (cur.valid) gi = fmaf(cur.nd, gsel_of(cur, r, gc0, gc1, gc3), gi);
#if defined(SLIM_GRAMRP_CHECK)
            if (cur.valid) {
              chk_gi = gi;
              chk_on = true;
              chk_r = r;
              chk_part = part;
              chk_nd = cur.nd;
            }
#endif
            have_gi = true;
            pend = __ballot(part);
          }
          // the next mover of this batch, if any (cd.c:121-133)
          uint64_t m = 0;
          float nx = 0.0f, d = 0.0f;
          if (pend != 0) {
            const float xeff = (xi > kEps || xi < -kEps) ? xi : 0.0f;
            const float cn = sqrtf(sq);
            const float num = cd_num(gi, xeff, sq);
            nx = num > l1 ? (num - l1) / cd_den(cn, l2) : 0.0f;
            const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
            d = neff - xeff;
            m = __ballot(part && nx != xi) & pend;
          }
          if (m != 0) {
            const int f = __builtin_ctzll(m);
            const float d_f = lane_bcast(d, f);
            const float nx_f = lane_bcast(nx, f), xi_f = lane_bcast(xi, f);
            dlt = fmaf(nx_f - xi_f, nx_f - xi_f, dlt);
            // the mover's item and row record: read back from the header buffer at a uniform address
            // (every lane the same 20 bytes) instead of four registers per lane kept for this moment
            const char* const hm = hdr_buf(0);
            const uint4 mrec = *reinterpret_cast<const uint4*>(hm + 256 + f * 16);
            const int row = uni(*reinterpret_cast<const int*>(hm + 1280 + f * 4));
            if (wave == 0 && lane == f) {
              s_urow[ucount] = row;
              s_unx[ucount] = nx;
            }
            ++ucount;
            pend = f == 63 ? 0ull : (pend & ~((2ull << f) - 1ull));
            if (d_f == 0.0f) continue;  // (a change below the epsilon of cd.c:27 moves no g)
            const uint32_t rx = uni(mrec.x), ry = uni(mrec.y), rz = uni(mrec.z), rw = uni(mrec.w);
            Uu += (unsigned long long)rz;
            nxt = make_row(row, -d_f, rx, ry, rw, nbase, nbs);
            done = true;
          } else {
            // the batch is exhausted
            p0 += 64;
            if (p0 >= nunion) {  // the sweep ends (cd.c:135-138)
              if (dlt < S.opt_tol) {
                conv = 1;
                niters = t + 1;
              } else {
                ++t;
              }
              phase = 3;
              done = true;
            } else {
#if defined(SLIM_GRAMRP_NOCHAINT)  // (bisect build: a batch ends only with nothing in flight)
              if (cur.valid) { p0 -= 64; break; }
#endif
              // the next batch becomes the decision batch; the header of the batch two behind it is
              // requested.  (With a row in flight everything read here was requested before that
              // row's head and has landed with its first group; with none, wait.)
              if (!cur.valid) {
                SLIM_VMCNT(0);
                asm volatile("" ::: "memory");
              }
              hb = hb + 1 >= 3 ? 0 : hb + 1;
              load_lanes();
              request_header(2);           // items of batch p0 + 128: requested one transition ago
              request_items(p0 + 192);
              ++ntrans;
              --ahead;
            }
          }
        }
      } else if (phase == 4) {
        // wavefront 0 counts the kept coefficients and claims the arena space
        __syncthreads();
        if (wave == 0) {
          s_e2[lane] = 0.0;   // (in wavefront 0's ring slots: nothing streams from here on)
          s_reg[lane] = 0.0;
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
        out_posted = false;
      }
      if (phase == 2) {
        // output (estimate.c:477-505): 64 item ids per turn, ascending; the g of the kept ones was
        // exported at the boundary behind the previous turn
        if (out_posted) {
          const float g0 = s_gB[par ^ 1][lane];
          if (keep) {
            if (wave == 0) s_e2[lane] += (double)xi * ((double)arow[i] + (double)g0);
            if (fits && wave == 0) {
              const int64_t dst = (int64_t)off + wpos + __popcll(mkeep & ((1ull << lane) - 1ull));
              S.out_ind[dst] = i;
              S.out_val[dst] = xi;
            }
          }
          wpos += __popcll(mkeep);
          ib += 64;
          out_posted = false;
        }
        mkeep = 0;
        while (ib < ncols) {
          i = ib + lane;
          xi = i < ncols ? x[i] : kInactive;
          const bool act = tile_active(xi);
          keep = act && fabsf(xi) > kEps;
          if (act && wave == 0)
            s_reg[lane] += 0.5 * (double)l2 * (double)xi * (double)xi + (double)l1 * (double)fabsf(xi);
          mkeep = __ballot(keep);
          if (mkeep != 0) break;
          ib += 64;
        }
        if (mkeep == 0) break;  // every id is written: the problem is done
        r = keep ? (int)(META[i].x & 0x1FFFFu) : 0;
      }

      SLIM_PT(t_adv)
#if SLIM_GRAMR_PROF
      if (phase == 1) {
        ++c_turns;
        if (cur.valid && nxt.valid && ntrans <= 1) ++c_chained;
        else if (cur.valid && nxt.valid) ++c_drain;
        else if (nxt.valid) ++c_unchained;
        else if (!cur.valid) ++c_sync;
      }
#endif
      // ---- part 2: the last groups of cur, the head of nxt in their shadow
      static_for<AH>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, k = CP + 1 + j;
        if (nxt.valid) {
          if constexpr (j == 0) head_base(nxt);
          request(nxt, j);
        }
        if (cur.valid) {
          // behind the group at position k: cur's last AH - 1 - j groups, three header requests per
          // batch that ended, the head's base bytes, j + 1 ring requests of nxt -- loads only (see s_urow)
          if (nxt.valid && ntrans == 0) SLIM_VMCNT(AH + kHeadOps);
          else if (nxt.valid && ntrans == 1) SLIM_VMCNT(AH + 3 + kHeadOps);
          else if (nxt.valid) SLIM_VMCNT(0);  // (set-up rows with their own loads, two batches ended: rare)
          else SLIM_VMCNT(AH - 1 - j);
          asm volatile("" ::: "memory");
          consume(cur, std::integral_constant<int, k>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      if (cur.valid) ring_base = slot_of(cur.base, K % NS);

      SLIM_PT(t_part2)
#if defined(SLIM_GRAMRP_CHECK)
      // (bisect build) a batch was entered with a row in flight: now that the row is applied, the
      // registers' g of that batch's lanes must be the replayed value -- for the lanes no later row of
      // this turn's decision could have touched, i.e. all (nxt is not applied yet)
      if (chk_on) {
        post(chk_part, chk_r);
        const float gx = s_gB[par ^ 1][lane];
        if (chk_part) {
          ++chk_n;
          if (gx != chk_gi) ++chk_bad;
          if (gx != chk_gi && chk_first_nd == 0.0f) {
            chk_first_nd = chk_nd;
            chk_first_a = (chk_gi - chk_L) / chk_nd;   // the entry the replay took
            chk_first_b = (gx - chk_L) / chk_nd;       // the entry the registers saw
          }
          if (gx != chk_gi) {
            const int kg = chk_r >> 13;
            ++chk_grp[kg == 0 ? 0 : (kg <= 3 ? 1 : (kg <= 9 ? 2 : 3))];
          }
        }
        chk_on = false;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (the line is reused by the regular post below)
      }
#endif
      // ---- boundary: the registers hold every row up to cur; the g of the visits decided next is exported
      if (phase == 1 || phase == 2) {
        // sweeps: the batch behind the decision batch -- or the decision batch itself, when it has not
        // been entered yet (the next turn does that); output: the kept ones among the 64 ids
        const bool next = phase == 1 && have_gi;
        post(next ? wantN : (phase == 1 ? part : keep), next ? rN : r);
        posted_next = next;
        out_posted = phase == 2;
      }
      SLIM_PT(t_post)
      cur = nxt;
    }

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
#if defined(SLIM_GRAMRP_CHECK)
        S.st_na[item] = (int)wave_sum((float)chk_bad) * 65536 + min(65535, (int)wave_sum((float)chk_same));
#else
        S.st_na[item] = s_na;
#endif
        S.st_sweeps[item] = niters;
        S.st_conv[item] = conv;
#if SLIM_GRAMR_PROF
        S.st_D[item] = (int64_t)((c_turns << 32) | c_chained);
        S.st_U[item] = (int64_t)((c_drain << 40) | (c_unchained << 20) | c_sync);
        S.st_B[item] = (int64_t)(((t_part1 >> 10) << 32) | (t_adv >> 10));
        S.st_G[item] = (int)(t_part2 >> 10);
        S.st_na[item] = (int)(t_post >> 10);
        (void)Uu;
        (void)nhi16_read;
#else
#if defined(SLIM_GRAMRP_CHECK)
        S.st_D[item] = (int64_t)wave_sum((float)chk_grp[0]) | ((int64_t)wave_sum((float)chk_grp[1]) << 16) |
                       ((int64_t)wave_sum((float)chk_grp[2]) << 32) | ((int64_t)wave_sum((float)chk_grp[3]) << 48);
#else
        S.st_D[item] = (int64_t)s_D * (int64_t)(conv ? niters : maxit);
#endif
#if defined(SLIM_GRAMRP_CHECK)
        {
          // (the lowest lane with a mismatch reports: the entry the replay took, the entry the registers saw)
          const uint64_t mm = __ballot(chk_first_nd != 0.0f);
          const int fl = mm ? __builtin_ctzll(mm) : 0;
          const float fa = lane_bcast(chk_first_a, fl), fb = lane_bcast(chk_first_b, fl);
          S.st_U[item] = mm ? (int64_t)(((uint64_t)__float_as_uint(fa) << 32) | (uint64_t)__float_as_uint(fb)) : 0;
          S.st_conv[item] = __ballot(misaligned != 0) ? 77 : conv;
        }
#else
        S.st_U[item] = (int64_t)Uu;
#endif
        S.st_G[item] = nrows_read;
        S.st_B[item] = (int64_t)nrows_read * (P.ldb + 16 * (int64_t)(nchunks < NT ? nchunks : NT)) + (int64_t)nhi16_read * 16;
#endif
#if 0
        S.st_err[item] = err;
        S.st_obj[item] = err + (float)reg;
#endif
      }
    }
    __syncthreads();
  }
  };
  body(P.lo, P.hi, P.base, P.meta, x_slab, S.ulist);
#undef SLIM_VMCNT
#undef SLIM_PT
#undef SLIM_LDS_DMA
}

}  // namespace slimamd
