// cd_gram.hpp -- coordinate descent in item space, on G = R^T R held in HBM.
//
// The residual kernels (cd_wave.hpp, cd_tile.hpp) keep r = y - yhat over the USERS and pay one
// gather per nnz of the visited column for the dot a_i . r, and one write per nnz when the
// coefficient moved.  cd.c:121-123 only ever needs r through the products a_i . r:
//
//     g_i := a_i . r = aTy_i - sum_j G_ij x_j ,      G = R^T R,   aTy = G[:, iC]
//
// so a problem can carry g over the ITEMS instead (ncols floats: it fits the LDS of a CU up to
// ~40K items) and never touch a user again:
//
//     visit of i  : num = g_i + x_i |a_i|^2        (== aTy_i - a_i.(yhat - x_i a_i), cd.c:121-123)
//                   x_i' by the same soft threshold, same epsilon rule          cd.c:124-128, :27
//     update      : g -= (x_i' - x_i) G[i, :]      (one contiguous row of G, 4 ncols bytes)
//
// Same update rule, same cap min(50 nnz, maxniters), same stop rule sum (dx)^2 < optTol, same
// visiting order as the tile kernel (keyed permutation of the UNION of the active sets of the
// 32 work-list neighbours of the item, cd_perm.hpp), so a column's result is the tile kernel's
// up to fp32 rounding and the oracle's tile walk checks it visit for visit.  What it needs is G:
// ncols^2 floats (C5, 20K items: 1.6 GB), exact in fp32 for binary R (co-rating counts), built
// once per matrix by the tile kernel's screen pass (S.gram_mode 3: a_i . y for every column i
// and every item of a tile IS a block of 32 rows of G; symmetric fill; for a binary matrix of up
// to ~1.2M users with y packed one word per user in LDS) -- worth it whenever most columns of a
// matrix are solved, and paid once for a model-selection grid (slim_mselect.c:94-113: 45
// (l1, l2) pairs over one R, each warm-started from the previous model).  The active set
// {i != iC : aTy_i > l1} (estimate.c:433-444) is read off row iC; a warm start folds the previous
// coefficients as g -= x_j G[j, :] (cd.c:108-110 in item space).
//
// One workgroup per problem.  Every wavefront walks the visiting order redundantly, 64 visits at
// a time: lane L holds visit p0 + L (coordinate, x, g, norms) and all lanes evaluate their
// update against the g they hold; the first lane whose coefficient moves is THE next change of
// the sequential algorithm (nothing before it changed anything), it is applied, every lane
// corrects the g it holds with one element of the fetched row (g_L -= d G[i_f, i_L]) and the
// batch goes on behind it.  The workgroup's threads own disjoint float4 slices of g in LDS and
// apply the row to their slice; nobody reads LDS inside a batch, so a batch needs two barriers,
// not two per update.  Control flow is workgroup-uniform because every wavefront computes the
// same values from the same data.
//
// Traffic per problem and sweep: 4 ncols bytes per UPDATE (a row of G; visits that change
// nothing touch no memory beyond the 64-visit batch header), against 4.1 bytes per nnz of R per
// problem and visit in the tile kernel: C5, 20K items of ~50K nnz each -- a sweep with 4000
// updates reads 0.32 GB here and ~8 GB there.
#pragma once
#include "cd_tile.hpp"

namespace slimamd {

// Union of the active sets of every tile (32 consecutive entries of the work list), ascending:
// ulist[tile][0 .. nunion[tile]).  One wavefront per tile; row iC of G is aTy of problem iC.
__global__ __launch_bounds__(64) void gram_union_kernel(const DevMatrix A, const SolveArgs S) {
  const int lane = threadIdx.x;
  const int grp = blockIdx.x;
  const int base = grp * 32;
  const int nprob = (S.nwork - base) < 32 ? (S.nwork - base) : 32;
  const int item = lane < nprob ? S.order[base + lane] : -1;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  int* __restrict__ ul = S.ulist + (int64_t)grp * S.u_stride;
  const float* __restrict__ G = S.G;
  const int ncols = A.ncols;
  int nu = 0;
  for (int ib = 0; ib < ncols; ib += 64) {
    const int i = ib + lane;
    bool any = false;
    for (int q = 0; q < nprob; ++q) {
      const int it = lane_bcast(item, q);
      const float a = i < ncols ? G[(int64_t)it * S.G_ld + i] : 0.0f;
      any = any || (i != it && a > S.l1);
    }
    const uint64_t m = __ballot(any);
    if (any) ul[nu + __popcll(m & lane_lt)] = i;
    nu += __popcll(m);
  }
  if (lane == 0) S.tile_nunion[grp] = nu;
}

// NW wavefronts per workgroup, V float4 of a row per thread (4 V 64 NW >= ncols_pad).
// V = 0: more items than the LDS holds (> ~40K) -- g lives in a per-workgroup slab in HBM (it is
// read and rewritten on every update: 12 ncols bytes per update instead of 4, most of it served
// by the Infinity Cache), every thread walks its float4 slices in a loop.
template <int NW, int V>
__global__ __launch_bounds__(64 * NW, (V == 0 && NW == 8) ? 4 : 1) void cd_gram_kernel(const DevMatrix A, const SolveArgs S) {
  constexpr int NT = 64 * NW;
  constexpr bool LDSG = V > 0;
  constexpr int VV = LDSG ? V : 1;  // (array bounds of the LDS form)
  extern __shared__ __attribute__((aligned(16))) float g_lds[];  // [ncols_pad]: g_i = a_i . r
  float* const g = LDSG ? g_lds : S.slab + (int64_t)blockIdx.x * S.slab_stride;
  __shared__ int s_p, s_na;
  __shared__ unsigned long long s_D, s_U;
  __shared__ double s_red[2][NW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const int ncols = A.ncols;
  const int n4 = S.ncols_pad >> 2;
  const float l1 = S.l1, l2 = S.l2;
  const float* __restrict__ Gm = S.G;
  const int64_t ld = S.G_ld;
  float* const x = S.xslab + (int64_t)blockIdx.x * S.x_stride;  // [ncols_pad], -inf = inactive
  float4* const g4 = reinterpret_cast<float4*>(g);
  float4* const x4 = reinterpret_cast<float4*>(x);
  const int64_t* __restrict__ colptr = A.colptr;

  for (;;) {
    if (tid == 0) {
      s_p = atomicAdd(S.queue, 1);
      s_na = 0;
      s_D = 0;
      s_U = 0;
    }
    __syncthreads();
    const int p = s_p;
    if (p >= S.nwork) break;
    const int item = uni(S.order[p]);
    const int grp = p >> 5;
    // position of the item's tile in the unsharded work list (cd_tile.hpp: gkey)
    const uint32_t gkey = (uint32_t)(grp * S.shard_count + S.shard_index);
    const int* __restrict__ ul = S.ulist + (int64_t)grp * S.u_stride;
    const int nunion = uni(S.tile_nunion[grp]);
    const float* __restrict__ arow = Gm + (int64_t)item * ld;  // aTy of this problem

    // -- g = aTy, x = 0 on the active set {i != iC : aTy_i > l1} (estimate.c:433-444), -inf elsewhere
    {
      const float4* __restrict__ a4 = reinterpret_cast<const float4*>(arow);
      int na = 0;
      const int nslice = LDSG ? V : (n4 + NT - 1) / NT;
#pragma unroll
      for (int j = 0; j < nslice; ++j) {
        const int c = tid + j * NT;
        if (c < n4) {
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
          g4[c] = a;
          x4[c] = xs;
        }
      }
      na = (int)wave_sum((float)na);  // (< 2^24: exact)
      if (lane == 0 && na) atomicAdd(&s_na, na);
    }
    __syncthreads();

    // -- warm start (estimate.c:453-464): previous coefficients of the coordinates active now
    //    (a negative value ends up 0 there: the flag-clearing loop resets every x < 0), folded
    //    into g row by row (cd.c:108-110 in item space): g -= x_j G[j, :].  Every thread keeps
    //    its slices of g in registers across the fold; FU rows are in flight together.
    int64_t nrows_read = 0;  // rows of G read by this problem (fold + updates): the byte model
    if (S.icolptr != nullptr && item < S.incols) {
      const int64_t ws = uni(S.icolptr[item]), we = uni(S.icolptr[item + 1]);
      for (int64_t e = ws + tid; e < we; e += NT) {
        const int k = S.icolind[e];
        if (k < ncols && tile_active(x[k])) {
          const float v = S.icolval[e];
          x[k] = v < 0.0f ? 0.0f : v;
        }
      }
      __syncthreads();
      {
        int nf = 0;
        for (int64_t e = ws + lane; e < we; e += 64) {
          const int k = S.icolind[e];
          nf += (k < ncols && x[k] > kEps) ? 1 : 0;
        }
        nrows_read += (int64_t)(int)wave_sum((float)nf);
      }
      constexpr int FU = !LDSG ? 4 : (V <= 2 ? 4 : (V <= 5 ? 2 : 1));
      // which row and coefficient entry e folds: entries that fold nothing (past the column,
      // outside the active set, below the epsilon of cd.c:27) read row iC with a zero
      // coefficient -- no branch in the stream of row loads
      auto entry = [&](const int64_t e, float& v, const float4*& rp) {
        int k = item;
        v = 0.0f;
        if (e < we) {
          const int kk = uni(S.icolind[e]);
          if (kk < ncols) {
            const float xk = uni(x[kk]);
            if (xk > kEps) {
              k = kk;
              v = xk;
            }
          }
        }
        rp = reinterpret_cast<const float4*>(Gm + (int64_t)k * ld);
      };
      if (LDSG) {
        float4 acc[VV];
#pragma unroll
        for (int j = 0; j < VV; ++j) {
          const int c = tid + j * NT;
          acc[j] = c < n4 ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int64_t e0 = ws; e0 < we; e0 += FU) {
          float xv[FU];
          const float4* rp[FU];
#pragma unroll
          for (int f = 0; f < FU; ++f) entry(e0 + f, xv[f], rp[f]);
          float4 rv[FU][VV];
#pragma unroll
          for (int f = 0; f < FU; ++f)
#pragma unroll
            for (int j = 0; j < VV; ++j) {
              const int c = tid + j * NT;
              rv[f][j] = c < n4 ? rp[f][c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
          for (int f = 0; f < FU; ++f)
#pragma unroll
            for (int j = 0; j < VV; ++j) {
              acc[j].x = fmaf(-xv[f], rv[f][j].x, acc[j].x);
              acc[j].y = fmaf(-xv[f], rv[f][j].y, acc[j].y);
              acc[j].z = fmaf(-xv[f], rv[f][j].z, acc[j].z);
              acc[j].w = fmaf(-xv[f], rv[f][j].w, acc[j].w);
            }
        }
#pragma unroll
        for (int j = 0; j < VV; ++j) {
          const int c = tid + j * NT;
          if (c < n4) g4[c] = acc[j];
        }
      } else {
        // g in HBM: one pass over the entries per 4 slices of g, which stay in registers for the
        // pass -- every row is read once in all (in pieces), g is read and written once
        constexpr int CS = 4;
        for (int c0 = tid; c0 < n4; c0 += CS * NT) {
          float4 acc[CS];
#pragma unroll
          for (int j = 0; j < CS; ++j) {
            const int c = c0 + j * NT;
            acc[j] = c < n4 ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          for (int64_t e0 = ws; e0 < we; e0 += FU) {
            float xv[FU];
            const float4* rp[FU];
#pragma unroll
            for (int f = 0; f < FU; ++f) entry(e0 + f, xv[f], rp[f]);
            float4 rv[FU][CS];
#pragma unroll
            for (int f = 0; f < FU; ++f)
#pragma unroll
              for (int j = 0; j < CS; ++j) {
                const int c = c0 + j * NT;
                rv[f][j] = c < n4 ? rp[f][c] : make_float4(0.f, 0.f, 0.f, 0.f);
              }
#pragma unroll
            for (int f = 0; f < FU; ++f)
#pragma unroll
              for (int j = 0; j < CS; ++j) {
                acc[j].x = fmaf(-xv[f], rv[f][j].x, acc[j].x);
                acc[j].y = fmaf(-xv[f], rv[f][j].y, acc[j].y);
                acc[j].z = fmaf(-xv[f], rv[f][j].z, acc[j].z);
                acc[j].w = fmaf(-xv[f], rv[f][j].w, acc[j].w);
              }
          }
#pragma unroll
          for (int j = 0; j < CS; ++j) {
            const int c = c0 + j * NT;
            if (c < n4) g4[c] = acc[j];
          }
        }
      }
    }

    // -- sweeps (cd.c:112-139)
    int maxit = 0;
    {
      const int64_t cap = 50 * (uni(colptr[item + 1]) - uni(colptr[item]));  // estimate.c:448-449
      maxit = cap < (int64_t)S.maxniters ? (int)cap : S.maxniters;
    }
    int niters = 0, conv = 0;
    unsigned long long Dq = 0, Uq = 0;  // SURVEY.md 8(d) counters, per lane (same in every wave)
    for (int t = 0;; ++t) {
      if (t >= maxit) {  // loop exhausted without convergence: niters = t + 1 (cd.c:140)
        niters = maxit + 1;
        break;
      }
      float dlt = 0.0f;
      const PermCtx pc = perm_make((uint32_t)nunion, perm_key(S.seed, gkey, (uint32_t)t));
      if (!LDSG) {
        // g in HBM: a pass over g per update would move 8 ncols bytes on top of the row's 4.
        // Batches of 64 MS visits (MS per lane); the updates of a batch are only NOTED (which
        // lane of which slot, by how much) while every lane keeps the g of its own visits current
        // from single elements of the rows (below); when the batch is decided, every thread
        // applies all of its rows to its slices of g in ONE pass -- g is read and written once per
        // batch, the rows once each, their loads independent of one another.
        constexpr int MS = 8;  // (8 slots per lane keep the kernel at 128 VGPRs: two workgroups per CU)
        for (int p0 = 0; p0 < nunion; p0 += 64 * MS) {
          __syncthreads();  // the previous batch's pass over g is complete
          int i[MS], len[MS];
          float xi[MS], sq[MS], cn[MS], gi[MS], dsv[MS];
          bool part[MS];
          uint64_t umask[MS];
#pragma unroll
          for (int sl = 0; sl < MS; ++sl) {
            const int pos = p0 + sl * 64 + lane;
            const bool valid = pos < nunion;
            i[sl] = 0;
            xi[sl] = kInactive;
            if (valid) {
              i[sl] = ul[perm_index(pc, (uint32_t)pos)];
              xi[sl] = x[i[sl]];
            }
            part[sl] = valid && tile_active(xi[sl]);
            sq[sl] = 0.0f;
            cn[sl] = 0.0f;
            len[sl] = 0;
            if (part[sl]) {
              sq[sl] = A.csq[i[sl]];
              cn[sl] = A.cnorm[i[sl]];
              len[sl] = (int)(colptr[i[sl] + 1] - colptr[i[sl]]);
            }
            gi[sl] = g[i[sl]];
            dsv[sl] = 0.0f;
            umask[sl] = 0ull;
            Dq += (unsigned long long)len[sl];
          }
          __syncthreads();  // every wavefront holds its batch: g may change now
          bool any_upd = false;
#pragma unroll
          for (int sl = 0; sl < MS; ++sl) {
            uint64_t pend = __ballot(part[sl]);
            while (pend) {
              const float xeff = (xi[sl] > kEps || xi[sl] < -kEps) ? xi[sl] : 0.0f;
              const float num = cd_num(gi[sl], xeff, sq[sl]);
              const float nx = num > l1 ? (num - l1) / cd_den(cn[sl], l2) : 0.0f;
              const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
              const float d = neff - xeff;
              const uint64_t m = __ballot(part[sl] && nx != xi[sl]) & pend;
              if (m == 0) break;
              const int f = __builtin_ctzll(m);
              const int i_f = lane_bcast(i[sl], f);
              const float d_f = lane_bcast(d, f);
              const float nx_f = lane_bcast(nx, f), xi_f = lane_bcast(xi[sl], f);
              dlt = fmaf(nx_f - xi_f, nx_f - xi_f, dlt);
              if (wave == 0 && lane == f) x[i[sl]] = nx;
              pend = f == 63 ? 0ull : (pend & ~((2ull << f) - 1ull));
              if (d_f != 0.0f) {
                ++nrows_read;
                if (lane == f) {
                  Uq += (unsigned long long)len[sl];
                  dsv[sl] = d_f;
                }
                umask[sl] |= 1ull << f;
                any_upd = true;
                // the visits still ahead (the rest of this slot, every later slot) see the update
                // through one element of the row each
                const float* __restrict__ row = Gm + (int64_t)i_f * ld;
#pragma unroll
                for (int s2 = 0; s2 < MS; ++s2)
                  if (s2 >= sl) gi[s2] = fmaf(-d_f, row[i[s2]], gi[s2]);
              }
            }
          }
          if (any_upd) {  // one pass over g for all the rows of the batch, in visiting order
            // (CS float4 of a row in flight per thread: with 8 wavefronts per CU, 4 of them are
            // 32 KB per CU -- by Little's law ~4 TB/s over the chip, which is what it measured)
            constexpr int CS = 4;
            for (int c0 = tid; c0 < n4; c0 += CS * NT) {
              float4 gv[CS];
#pragma unroll
              for (int j = 0; j < CS; ++j) {
                const int c = c0 + j * NT;
                gv[j] = g4[c < n4 ? c : c0];
              }
#pragma unroll
              for (int sl = 0; sl < MS; ++sl) {
                uint64_t mm = umask[sl];
                while (mm) {
                  const int f = __builtin_ctzll(mm);
                  mm &= mm - 1ull;
                  const int i_f = lane_bcast(i[sl], f);
                  const float d_f = lane_bcast(dsv[sl], f);
                  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(Gm + (int64_t)i_f * ld);
                  float4 rv[CS];
#pragma unroll
                  for (int j = 0; j < CS; ++j) {
                    const int c = c0 + j * NT;
                    rv[j] = r4[c < n4 ? c : c0];
                  }
#pragma unroll
                  for (int j = 0; j < CS; ++j) {
                    gv[j].x = fmaf(-d_f, rv[j].x, gv[j].x);
                    gv[j].y = fmaf(-d_f, rv[j].y, gv[j].y);
                    gv[j].z = fmaf(-d_f, rv[j].z, gv[j].z);
                    gv[j].w = fmaf(-d_f, rv[j].w, gv[j].w);
                  }
                }
              }
#pragma unroll
              for (int j = 0; j < CS; ++j) {
                const int c = c0 + j * NT;
                if (c < n4) g4[c] = gv[j];
              }
            }
          }
        }
      } else
      for (int p0 = 0; p0 < nunion; p0 += 64) {
        __syncthreads();  // every thread's row updates of the previous batch are in LDS
        const int pos = p0 + lane;
        const bool valid = pos < nunion;
        int i = 0;
        float xi = kInactive, sq = 0.0f, cn = 0.0f;
        int len = 0;
        if (valid) {
          i = ul[perm_index(pc, (uint32_t)pos)];
          xi = x[i];
        }
        const bool part = valid && tile_active(xi);
        if (part) {
          sq = A.csq[i];
          cn = A.cnorm[i];
          len = (int)(colptr[i + 1] - colptr[i]);
        }
        float gi = g[i];
        __syncthreads();  // every wavefront holds its batch: the slices may change now
        uint64_t pend = __ballot(part);
        Dq += (unsigned long long)len;
        while (pend) {
          const float xeff = (xi > kEps || xi < -kEps) ? xi : 0.0f;
          const float num = cd_num(gi, xeff, sq);
          const float nx = num > l1 ? (num - l1) / cd_den(cn, l2) : 0.0f;
          const float neff = (nx > kEps || nx < -kEps) ? nx : 0.0f;
          const float d = neff - xeff;
          const uint64_t m = __ballot(part && nx != xi) & pend;
          if (m == 0) break;  // nothing else in the batch moves
          const int f = __builtin_ctzll(m);
          const int i_f = lane_bcast(i, f);
          const float d_f = lane_bcast(d, f);
          const float nx_f = lane_bcast(nx, f), xi_f = lane_bcast(xi, f);
          dlt = fmaf(nx_f - xi_f, nx_f - xi_f, dlt);
          if (wave == 0 && lane == f) x[i] = nx;
          pend = f == 63 ? 0ull : (pend & ~((2ull << f) - 1ull));
          if (d_f != 0.0f) {
            ++nrows_read;
            if (lane == f) Uq += (unsigned long long)len;
            const float* __restrict__ row = Gm + (int64_t)i_f * ld;
            const float4* __restrict__ r4 = reinterpret_cast<const float4*>(row);
            const float gsel = row[i];  // the element of the row this lane's visit needs
            if (LDSG) {
              float4 rv[VV];
#pragma unroll
              for (int j = 0; j < VV; ++j) {
                const int c = tid + j * NT;
                rv[j] = c < n4 ? r4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
              }
              gi = fmaf(-d_f, gsel, gi);
#pragma unroll
              for (int j = 0; j < VV; ++j) {
                const int c = tid + j * NT;
                if (c < n4) {
                  float4 gv = g4[c];
                  gv.x = fmaf(-d_f, rv[j].x, gv.x);
                  gv.y = fmaf(-d_f, rv[j].y, gv.y);
                  gv.z = fmaf(-d_f, rv[j].z, gv.z);
                  gv.w = fmaf(-d_f, rv[j].w, gv.w);
                  g4[c] = gv;
                }
              }
            } else {
              constexpr int CS = 4;
              for (int c0 = tid; c0 < n4; c0 += CS * NT) {
                float4 rv[CS], gv[CS];
#pragma unroll
                for (int j = 0; j < CS; ++j) {
                  const int c = c0 + j * NT;
                  const int cc = c < n4 ? c : c0;  // (clamped: no load under a condition)
                  rv[j] = r4[cc];
                  gv[j] = g4[cc];
                }
#pragma unroll
                for (int j = 0; j < CS; ++j) {
                  const int c = c0 + j * NT;
                  if (c < n4) {
                    gv[j].x = fmaf(-d_f, rv[j].x, gv[j].x);
                    gv[j].y = fmaf(-d_f, rv[j].y, gv[j].y);
                    gv[j].z = fmaf(-d_f, rv[j].z, gv[j].z);
                    gv[j].w = fmaf(-d_f, rv[j].w, gv[j].w);
                    g4[c] = gv[j];
                  }
                }
              }
              gi = fmaf(-d_f, gsel, gi);
            }
          }
        }
      }
      if (dlt < S.opt_tol) {  // cd.c:135-138
        conv = 1;
        niters = t + 1;
        break;
      }
    }
    if (wave == 0) {
      if (Dq) atomicAdd(&s_D, Dq);
      if (Uq) atomicAdd(&s_U, Uq);
    }
    __syncthreads();

    // -- 1/2 ||r||^2 and the objective (estimate.c:477-489) in item space:
    //    ||y - A x||^2 = y.y - 2 x.aTy + x.G x = |a_iC|^2 - sum_i x_i (aTy_i + g_i)
    {
      double e2 = 0.0, reg = 0.0;
      for (int i = tid; i < ncols; i += NT) {
        const float xv = x[i];
        if (tile_active(xv)) {
          reg += 0.5 * (double)l2 * (double)xv * (double)xv + (double)l1 * (double)fabsf(xv);
          if (xv > kEps || xv < -kEps) e2 += (double)xv * ((double)arow[i] + (double)g[i]);
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        e2 += __shfl_xor(e2, o);
        reg += __shfl_xor(reg, o);
      }
      if (lane == 0) {
        s_red[0][wave] = e2;
        s_red[1][wave] = reg;
      }
    }
    __syncthreads();

    // -- output: wavefront 0 compacts |x| > 1e-7, ascending ids (estimate.c:492-505)
    if (wave == 0) {
      double e2 = 0.0, reg = 0.0;
      for (int w = 0; w < NW; ++w) {
        e2 += s_red[0][w];
        reg += s_red[1][w];
      }
      const float err = (float)(0.5 * ((double)A.csq[item] - e2));
      int nz = 0;
      for (int ib = 0; ib < ncols; ib += 64) {
        const int i = ib + lane;
        const float xv = i < ncols ? x[i] : kInactive;
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
          const float xv = i < ncols ? x[i] : kInactive;
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
      if (lane == 0) {
        if (!fits) atomicMax(S.overflow, 1);
        S.out_cnt[item] = fits ? nz : -nz - 1;
        S.out_off[item] = (int64_t)off;
        S.st_na[item] = s_na;
        S.st_sweeps[item] = niters;
        S.st_conv[item] = conv;
        S.st_D[item] = (int64_t)s_D;
        S.st_U[item] = (int64_t)s_U;
        S.st_G[item] = nrows_read;  // (the engine reports the staging pass's G for the column)
        S.st_err[item] = err;
        S.st_obj[item] = err + (float)reg;
      }
    }
    __syncthreads();
  }
}

}  // namespace slimamd
