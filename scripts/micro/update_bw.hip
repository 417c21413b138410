// Microbenchmark: what an UPDATE of a tile visit costs on MI355X, by the form of the
// write-back.  Not part of the product.
//
// The tile kernel (slim_amd/csrc/cd_tile.hpp) gathers one 128-byte residual line per nnz of the
// visited column slice (32 problems x 4 bytes of one user) and, when a coefficient changed,
// writes the slice back.  Usually ONE of the 32 problems changed.  This program reproduces the
// access pattern -- 256 workgroups x 16 wavefronts, every workgroup roaming in its own slab,
// NCH chunks of 1024 lines per visit, a fraction of the visits updating -- and times the
// write-back forms against each other:
//
//   0  no update at all (gather only: the floor)
//   1  whole lines: all 32 lanes store, chunks 0 .. n-3 gathered again first (round 3's kernel)
//   2  changed lanes only, plain 4-byte stores; chunks 0 .. n-3: the changed lane loads its own
//      old value first (4-byte gather)
//   3  changed lanes only, no-return float atomic add, workgroup scope; nothing gathered again
//   4  the same at agent scope
//   5  whole lines for the two chunks still in registers, atomics (workgroup scope) for the rest
//
// NCHG = how many of the 32 problems change on an updating visit (1 on C4, ~6 on C5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

#define LN(c, j) (*reinterpret_cast<float*>(bb + ((line(c, j) << 7) | qoff)))
template <int MODE, int NCH>
__global__ __launch_bounds__(1024, 4) void visits(float* __restrict__ buf, uint32_t slab_lines, int nvisits,
                                                  uint32_t upd_per_1024, int nchg, float* out) {
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 5, q = lane & 31;
  const int wave = threadIdx.x >> 6;
  float* base = buf + (size_t)blockIdx.x * slab_lines * 32;
  char* const bb = reinterpret_cast<char*>(base);
  const uint32_t qoff = (uint32_t)q << 2;
  float acc = 0;
  __shared__ float s_part[16];
  for (int v = 0; v < nvisits; ++v) {
    const uint32_t vkey = (blockIdx.x * 1000003u + (uint32_t)v) * 16u;
    // line of (chunk c, step j) for this wave / lane group (slab_lines is a power of two)
    const uint32_t lmask = slab_lines - 1u;
    uint32_t salt = 0;
    auto line = [&](int c, int j) -> uint32_t {
      const uint32_t b = mix((vkey + (uint32_t)c) * 4099u + (uint32_t)wave);
      return (b + (uint32_t)(j * 2 + grp) * 0x9E3779B1u + salt) & lmask;
    };
    float keep[2][32];
    // gather: all chunks, the last two stay in registers
    for (int c = 0; c < NCH - 2; ++c) {
      float r[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) { r[j] = LN(c, j); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int j = 0; j < 32; ++j) acc += r[j];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = NCH - 2 + k;
      if (c < 0) continue;
#pragma unroll
      for (int j = 0; j < 32; ++j) { keep[k][j] = LN(c, j); __builtin_amdgcn_sched_barrier(0); }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (NCH - 2 + k < 0) continue;
#pragma unroll
      for (int j = 0; j < 32; ++j) acc += keep[k][j];
    }
    // the exchange of the partial dots: one LDS round + barrier
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    float tot = 0;
    for (int w = 0; w < 16; ++w) tot += s_part[w];
    const bool upd = (mix(vkey + 77u) & 1023u) < upd_per_1024;
    // which problems change: nchg of them, starting at a visit-dependent lane
    const int q0 = (int)(mix(vkey + 5u) & 31u);
    const bool mine = ((q - q0) & 31) < nchg;
    const float d = tot * 1e-30f + 1e-6f;
    asm volatile("" : "+v"(salt));  // (the update recomputes its addresses, as the kernel does from the ids)
    if (MODE != 0 && upd) {
      if (MODE == 1) {
        _Pragma("unroll") for (int k = (NCH >= 2 ? 0 : 1); k < 2; ++k) { const int c = NCH - 2 + k;
#pragma unroll
          for (int j = 0; j < 32; ++j) { LN(c, j) = keep[k][j] - (mine ? d : 0.0f); __builtin_amdgcn_sched_barrier(0); }
        }
        for (int c = 0; c < NCH - 2; ++c) {
          float r[32];
#pragma unroll
          for (int j = 0; j < 32; ++j) { r[j] = LN(c, j); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
          for (int j = 0; j < 32; ++j) { LN(c, j) = r[j] - (mine ? d : 0.0f); __builtin_amdgcn_sched_barrier(0); }
        }
      } else if (MODE == 2) {
        _Pragma("unroll") for (int k = (NCH >= 2 ? 0 : 1); k < 2; ++k) { const int c = NCH - 2 + k;
          if (mine) {
#pragma unroll
            for (int j = 0; j < 32; ++j) { LN(c, j) = keep[k][j] - d; __builtin_amdgcn_sched_barrier(0); }
          }
        }
        for (int c = 0; c < NCH - 2; ++c) {
          float r[32];
          if (mine) {
#pragma unroll
            for (int j = 0; j < 32; ++j) { r[j] = LN(c, j); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < 32; ++j) { LN(c, j) = r[j] - d; __builtin_amdgcn_sched_barrier(0); }
          }
        }
      } else if (MODE == 3 || MODE == 4) {
        for (int c = 0; c < NCH; ++c) {
          if (mine) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
              if (MODE == 3)
                __hip_atomic_fetch_add(&LN(c, j), -d, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
              else
                __hip_atomic_fetch_add(&LN(c, j), -d, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      } else if (MODE == 5) {
        _Pragma("unroll") for (int k = (NCH >= 2 ? 0 : 1); k < 2; ++k) { const int c = NCH - 2 + k;
#pragma unroll
          for (int j = 0; j < 32; ++j) { LN(c, j) = keep[k][j] - (mine ? d : 0.0f); __builtin_amdgcn_sched_barrier(0); }
        }
        for (int c = 0; c < NCH - 2; ++c) {
          if (mine) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
              __hip_atomic_fetch_add(&LN(c, j), -d, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int NCH>
double run(float* buf, size_t slab_bytes, int blocks, int nvisits, uint32_t upd, int nchg, float* out,
           bool quiet = false) {
  const uint32_t lines = (uint32_t)(slab_bytes / 128);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((visits<MODE, NCH>), dim3(blocks), dim3(1024), 0, 0, buf, lines, 8, upd, nchg, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((visits<MODE, NCH>), dim3(blocks), dim3(1024), 0, 0, buf, lines, nvisits, upd, nchg, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double glines = (double)blocks * nvisits * NCH * 1024.0;
  if (!quiet)
    printf("mode %d chunks %d upd %4u/1024 changed %2d: %8.2f ms  %7.2f us/visit  gathered lines %6.1f G/s (%6.0f GB/s)\n",
           MODE, NCH, upd, nchg, ms, ms * 1e3 / nvisits, glines / ms / 1e6, glines * 128.0 / ms / 1e6);
  return ms;
}

template <int NCH>
void sweep(float* buf, size_t slab, int blocks, int nv, uint32_t upd, int nchg, float* out) {
  run<0, NCH>(buf, slab, blocks, nv, upd, nchg, out);
  run<1, NCH>(buf, slab, blocks, nv, upd, nchg, out);
  run<2, NCH>(buf, slab, blocks, nv, upd, nchg, out);
  run<3, NCH>(buf, slab, blocks, nv, upd, nchg, out);
  run<4, NCH>(buf, slab, blocks, nv, upd, nchg, out);
  run<5, NCH>(buf, slab, blocks, nv, upd, nchg, out);
}

int main(int argc, char** argv) {
  const int blocks = 256;
  const size_t slab = (size_t)32 << 20;  // 250K users x 128 B: a C4 member in a cluster of 4
  float* buf; CK(hipMalloc(&buf, slab * blocks)); CK(hipMemset(buf, 0, slab * blocks));
  float* out; CK(hipMalloc(&out, 4));
  if (argc > 2 && !strcmp(argv[1], "one")) {
    // one launch of one mode (for a PMC pass): update_bw one <mode> [chunks 3|6] [upd] [nchg]
    const int mode = atoi(argv[2]);
    const int nch = argc > 3 ? atoi(argv[3]) : 3;
    const uint32_t upd = argc > 4 ? (uint32_t)atoi(argv[4]) : 395u;
    const int nchg = argc > 5 ? atoi(argv[5]) : 1;
    const int nv = 2000;
#define ONE(M) \
  if (mode == M) { if (nch == 3) run<M, 3>(buf, slab, blocks, nv, upd, nchg, out); else run<M, 6>(buf, slab, blocks, nv, upd, nchg, out); }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5)
    printf("gathered bytes per timed launch (dot pass only): %.4e\n", (double)blocks * nv * nch * 1024.0 * 128.0);
    return 0;
  }
  printf("# C4-like: 3 chunks per visit, 38.6 %% of the visits update, 1 of 32 problems changed\n");
  sweep<3>(buf, slab, blocks, 2000, 395u, 1, out);
  printf("# every visit updates, 1 changed\n");
  sweep<3>(buf, slab, blocks, 2000, 1024u, 1, out);
  printf("# C5-like: 6 chunks per visit, 84 %% update, 6 of 32 changed\n");
  sweep<6>(buf, slab, blocks, 1000, 860u, 6, out);
  printf("# 6 chunks, every visit updates, 1 changed\n");
  sweep<6>(buf, slab, blocks, 1000, 1024u, 1, out);
  printf("# 1 chunk (short columns), 55 %% update, 1 changed\n");
  sweep<1>(buf, slab, blocks, 4000, 563u, 1, out);
  return 0;
}
