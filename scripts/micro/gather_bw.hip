// Microbenchmark: random-sector gather bandwidth on MI355X as a function of the
// contiguous granule per 16-lane group.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// GRAN = bytes contiguous per lane group (64, 128, 256); lanes per group = GRAN/4
template <int GRAN, int UNROLL>
__global__ __launch_bounds__(1024) void gather(const float* __restrict__ buf, uint32_t ngran, int iters, float* out) {
  const int lane = threadIdx.x & 63;
  constexpr int LPG = GRAN / 4;           // lanes per granule
  const int grp = lane / LPG, sub = lane % LPG;
  const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  float acc = 0;
  for (int it = 0; it < iters; ++it) {
    float v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      uint32_t g = mix(wid * 7919u + (uint32_t)(it * UNROLL + j) * 64u + (uint32_t)grp) % ngran;
      v[j] = buf[(size_t)g * LPG + sub];
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) acc += v[j];
  }
  if (acc == 123.456f) out[0] = acc;
}

// random whole-granule stores (the update pass of the tile kernel writes whole lines)
template <int GRAN, int UNROLL>
__global__ __launch_bounds__(1024) void scatter(float* __restrict__ buf, uint32_t ngran, int iters) {
  const int lane = threadIdx.x & 63;
  constexpr int LPG = GRAN / 4;
  const int grp = lane / LPG, sub = lane % LPG;
  const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      uint32_t g = mix(wid * 7919u + (uint32_t)(it * UNROLL + j) * 64u + (uint32_t)grp) % ngran;
      buf[(size_t)g * LPG + sub] = (float)it;
    }
  }
}

// every block gathers random 128-byte lines inside its OWN slab of `slab_lines` lines (the tile
// kernel's pattern: a workgroup only touches its residual slab); footprint = blocks x slab
template <int UNROLL>
__global__ __launch_bounds__(1024) void gather_slab(const float* __restrict__ buf, uint32_t slab_lines, int iters, float* out) {
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 5, sub = lane & 31;
  const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const float* base = buf + (size_t)blockIdx.x * slab_lines * 32;
  float acc = 0;
  for (int it = 0; it < iters; ++it) {
    float v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      uint32_t g = mix(wid * 7919u + (uint32_t)(it * UNROLL + j) * 64u + (uint32_t)grp) % slab_lines;
      v[j] = base[(size_t)g * 32 + sub];
    }
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) acc += v[j];
  }
  if (acc == 123.456f) out[0] = acc;
}

void run_slab(const float* buf, size_t slab_bytes, int blocks, int threads, int iters, float* out) {
  const uint32_t lines = (uint32_t)(slab_bytes / 128);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather_slab<16>), dim3(blocks), dim3(threads), 0, 0, buf, lines, 4, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gather_slab<16>), dim3(blocks), dim3(threads), 0, 0, buf, lines, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes_moved = (double)blocks * threads / 64 * iters * 16 * 256.0;
  printf("own slab %6.0f MiB x %4d blocks x %4d thr (footprint %6.1f GiB): %8.1f GB/s\n",
         slab_bytes / 1048576.0, blocks, threads, (double)slab_bytes * blocks / 1073741824.0,
         bytes_moved / ms / 1e6);
}

template <int GRAN, int UNROLL>
void run(const float* buf, size_t bytes, int blocks, int threads, int iters, float* out) {
  uint32_t ngran = (uint32_t)(bytes / GRAN);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<GRAN, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, ngran, 4, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<GRAN, UNROLL>), dim3(blocks), dim3(threads), 0, 0, buf, ngran, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double waves = (double)blocks * threads / 64;
  double bytes_moved = waves * iters * UNROLL * 256.0;
  printf("gran %3d B unroll %2d blocks %4d x %4d thr: %8.1f GB/s  (%.1f GB/s per block)\n", GRAN, UNROLL, blocks, threads,
         bytes_moved / ms / 1e6, bytes_moved / ms / 1e6 / blocks);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 's') {
    // footprint sweep: does the random-line rate depend on the slab a workgroup roams in?
    size_t total = (size_t)160 << 30;
    float* big; CK(hipMalloc(&big, total)); CK(hipMemset(big, 0, total));
    float* o; CK(hipMalloc(&o, 4));
    for (int blocks : {256, 512})
      for (size_t mb : {4, 32, 128, 320})
        run_slab(big, mb << 20, blocks, blocks == 256 ? 1024 : 512, 1000, o);
    // all blocks roaming over one region (the earlier measurement), 16 and 128 GiB
    run<128, 16>(big, (size_t)16 << 30, 256, 1024, 1000, o);
    run<128, 16>(big, (size_t)128 << 30, 256, 1024, 1000, o);
    return 0;
  }
  size_t bytes = (size_t)16 << 30;   // 16 GiB region
  float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  float* out; CK(hipMalloc(&out, 4));
  int iters = 2000;
  if (argc > 1 && argv[1][0] == 'c') {
    // PMC calibration: one gather launch and one scatter launch of exactly known bytes
    // (256 blocks x 16 waves x iters x 16 x 256 B), random 128-byte lines
    const int it = 1000;
    hipLaunchKernelGGL((gather<128, 16>), dim3(256), dim3(1024), 0, 0, buf, (uint32_t)(bytes / 128), it, out);
    hipLaunchKernelGGL((scatter<128, 16>), dim3(256), dim3(1024), 0, 0, buf, (uint32_t)(bytes / 128), it);
    CK(hipDeviceSynchronize());
    printf("calibration: each launch moves %.0f bytes\n", 256.0 * 16 * it * 16 * 256.0);
    return 0;
  }
  for (int blocks : {1, 256, 512}) {
    for (int threads : {256, 1024}) {
      run<64, 16>(buf, bytes, blocks, threads, iters, out);
      run<128, 16>(buf, bytes, blocks, threads, iters, out);
      run<256, 16>(buf, bytes, blocks, threads, iters, out);
    }
  }
  run<64, 4>(buf, bytes, 256, 1024, iters, out);
  run<64, 32>(buf, bytes, 256, 1024, iters / 2, out);
  run<128, 32>(buf, bytes, 256, 1024, iters / 2, out);
  // small region (64 MiB): TLB / cache effects
  run<64, 16>(buf, (size_t)64 << 20, 1, 1024, iters, out);
  run<64, 16>(buf, (size_t)64 << 20, 256, 1024, iters, out);
  return 0;
}
