// engine.hip -- HBM staging of the training matrix and the CD solve driver.
//
// Staging is the device form of CreateTrainingMatrix
// (/root/reference/src/libslim/setup.c:109-135): the caller's CSR is copied to
// HBM once (or adopted if it already lives there), the column view is built by
// a stable radix sort on the item id (keeps user ids ascending inside every
// column, which is what gk_csr_CreateIndex + slim_csr_SortIndices guarantee,
// setup.c:128,132), and the column norms are reduced one wavefront per column
// (gk_csr_ComputeNorms, setup.c:130).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <numeric>
#include <memory>
#include <string>
#include <thread>

#include <rocprim/device/device_radix_sort.hpp>

#include "tile_inst.hpp"
#include "gram_inst.hpp"
#include "gramr_inst.hpp"
#include "cd_wave.hpp"
#include "engine.hpp"
#include "host_csr.hpp"

// ---- the opaque handle ---------------------------------------------------------
inline uint64_t slimgpu_next_uid() {
  static std::atomic<uint64_t> n{0};
  return ++n;
}

struct slimgpu_matrix {
  const uint64_t uid = slimgpu_next_uid();  // (a handle's identity beyond its address)
  int device = 0;
  hipStream_t stream = nullptr;
  int32_t nrows = 0, ncols = 0;
  int64_t nnz = 0;
  bool binary = false;
  bool owns_csr = false;
  bool exact_gram = false;  // ratings are not small integers: aTy sums formed in a fixed order
  bool nonpositive = false;  // some rating is <= 0: a co-rating sum can cancel to exactly 0
  // CSR
  int64_t* d_rowptr = nullptr;
  int32_t* d_rowind = nullptr;
  float* d_rowval = nullptr;
  // CSC + per-column scalars
  int64_t* d_colptr = nullptr;
  int32_t* d_colind = nullptr;
  float* d_colval = nullptr;
  float* d_cnorm = nullptr;
  float* d_csq = nullptr;
  std::vector<int64_t> h_cost;  // scheduling proxy per column (Gram work G)
  std::vector<int64_t> h_rowptr;  // host copy, fetched on first clustered solve
  // column slice boundaries for tile clusters of size K (index log2 K), built on demand
  int32_t* d_ubounds[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t* d_csplit[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int32_t max_range_rows[6] = {0, 0, 0, 0, 0, 0};
  // the G builder's user passes (learn_cd, gram_passes): 32 * np user ranges of equal nnz and the
  // slice boundaries of every column over them, [ncols][32 * np + 1]
  int32_t* d_gubounds = nullptr;
  int64_t* d_gcsplit = nullptr;
  int gsplit_np = 0;
  int32_t gsplit_max_rows = 0;
  double setup_ms = 0;
  int num_cus = 256;
  // workspace reused by successive solves
  struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
  };
  // copies of this matrix on other devices of the node (multi_gpu.cpp); owned by this handle
  std::vector<slimgpu_matrix*> replicas;
  // screen sums (a_i . y for every column i and every item of a tile) of the most recent tile
  // solve, reusable by the next solve of the same columns in the same geometry (model-selection
  // grids: slim_mselect.c:99-113 solves every (l1, l2) pair over the same R)
  Buf ws_gram;
  std::vector<int32_t> gram_order;  // work list the sums belong to (empty: none recorded)
  int gram_geom[6] = {0, 0, 0, 0, 0, 0};  // tileP, K, K_hi, nheavy, shard count, shard index
  // G = R^T R (item-space CD, cd_gram.hpp): [ncols][G_ld] floats, built on the first solve that
  // takes that path and kept with the handle
  Buf ws_G, ws_nunion;
  int64_t G_ld = 0;
  bool G_ready = false;
  double G_build_ms = 0, G_alloc_ms = 0, G_sums_ms = 0, G_sums_kernel_ms = 0, G_pack_ms = 0;
  // G as byte planes in popularity order (gram_pack.hpp), what cd_gramr.hpp streams: built from
  // the float G right after it, when every entry is a non-negative integer below 2^24
  Buf ws_Glo, ws_Ghi, ws_Ghi2, ws_Gbase, ws_Gdiag, ws_Gmeta, ws_hioff, ws_hi2off, ws_hik, ws_hi2k, ws_rankof, ws_itemof;
  int64_t Gp_ldb = 0;
  int32_t Gp_nchunks = 0;
  bool Gp_ready = false, Gp_tried = false;
  bool Gf_dropped = false;          // the floats of G were freed once the planes stood (drop_float_gram)
  double Gp_bytes_per_row = 0;      // average bytes of a packed row (lo + hi + hi2)
  int expect_solves = 0;            // announced by the caller (model-selection grids)
  std::vector<int32_t> last_order;  // work list of the most recent solve
  Buf ws_order, ws_cnt, ws_off, ws_stat_i, ws_stat_l, ws_stat_f, ws_misc, ws_arena_i, ws_arena_v,
      ws_slab, ws_xslab, ws_ulist, ws_trace, ws_mailbox, ws_part, ws_icolptr, ws_icolind,
      ws_icolval;
  Buf ws_tkeys[2], ws_tpay[2], ws_ttmp;  // the row view of a resident model (transpose_on_device)
};

// A learned model resident in HBM (SLIMGPU_LearnResident): the column view as SaveModel lays it out
// (estimate.c:570-593; ids ascending in every column) and the row view, formed on the device.
struct slimgpu_model {
  int device = 0;
  int32_t n = 0;      // nrows = ncols of W
  int64_t nnz = 0;
  int64_t* d_colptr = nullptr;
  int32_t* d_colind = nullptr;
  float* d_colval = nullptr;
  int64_t* d_rowptr = nullptr;
  int32_t* d_rowind = nullptr;
  float* d_rowval = nullptr;
  // g of every problem as the solve that produced this model left it (cd_gramr.hpp, g_save): the
  // next solve of the same problems on the same handle starts from it when only l2 moved (the same
  // l1 = the same active sets) instead of re-folding this model into g row by row.  The buffer travels
  // down the chain of a grid's models (the next solve updates it in place and takes it over).
  mutable float* d_gsave = nullptr;
  mutable bool gsave_valid = false;
  int64_t gsave_stride = 0;
  double gsave_l1 = 0;
  uint64_t gsave_owner = 0;  // uid of the matrix handle
  // a fetch to the host running beside the next solve (model_fetch_begin)
  std::thread fetcher;
  bool fetch_begun = false;
  slim_csr_t* fetched = nullptr;
  int32_t fetch_status = 0;
  std::string fetch_error;
  double fetch_ms = 0;
};

namespace slimamd {

namespace {

thread_local slimgpu_stats_t g_stats;
thread_local ColumnStats g_colstats;

struct HipError {
  hipError_t code;
  std::string where;
};

struct InputError {  // malformed caller data: SLIM_ERROR_INPUT
  std::string msg;
};

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) throw HipError{_e, std::string(#expr)};                  \
  } while (0)

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int32_t status_of(const HipError& e) {
  return e.code == hipErrorOutOfMemory ? SLIM_ERROR_MEMORY : SLIM_ERROR;
}

void report(const HipError& e, const char* what) {
  set_error(std::string(what) + ": HIP error '" + hipGetErrorString(e.code) + "' in " + e.where +
            " -- the SLIM CD path needs a gfx950 GPU; there is no CPU fallback");
}

template <class T>
T* dev_alloc(size_t n) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, sizeof(T) * (n ? n : 1)));
  return static_cast<T*>(p);
}

// grow-only workspace buffer.  `evict` (optional): the handle whose screen-sum cache is given
// up when the device is out of memory -- the cache only saves a pass, nothing depends on it.
void drop_screen_cache(slimgpu_matrix* m);
template <class T>
T* ws_get(slimgpu_matrix::Buf& b, size_t n, slimgpu_matrix* evict = nullptr) {
  const size_t need = sizeof(T) * (n ? n : 1);
  if (b.bytes < need) {
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    hipError_t e = hipMalloc(&b.p, need);
    if (e == hipErrorOutOfMemory && evict && evict->ws_gram.p && &b != &evict->ws_gram) {
      (void)hipGetLastError();
      drop_screen_cache(evict);
      e = hipMalloc(&b.p, need);
    }
    if (e != hipSuccess) {
      b.p = nullptr;
      throw HipError{e, "hipMalloc(workspace)"};
    }
    b.bytes = need;
  }
  return static_cast<T*>(b.p);
}

// ---- staging kernels -----------------------------------------------------------

__global__ void k_max_index(const int32_t* __restrict__ ind, int64_t n, int32_t* out) {
  int32_t m = -1;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (int64_t)gridDim.x * blockDim.x)
    m = max(m, ind[k]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// Input checks done while staging (the reference trusts its caller; a GPU kernel fed a
// malformed CSR corrupts memory instead of crashing cleanly).  Bits of the flag word:
constexpr int kBadRowptr = 1;   // rowptr not non-decreasing / not ending at nnz
constexpr int kBadColumn = 2;   // item id outside [0, ncols)
constexpr int kDupEntry = 4;    // the same (user, item) pair twice

// key = item id, payload = (user id << 32 | value bits); one wavefront per row
__global__ void k_pack_rows(int32_t nrows, int32_t ncols, int64_t nnz,
                            const int64_t* __restrict__ rowptr,
                            const int32_t* __restrict__ rowind,
                            const float* __restrict__ rowval, uint32_t* __restrict__ keys,
                            uint64_t* __restrict__ payload, int32_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < nrows; u += nwaves) {
    int64_t s = rowptr[u], e = rowptr[u + 1];
    if (s < 0 || e < s || e > nnz || (u == 0 && s != 0)) {
      if (lane == 0) atomicOr(flags, kBadRowptr);
      continue;  // never index with a bad offset
    }
    for (int64_t k = s + lane; k < e; k += 64) {
      if ((uint32_t)rowind[k] >= (uint32_t)ncols) atomicOr(flags, kBadColumn);
      keys[k] = (uint32_t)rowind[k];
      const uint32_t bits = rowval ? __float_as_uint(rowval[k]) : 0x3F800000u;
      payload[k] = ((uint64_t)(uint32_t)u << 32) | bits;
    }
  }
}

__global__ void k_unpack_cols(int64_t nnz, const uint64_t* __restrict__ payload,
                              int32_t* __restrict__ colind, float* __restrict__ colval) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t p = payload[k];
    colind[k] = (int32_t)(p >> 32);
    if (colval) colval[k] = __uint_as_float((uint32_t)p);
  }
}

// after the stable sort a repeated (user, item) pair is two adjacent equal entries of a column
__global__ void k_check_dups(int64_t nnz, const uint32_t* __restrict__ keys,
                             const uint64_t* __restrict__ payload, int32_t* __restrict__ flags) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; k < nnz;
       k += (int64_t)gridDim.x * blockDim.x)
    if (keys[k] == keys[k - 1] && (payload[k] >> 32) == (payload[k - 1] >> 32))
      atomicOr(flags, kDupEntry);
}

// colptr from the sorted keys: entry k opens every column in (key[k-1], key[k]]
__global__ void k_col_offsets(int64_t nnz, int32_t ncols, const uint32_t* __restrict__ keys,
                              int64_t* __restrict__ colptr) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= nnz;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo = k == 0 ? 0 : (int64_t)keys[k - 1] + 1;
    const int64_t hi = k == nnz ? (int64_t)ncols : (int64_t)keys[k];
    for (int64_t c = lo; c <= hi; ++c) colptr[c] = k;
  }
}

// one wavefront per column: fp32 sum of squares, norm, and the Gram work G
__global__ void k_col_scalars(int32_t ncols, const int64_t* __restrict__ colptr,
                              const int32_t* __restrict__ colind,
                              const float* __restrict__ colval,
                              const int64_t* __restrict__ rowptr, float* __restrict__ csq,
                              float* __restrict__ cnorm, int64_t* __restrict__ cost,
                              int32_t* __restrict__ inexact) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < ncols; c += nwaves) {
    const int64_t s = colptr[c], e = colptr[c + 1];
    float ss = 0.0f;
    int64_t g = 0;
    for (int64_t k = s + lane; k < e; k += 64) {
      const float v = colval ? colval[k] : 1.0f;
      ss += v * v;
      // float sums of products of small integers are exact in any order; anything else
      // (fractional ratings, huge values) must not be accumulated with float atomics
      if (v != rintf(v) || fabsf(v) > 2048.0f) atomicOr(inexact, 1);
      // (bit 1: a rating <= 0 -- co-rating sums can then cancel to 0, which matters to FSLIM's
      // candidate rule, neighbors.c:46-60)
      if (!(v > 0.0f)) atomicOr(inexact, 2);
      const int32_t u = colind[k];
      g += rowptr[u + 1] - rowptr[u];
    }
    for (int off = 32; off > 0; off >>= 1) {
      ss += __shfl_xor(ss, off);
      g += __shfl_xor(g, off);
    }
    if (lane == 0) {
      if (!colval) ss = (float)(e - s);
      if (!(ss < 16777216.0f)) atomicOr(inexact, 1);  // a dot product can reach |a_i||a_j|
      csq[c] = ss;
      cnorm[c] = sqrtf(ss);
      cost[c] = g;
    }
  }
}

// csplit[c][j] = first position of column c whose user id is >= ubounds[j]
__global__ void k_col_split(int32_t ncols, int32_t K, const int32_t* __restrict__ ubounds,
                            const int64_t* __restrict__ colptr,
                            const int32_t* __restrict__ colind, int64_t* __restrict__ csplit) {
  const int64_t n = (int64_t)ncols * (K + 1);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = (int32_t)(t / (K + 1)), j = (int32_t)(t % (K + 1));
    int64_t lo = colptr[c], hi = colptr[c + 1];
    if (j == K) {
      lo = hi;
    } else if (j > 0) {
      const int32_t ub = ubounds[j];
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (colind[mid] < ub) lo = mid + 1; else hi = mid;
      }
    }
    csplit[t] = lo;
  }
}

// one wavefront per column: the entries a launch left in its arena, into column order
__global__ void k_gather_columns(int32_t ncols, const int64_t* __restrict__ colptr,
                                 const int64_t* __restrict__ src_off, const int32_t* __restrict__ src_ind,
                                 const float* __restrict__ src_val, int32_t* __restrict__ colind,
                                 float* __restrict__ colval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < ncols; c += nwaves) {
    const int64_t d = colptr[c], n = colptr[c + 1] - d, s = src_off[c];
    for (int64_t k = lane; k < n; k += 64) {
      colind[d + k] = src_ind[s + k];
      colval[d + k] = src_val[s + k];
    }
  }
}
void drop_screen_cache(slimgpu_matrix* m) {
  if (m->ws_gram.p) (void)hipFree(m->ws_gram.p);
  m->ws_gram.p = nullptr;
  m->ws_gram.bytes = 0;
  m->gram_order.clear();
}

int grid_for(int64_t n, int block, int cap_blocks) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap_blocks) g = cap_blocks;
  return (int)g;
}

void pick_device(slimgpu_matrix* m, const LearnOptions& opt) {
  (void)hipGetLastError();  // a failure of an earlier call must not be reported by this one
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (count <= 0) throw HipError{hipErrorNoDevice, "hipGetDeviceCount"};
  if (opt.device >= 0) {
    HIP_TRY(hipSetDevice(opt.device));
    m->device = opt.device;
  } else {
    HIP_TRY(hipGetDevice(&m->device));
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, m->device));
  m->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
}

// Build CSC + scalars from the device CSR of m (nrows/nnz/ncols already set).
void build_column_view(slimgpu_matrix* m) {
  hipStream_t st = m->stream;
  const int64_t nnz = m->nnz;
  m->d_colptr = dev_alloc<int64_t>((size_t)m->ncols + 1);
  // (+ 64 entries of slack: the G builder reads whole 128-byte lines of a column slice, cd_tile.hpp)
  m->d_colind = dev_alloc<int32_t>((size_t)nnz + 64);
  m->d_colval = m->binary ? nullptr : dev_alloc<float>((size_t)nnz);
  m->d_cnorm = dev_alloc<float>((size_t)m->ncols);
  m->d_csq = dev_alloc<float>((size_t)m->ncols);
  int64_t* d_cost = dev_alloc<int64_t>((size_t)m->ncols);

  if (nnz > 0) {
    if (nnz > 0xFFFFFFF0ll) throw HipError{hipErrorInvalidValue, "nnz >= 2^32 not supported"};
    uint32_t* keys_in = dev_alloc<uint32_t>((size_t)nnz);
    uint32_t* keys_out = dev_alloc<uint32_t>((size_t)nnz);
    uint64_t* pay_in = dev_alloc<uint64_t>((size_t)nnz);
    uint64_t* pay_out = dev_alloc<uint64_t>((size_t)nnz);
    const int cap = m->num_cus * 16;
    int32_t* d_flags = dev_alloc<int32_t>(1);
    HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_pack_rows, dim3(grid_for((int64_t)m->nrows * 64, 256, cap)), dim3(256), 0,
                       st, m->nrows, m->ncols, nnz, m->d_rowptr, m->d_rowind, m->d_rowval, keys_in,
                       pay_in, d_flags);
    HIP_TRY(hipGetLastError());
    int32_t h_flags = 0;
    HIP_TRY(hipMemcpyAsync(&h_flags, d_flags, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    auto release_tmp = [&]() {
      (void)hipFree(d_flags);
      (void)hipFree(keys_in);
      (void)hipFree(keys_out);
      (void)hipFree(pay_in);
      (void)hipFree(pay_out);
      (void)hipFree(d_cost);
    };
    if (h_flags) {  // checked before the sort: its bit count assumes ids < ncols
      release_tmp();
      throw InputError{h_flags & kBadRowptr
                           ? "rowptr is not a non-decreasing offset array ending at nnz"
                           : "item id outside [0, ncols)"};
    }
    unsigned bits = 1;
    while ((1ull << bits) < (unsigned long long)m->ncols) ++bits;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, pay_in, pay_out,
                                      (size_t)nnz, 0u, bits, st));
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, pay_in, pay_out,
                                      (size_t)nnz, 0u, bits, st));
    hipLaunchKernelGGL(k_check_dups, dim3(grid_for(nnz, 256, cap)), dim3(256), 0, st, nnz,
                       keys_out, pay_out, d_flags);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_unpack_cols, dim3(grid_for(nnz, 256, cap)), dim3(256), 0, st, nnz,
                       pay_out, m->d_colind, m->d_colval);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_col_offsets, dim3(grid_for(nnz + 1, 256, cap)), dim3(256), 0, st, nnz,
                       m->ncols, keys_out, m->d_colptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&h_flags, d_flags, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipFree(tmp));
    HIP_TRY(hipFree(keys_in));
    HIP_TRY(hipFree(keys_out));
    HIP_TRY(hipFree(pay_in));
    HIP_TRY(hipFree(pay_out));
    HIP_TRY(hipFree(d_flags));
    if (h_flags & kDupEntry) {
      // the reference walks duplicates as separate entries (norm from v1^2 + v2^2, dots from
      // v1 + v2): not a well-defined problem, and two lanes updating one residual race here
      (void)hipFree(d_cost);
      throw InputError{"duplicate (user, item) entries in the rating matrix (SLIM_GPU_DUPLICATES=sum "
                       "merges them while a host matrix is staged)"};
    }
  } else {
    HIP_TRY(hipMemsetAsync(m->d_colptr, 0, sizeof(int64_t) * ((size_t)m->ncols + 1), st));
  }
  int32_t* d_inexact = dev_alloc<int32_t>(1);
  HIP_TRY(hipMemsetAsync(d_inexact, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(k_col_scalars, dim3(grid_for((int64_t)m->ncols * 64, 256, m->num_cus * 16)),
                     dim3(256), 0, st, m->ncols, m->d_colptr, m->d_colind, m->d_colval,
                     m->d_rowptr, m->d_csq, m->d_cnorm, d_cost, d_inexact);
  HIP_TRY(hipGetLastError());
  int32_t h_inexact = 0;
  HIP_TRY(hipMemcpyAsync(&h_inexact, d_inexact, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  m->h_cost.resize((size_t)m->ncols);
  HIP_TRY(hipMemcpyAsync(m->h_cost.data(), d_cost, sizeof(int64_t) * (size_t)m->ncols,
                         hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipFree(d_cost));
  HIP_TRY(hipFree(d_inexact));
  m->exact_gram = (h_inexact & 1) != 0;
  m->nonpositive = (h_inexact & 2) != 0;
}

// The other view of a square sparse matrix held as (ptr, ind, val) with n rows, on the device: the
// staging pass's own steps (key = id, payload = (source row << 32 | value), stable radix sort by key)
// -- entries of a destination row keep the order of their source rows, i.e. ascending ids, the arrays
// csr_build_index (host_csr.cpp) forms.  The outputs are allocated here; temporaries live with the handle.
void transpose_on_device(slimgpu_matrix* m, int32_t n, int64_t nnz, const int64_t* d_ptr,
                         const int32_t* d_ind, const float* d_val, int64_t** o_ptr, int32_t** o_ind,
                         float** o_val) {
  hipStream_t st = m->stream;
  *o_ptr = dev_alloc<int64_t>((size_t)n + 1);
  *o_ind = dev_alloc<int32_t>((size_t)std::max<int64_t>(nnz, 1));
  *o_val = dev_alloc<float>((size_t)std::max<int64_t>(nnz, 1));
  if (nnz <= 0) {
    HIP_TRY(hipMemsetAsync(*o_ptr, 0, sizeof(int64_t) * ((size_t)n + 1), st));
    return;
  }
  if (nnz > 0xFFFFFFF0ll) throw HipError{hipErrorInvalidValue, "model nnz >= 2^32 not supported"};
  uint32_t* keys_in = ws_get<uint32_t>(m->ws_tkeys[0], (size_t)nnz, m);
  uint32_t* keys_out = ws_get<uint32_t>(m->ws_tkeys[1], (size_t)nnz, m);
  uint64_t* pay_in = ws_get<uint64_t>(m->ws_tpay[0], (size_t)nnz, m);
  uint64_t* pay_out = ws_get<uint64_t>(m->ws_tpay[1], (size_t)nnz, m);
  const int cap = m->num_cus * 16;
  int32_t* d_flags = ws_get<int32_t>(m->ws_misc, 16, m);  // (the solver's scalars: read back already)
  hipLaunchKernelGGL(k_pack_rows, dim3(grid_for((int64_t)n * 64, 256, cap)), dim3(256), 0, st, n, n, nnz, d_ptr,
                     d_ind, d_val, keys_in, pay_in, d_flags);
  HIP_TRY(hipGetLastError());
  unsigned bits = 1;
  while ((1ull << bits) < (unsigned long long)n) ++bits;
  size_t tmp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, pay_in, pay_out, (size_t)nnz, 0u, bits, st));
  void* tmp = ws_get<uint8_t>(m->ws_ttmp, tmp_bytes ? tmp_bytes : 1, m);
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, pay_in, pay_out, (size_t)nnz, 0u, bits, st));
  hipLaunchKernelGGL(k_unpack_cols, dim3(grid_for(nnz, 256, cap)), dim3(256), 0, st, nnz, pay_out, *o_ind, *o_val);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_col_offsets, dim3(grid_for(nnz + 1, 256, cap)), dim3(256), 0, st, nnz, n, keys_out, *o_ptr);
  HIP_TRY(hipGetLastError());
}

// user ranges of equal nnz + per-column slice boundaries for clusters of size K = 1 << lg
void ensure_cluster_split(slimgpu_matrix* m, int lg) {
  if (m->d_csplit[lg]) return;
  const int K = 1 << lg;
  std::vector<int32_t> ub((size_t)K + 1, 0);
  ub[K] = m->nrows;
  if (K > 1) {
    if (m->h_rowptr.empty()) {
      m->h_rowptr.resize((size_t)m->nrows + 1);
      HIP_TRY(hipMemcpy(m->h_rowptr.data(), m->d_rowptr, sizeof(int64_t) * ((size_t)m->nrows + 1),
                        hipMemcpyDeviceToHost));
    }
    for (int j = 1; j < K; ++j) {
      const int64_t want = m->nnz / K * j;
      ub[j] = (int32_t)(std::lower_bound(m->h_rowptr.begin(), m->h_rowptr.end(), want) -
                        m->h_rowptr.begin());
      ub[j] = std::min(std::max(ub[j], ub[j - 1]), m->nrows);
    }
  }
  int32_t mx = 1;
  for (int j = 0; j < K; ++j) mx = std::max(mx, ub[j + 1] - ub[j]);
  m->max_range_rows[lg] = mx;
  m->d_ubounds[lg] = dev_alloc<int32_t>((size_t)K + 1);
  HIP_TRY(hipMemcpy(m->d_ubounds[lg], ub.data(), sizeof(int32_t) * ((size_t)K + 1),
                    hipMemcpyHostToDevice));
  m->d_csplit[lg] = dev_alloc<int64_t>((size_t)m->ncols * (K + 1));
  hipLaunchKernelGGL(k_col_split, dim3(grid_for((int64_t)m->ncols * (K + 1), 256, m->num_cus * 8)),
                     dim3(256), 0, m->stream, m->ncols, K, m->d_ubounds[lg], m->d_colptr,
                     m->d_colind, m->d_csplit[lg]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(m->stream));
}

// the same for the G builder's user passes: nr = 32 * np ranges (np passes of a cluster of 32)
void ensure_gram_split(slimgpu_matrix* m, int np) {
  if (m->gsplit_np == np && m->d_gcsplit) return;
  (void)hipFree(m->d_gubounds);
  (void)hipFree(m->d_gcsplit);
  m->d_gubounds = nullptr;
  m->d_gcsplit = nullptr;
  m->gsplit_np = 0;
  const int nr = 32 * np;
  std::vector<int32_t> ub((size_t)nr + 1, 0);
  ub[(size_t)nr] = m->nrows;
  if (m->h_rowptr.empty()) {
    m->h_rowptr.resize((size_t)m->nrows + 1);
    HIP_TRY(hipMemcpy(m->h_rowptr.data(), m->d_rowptr, sizeof(int64_t) * ((size_t)m->nrows + 1),
                      hipMemcpyDeviceToHost));
  }
  for (int j = 1; j < nr; ++j) {
    const int64_t want = (int64_t)((double)m->nnz / nr * j);
    ub[(size_t)j] = (int32_t)(std::lower_bound(m->h_rowptr.begin(), m->h_rowptr.end(), want) - m->h_rowptr.begin());
    ub[(size_t)j] = std::min(std::max(ub[(size_t)j], ub[(size_t)j - 1]), m->nrows);
  }
  int32_t mx = 1;
  for (int j = 0; j < nr; ++j) mx = std::max(mx, ub[(size_t)j + 1] - ub[(size_t)j]);
  m->gsplit_max_rows = mx;
  m->d_gubounds = dev_alloc<int32_t>((size_t)nr + 1);
  HIP_TRY(hipMemcpy(m->d_gubounds, ub.data(), sizeof(int32_t) * ((size_t)nr + 1), hipMemcpyHostToDevice));
  m->d_gcsplit = dev_alloc<int64_t>((size_t)m->ncols * ((size_t)nr + 1));
  hipLaunchKernelGGL(k_col_split, dim3(grid_for((int64_t)m->ncols * (nr + 1), 256, m->num_cus * 8)), dim3(256), 0,
                     m->stream, m->ncols, nr, m->d_gubounds, m->d_colptr, m->d_colind, m->d_gcsplit);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(m->stream));
  m->gsplit_np = np;
}

void destroy(slimgpu_matrix* m) {
  if (!m) return;
  for (slimgpu_matrix* r : m->replicas) destroy(r);
  m->replicas.clear();
  (void)hipSetDevice(m->device);
  if (m->owns_csr) {
    (void)hipFree(m->d_rowptr);
    (void)hipFree(m->d_rowind);
    (void)hipFree(m->d_rowval);
  }
  (void)hipFree(m->d_colptr);
  (void)hipFree(m->d_colind);
  (void)hipFree(m->d_colval);
  (void)hipFree(m->d_cnorm);
  (void)hipFree(m->d_csq);
  for (int k = 0; k < 6; ++k) {
    (void)hipFree(m->d_ubounds[k]);
    (void)hipFree(m->d_csplit[k]);
  }
  (void)hipFree(m->d_gubounds);
  (void)hipFree(m->d_gcsplit);
  for (slimgpu_matrix::Buf* b :
       {&m->ws_order, &m->ws_cnt, &m->ws_off, &m->ws_stat_i, &m->ws_stat_l, &m->ws_stat_f,
        &m->ws_misc, &m->ws_arena_i, &m->ws_arena_v, &m->ws_slab, &m->ws_xslab, &m->ws_ulist,
        &m->ws_trace, &m->ws_mailbox, &m->ws_part, &m->ws_icolptr, &m->ws_icolind, &m->ws_icolval,
        &m->ws_gram, &m->ws_G, &m->ws_nunion, &m->ws_Glo, &m->ws_Ghi, &m->ws_Ghi2, &m->ws_Gbase, &m->ws_Gdiag, &m->ws_Gmeta, &m->ws_hioff,
        &m->ws_hi2off, &m->ws_hik, &m->ws_hi2k, &m->ws_rankof, &m->ws_itemof, &m->ws_tkeys[0], &m->ws_tkeys[1],
        &m->ws_tpay[0], &m->ws_tpay[1], &m->ws_ttmp})
    if (b->p) (void)hipFree(b->p);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

}  // namespace

slimgpu_stats_t& last_stats() { return g_stats; }
ColumnStats& last_column_stats() { return g_colstats; }

LearnOptions decode_options(const int32_t* io, const double* dopt) {
  LearnOptions o;
  auto geti = [&](int idx, int32_t def) { return (!io || io[idx] == -1) ? def : io[idx]; };
  auto getd = [&](int idx, double def) { return (!dopt || dopt[idx] == -1) ? def : dopt[idx]; };
  o.nthreads = geti(SLIM_OPTION_NTHREADS, 1);
  o.nnbrs = geti(SLIM_OPTION_NNBRS, 0);
  o.simtype = geti(SLIM_OPTION_SIMTYPE, SLIM_SIMTYPE_COS);
  o.dbglvl = geti(SLIM_OPTION_DBGLVL, 0);
  o.algo = geti(SLIM_OPTION_ALGO, SLIM_ALGO_CD);
  o.ordered = geti(SLIM_OPTION_ORDERED, 0);
  o.maxniters = geti(SLIM_OPTION_MAXNITERS, 10000);
  o.l1r = getd(SLIM_OPTION_L1R, 1.0);
  o.l2r = getd(SLIM_OPTION_L2R, 1.0);
  o.optTol = getd(SLIM_OPTION_OPTTOL, 1e-7);
  o.col_begin = geti(SLIM_OPTION_GPU_COLBEGIN, 0);
  o.col_end = geti(SLIM_OPTION_GPU_COLEND, -1);
  o.seed = (uint32_t)geti(SLIM_OPTION_GPU_SEED, 1);
  o.device = geti(SLIM_OPTION_GPU_DEVICE, -1);
  o.kernel = geti(SLIM_OPTION_GPU_KERNEL, SLIMGPU_KERNEL_AUTO);
  o.cluster = geti(SLIM_OPTION_GPU_CLUSTER, 0);
  o.heavy_tiles = geti(SLIM_OPTION_GPU_HEAVYTILES, -1);
  o.heavy_cluster = geti(SLIM_OPTION_GPU_HEAVYCLUSTER, 0);
  o.ngpus = std::max(1, geti(SLIM_OPTION_GPU_NGPUS, 1));
  o.shard_count = std::max(1, geti(SLIM_OPTION_GPU_SHARDCOUNT, 1));
  o.shard_index = geti(SLIM_OPTION_GPU_SHARDINDEX, 0);
  return o;
}

int32_t device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

slimgpu_matrix_t* matrix_from_host(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                                   const float* rowval, const LearnOptions& opt,
                                   int32_t* status) {
  if (nrows < 0 || !rowptr || (rowptr[nrows] > 0 && !rowind)) {
    set_error("SLIMGPU_MatrixFromHost: bad CSR arguments");
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  const double t0 = now_ms();
  // Repeated (user, item) pairs.  The reference copies them verbatim (setup.c:119-126) and then
  // treats them inconsistently -- the LAST value of a pair is the target y[u] (estimate.c:406-408),
  // dot products take both entries, the norm is v1^2 + v2^2 -- and two lanes updating one
  // residual would race here, so the engine rejects them (default) or, with
  // SLIM_GPU_DUPLICATES=sum, merges them before staging: the values of a pair are added
  // (an implicit-feedback matrix, rowval == NULL, keeps one entry and stays binary).
  std::vector<int64_t> mptr;
  std::vector<int32_t> mind;
  std::vector<float> mval;
  try {  // (host vectors of nnz entries: a failed allocation is SLIM_ERROR_MEMORY, not a terminate)
  if (const char* e = std::getenv("SLIM_GPU_DUPLICATES"); e && std::strcmp(e, "sum") == 0 && nrows > 0) {
    bool any = false;
    std::vector<std::pair<int32_t, float>> row;
    mptr.assign((size_t)nrows + 1, 0);
    mind.reserve((size_t)rowptr[nrows]);
    if (rowval) mval.reserve((size_t)rowptr[nrows]);
    for (int32_t u = 0; u < nrows; ++u) {
      const int64_t s0 = rowptr[u], e0 = rowptr[u + 1];
      row.clear();
      for (int64_t k = s0; k < e0; ++k) row.emplace_back(rowind[k], rowval ? rowval[k] : 1.0f);
      bool sorted = true;
      for (size_t k = 1; k < row.size(); ++k) sorted = sorted && row[k - 1].first < row[k].first;
      if (!sorted) {  // (strictly ascending rows hold no repeated pair and are copied as they are)
        std::stable_sort(row.begin(), row.end(),
                         [](const auto& a, const auto& b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t k = 0; k < row.size(); ++k) {
          if (w > 0 && row[w - 1].first == row[k].first) {
            if (rowval) row[w - 1].second += row[k].second;
            any = true;
          } else {
            row[w++] = row[k];
          }
        }
        row.resize(w);
      }
      for (const auto& pr : row) {
        mind.push_back(pr.first);
        if (rowval) mval.push_back(pr.second);
      }
      mptr[(size_t)u + 1] = (int64_t)mind.size();
    }
    if (any) {  // stage the merged copy (rows now ascending by item id)
      rowptr = reinterpret_cast<const ssize_t*>(mptr.data());
      rowind = mind.data();
      if (rowval) rowval = mval.data();
    }
  }
  } catch (const std::bad_alloc&) {
    set_error("SLIMGPU_MatrixFromHost: out of host memory while merging repeated pairs");
    if (status) *status = SLIM_ERROR_MEMORY;
    return nullptr;
  }
  slimgpu_matrix* m = nullptr;
  try {
    m = new slimgpu_matrix();
    pick_device(m, opt);
    m->nrows = nrows;
    m->nnz = rowptr[nrows];
    m->binary = rowval == nullptr;
    m->ncols = max_index_plus_one(m->nnz, rowind);  // setup.c:117
    if (m->ncols <= 0) m->ncols = 1;
    m->owns_csr = true;
    m->d_rowptr = dev_alloc<int64_t>((size_t)nrows + 1);
    m->d_rowind = dev_alloc<int32_t>((size_t)m->nnz);
    m->d_rowval = m->binary ? nullptr : dev_alloc<float>((size_t)m->nnz);
    static_assert(sizeof(ssize_t) == sizeof(int64_t), "LP64 expected");
    HIP_TRY(hipMemcpyAsync(m->d_rowptr, rowptr, sizeof(int64_t) * ((size_t)nrows + 1),
                           hipMemcpyHostToDevice, m->stream));
    if (m->nnz > 0) {
      HIP_TRY(hipMemcpyAsync(m->d_rowind, rowind, sizeof(int32_t) * (size_t)m->nnz,
                             hipMemcpyHostToDevice, m->stream));
      if (!m->binary)
        HIP_TRY(hipMemcpyAsync(m->d_rowval, rowval, sizeof(float) * (size_t)m->nnz,
                               hipMemcpyHostToDevice, m->stream));
    }
    build_column_view(m);
    m->setup_ms = now_ms() - t0;
    if (status) *status = SLIM_OK;
    return m;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixFromHost");
    if (status) *status = status_of(e);
    destroy(m);
    return nullptr;
  } catch (const InputError& e) {
    set_error(std::string("SLIMGPU_MatrixFromHost: ") + e.msg);
    if (status) *status = SLIM_ERROR_INPUT;
    destroy(m);
    return nullptr;
  } catch (const std::bad_alloc&) {
    set_error("SLIMGPU_MatrixFromHost: out of host memory");
    if (status) *status = SLIM_ERROR_MEMORY;
    destroy(m);
    return nullptr;
  }
}

slimgpu_matrix_t* matrix_from_device(int32_t nrows, int32_t ncols, const int64_t* d_rowptr,
                                     const int32_t* d_rowind, const float* d_rowval,
                                     const LearnOptions& opt, int32_t* status) {
  if (nrows < 0 || !d_rowptr) {
    set_error("SLIMGPU_MatrixFromDevice: bad CSR arguments");
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  auto* m = new slimgpu_matrix();
  const double t0 = now_ms();
  try {
    pick_device(m, opt);
    m->nrows = nrows;
    m->owns_csr = false;
    m->d_rowptr = const_cast<int64_t*>(d_rowptr);
    m->d_rowind = const_cast<int32_t*>(d_rowind);
    m->d_rowval = const_cast<float*>(d_rowval);
    m->binary = d_rowval == nullptr;
    HIP_TRY(hipMemcpy(&m->nnz, d_rowptr + nrows, sizeof(int64_t), hipMemcpyDeviceToHost));
    if (ncols <= 0) {
      int32_t* d_max = dev_alloc<int32_t>(1);
      int32_t init = -1;
      HIP_TRY(hipMemcpy(d_max, &init, sizeof(int32_t), hipMemcpyHostToDevice));
      if (m->nnz > 0) {
        hipLaunchKernelGGL(k_max_index, dim3(grid_for(m->nnz, 256, m->num_cus * 8)), dim3(256), 0,
                           m->stream, m->d_rowind, m->nnz, d_max);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(m->stream));
      }
      HIP_TRY(hipMemcpy(&init, d_max, sizeof(int32_t), hipMemcpyDeviceToHost));
      HIP_TRY(hipFree(d_max));
      ncols = init + 1;
    }
    m->ncols = ncols > 0 ? ncols : 1;
    build_column_view(m);
    m->setup_ms = now_ms() - t0;
    if (status) *status = SLIM_OK;
    return m;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixFromDevice");
    if (status) *status = status_of(e);
    destroy(m);
    return nullptr;
  } catch (const InputError& e) {
    set_error(std::string("SLIMGPU_MatrixFromDevice: ") + e.msg);
    if (status) *status = SLIM_ERROR_INPUT;
    destroy(m);
    return nullptr;
  } catch (const std::bad_alloc&) {
    set_error("SLIMGPU_MatrixFromDevice: out of host memory");
    if (status) *status = SLIM_ERROR_MEMORY;
    destroy(m);
    return nullptr;
  }
}

void matrix_free(slimgpu_matrix_t* m) { destroy(m); }

// A copy of a staged matrix on another device: the FINISHED views (CSR, CSC, column scalars) go
// device to device (hipMemcpyPeerAsync: over xGMI where the devices are peers), so the target
// neither sorts nor needs the 24 bytes per nnz of sort temporaries.  From one root the N - 1
// copies of a node run on N - 1 different links at once.
slimgpu_matrix_t* matrix_clone_to_device(const slimgpu_matrix_t* src, int32_t device,
                                         int32_t* status) {
  auto* m = new slimgpu_matrix();
  const double t0 = now_ms();
  try {
    (void)hipGetLastError();
    LearnOptions o;
    o.device = device;
    pick_device(m, o);
    m->nrows = src->nrows;
    m->ncols = src->ncols;
    m->nnz = src->nnz;
    m->binary = src->binary;
    m->exact_gram = src->exact_gram;
    m->nonpositive = src->nonpositive;
    m->owns_csr = true;
    m->h_cost = src->h_cost;
    const size_t nz = (size_t)std::max<int64_t>(m->nnz, 1);
    m->d_rowptr = dev_alloc<int64_t>((size_t)m->nrows + 1);
    m->d_rowind = dev_alloc<int32_t>(nz);
    m->d_rowval = m->binary ? nullptr : dev_alloc<float>(nz);
    m->d_colptr = dev_alloc<int64_t>((size_t)m->ncols + 1);
    m->d_colind = dev_alloc<int32_t>(nz + 64);
    m->d_colval = m->binary ? nullptr : dev_alloc<float>(nz);
    m->d_cnorm = dev_alloc<float>((size_t)m->ncols);
    m->d_csq = dev_alloc<float>((size_t)m->ncols);
    // device to device over xGMI when the two devices can reach each other (asked, not
    // assumed: a partitioned node or an IOMMU setting can say no), else through a pinned host
    // buffer, 256 MB at a time -- slower, never wrong.  SLIM_GPU_PEER=0 forces the host route.
    int can_peer = m->device == src->device ? 1 : 0;
    if (!can_peer) {
      if (hipDeviceCanAccessPeer(&can_peer, m->device, src->device) != hipSuccess) can_peer = 0;
      (void)hipGetLastError();
    }
    if (const char* e = std::getenv("SLIM_GPU_PEER")) can_peer = can_peer && std::atoi(e) != 0;
    void* bounce = nullptr;
    const size_t bounce_bytes = size_t(256) << 20;
    struct BounceFree {
      void** p;
      ~BounceFree() {
        if (*p) (void)hipHostFree(*p);
      }
    } bounce_guard{&bounce};
    auto peer = [&](void* dst, const void* from, size_t bytes) {
      if (bytes == 0 || !dst || !from) return;
      if (can_peer) {
        HIP_TRY(hipMemcpyPeerAsync(dst, m->device, from, src->device, bytes, m->stream));
        return;
      }
      if (!bounce) HIP_TRY(hipHostMalloc(&bounce, bounce_bytes, hipHostMallocDefault));
      for (size_t off = 0; off < bytes; off += bounce_bytes) {
        const size_t n = std::min(bounce_bytes, bytes - off);
        HIP_TRY(hipSetDevice(src->device));
        HIP_TRY(hipMemcpy(bounce, static_cast<const char*>(from) + off, n, hipMemcpyDeviceToHost));
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(hipMemcpy(static_cast<char*>(dst) + off, bounce, n, hipMemcpyHostToDevice));
      }
    };
    peer(m->d_rowptr, src->d_rowptr, sizeof(int64_t) * ((size_t)m->nrows + 1));
    peer(m->d_colptr, src->d_colptr, sizeof(int64_t) * ((size_t)m->ncols + 1));
    peer(m->d_cnorm, src->d_cnorm, sizeof(float) * (size_t)m->ncols);
    peer(m->d_csq, src->d_csq, sizeof(float) * (size_t)m->ncols);
    if (m->nnz > 0) {
      peer(m->d_rowind, src->d_rowind, sizeof(int32_t) * (size_t)m->nnz);
      peer(m->d_colind, src->d_colind, sizeof(int32_t) * (size_t)m->nnz);
      peer(m->d_rowval, src->d_rowval, sizeof(float) * (size_t)m->nnz);
      peer(m->d_colval, src->d_colval, sizeof(float) * (size_t)m->nnz);
    }
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->setup_ms = now_ms() - t0;
    if (status) *status = SLIM_OK;
    return m;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixFromHost (device-to-device copy of the staged matrix)");
    if (status) *status = status_of(e);
    destroy(m);
    return nullptr;
  }
}

void matrix_add_replica(slimgpu_matrix_t* m, slimgpu_matrix_t* replica) {
  m->replicas.push_back(replica);
}
const std::vector<slimgpu_matrix_t*>& matrix_replicas(const slimgpu_matrix_t* m) {
  return m->replicas;
}
void matrix_adopt_csr(slimgpu_matrix_t* m) { m->owns_csr = true; }
int32_t matrix_device(const slimgpu_matrix_t* m) { return m ? m->device : -1; }
void matrix_set_setup_ms(slimgpu_matrix_t* m, double ms) { m->setup_ms = ms; }

int32_t matrix_info(const slimgpu_matrix_t* m, int32_t* nrows, int32_t* ncols, int64_t* nnz) {
  if (!m) return SLIM_ERROR_INPUT;
  if (nrows) *nrows = m->nrows;
  if (ncols) *ncols = m->ncols;
  if (nnz) *nnz = m->nnz;
  return SLIM_OK;
}

double matrix_setup_ms(const slimgpu_matrix_t* m) { return m ? m->setup_ms : 0.0; }

void matrix_expect_solves(slimgpu_matrix_t* m, int32_t n) {
  if (!m) return;
  m->expect_solves = n;
  for (slimgpu_matrix* r : m->replicas) r->expect_solves = n;
}

int32_t matrix_column_cost(const slimgpu_matrix_t* m, int64_t* cost) {
  if (!m || !cost) return SLIM_ERROR_INPUT;
  std::memcpy(cost, m->h_cost.data(), sizeof(int64_t) * m->h_cost.size());
  return SLIM_OK;
}

int32_t matrix_get_column_view(const slimgpu_matrix_t* m, int64_t* colptr, int32_t* colind,
                               float* colval, float* cnorms) {
  if (!m) return SLIM_ERROR_INPUT;
  try {
    HIP_TRY(hipSetDevice(m->device));
    if (colptr)
      HIP_TRY(hipMemcpy(colptr, m->d_colptr, sizeof(int64_t) * ((size_t)m->ncols + 1),
                        hipMemcpyDeviceToHost));
    if (colind && m->nnz)
      HIP_TRY(hipMemcpy(colind, m->d_colind, sizeof(int32_t) * (size_t)m->nnz,
                        hipMemcpyDeviceToHost));
    if (colval && m->nnz && m->d_colval)
      HIP_TRY(hipMemcpy(colval, m->d_colval, sizeof(float) * (size_t)m->nnz,
                        hipMemcpyDeviceToHost));
    if (cnorms)
      HIP_TRY(hipMemcpy(cnorms, m->d_cnorm, sizeof(float) * (size_t)m->ncols,
                        hipMemcpyDeviceToHost));
    return SLIM_OK;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixGetColumnView");
    return status_of(e);
  }
}

// ---- the solve -----------------------------------------------------------------

namespace {

KernelFn pick_kernel(bool lds, bool has_val) {
  if (lds) return has_val ? cd_wave_kernel<true, true> : cd_wave_kernel<true, false>;
  return has_val ? cd_wave_kernel<false, true> : cd_wave_kernel<false, false>;
}

int round_up(int v, int q) { return (v + q - 1) / q * q; }

// test-only behaviour switches: honoured only when SLIM_GPU_TEST_HOOKS=1 is set as well
bool test_hook(const char* name) {
  const char* master = std::getenv("SLIM_GPU_TEST_HOOKS");
  return master && std::atoi(master) == 1 && std::getenv(name) != nullptr;
}

constexpr int kBitmapBytes = 64 * 1024;  // dynamic LDS of a tile workgroup (user bitmap): two
                                         // 8-wave workgroups per CU still fit (2 x (64 + 10) KB)

}  // namespace

namespace {

// G (floats, m->ws_G) -> byte planes in popularity order (gram_pack.hpp).  Returns false, and
// leaves the handle on the float kernels, when G holds anything but integers in [0, 2^24)
// (fractional or negative ratings) or the planes do not fit the free memory.
bool pack_gram(slimgpu_matrix* m) {
  if (std::getenv("SLIM_GPU_NO_GRAMR")) return false;  // (not an attempt: a later call may pack)
  m->Gp_tried = true;
  const int32_t ncols = m->ncols;
  hipStream_t st = m->stream;
  const double t0 = now_ms();
  // popularity order: ratings per item descending, ties by id
  std::vector<int64_t> cp((size_t)ncols + 1);
  HIP_TRY(hipMemcpy(cp.data(), m->d_colptr, sizeof(int64_t) * cp.size(), hipMemcpyDeviceToHost));
  const int32_t nchunks = (ncols + 15) / 16;
  // Nothing reads the planes beyond the largest on-chip instantiation (gramr_kernel(): 13 groups of
  // 8192 ranks = 106 496 items), and their layout ends not far behind it: one base byte per group in
  // a 16-byte record per thread (16 groups), 17 bits of rank and 4 bits of hi_k / hi2_k in the row
  // record (131 072 ranks, 15 groups).  Larger matrices stay on the float kernels.
  static_assert(kGramrMaxGroups <= 15 && kGramrMaxGroups * kPackGroup <= (1 << 17),
                "row record: 17 bits of rank, 4 bits per plane length; base bytes: 16 groups per thread");
  if ((nchunks + kGramrNT - 1) / kGramrNT > kGramrMaxGroups) return false;
  std::vector<int32_t> item_of((size_t)nchunks * 16, -1), rank_of((size_t)ncols);
  std::iota(item_of.begin(), item_of.begin() + ncols, 0);
  std::stable_sort(item_of.begin(), item_of.begin() + ncols, [&](int32_t a, int32_t b) {
    return cp[(size_t)a + 1] - cp[(size_t)a] > cp[(size_t)b + 1] - cp[(size_t)b];
  });
  for (int32_t r = 0; r < ncols; ++r) rank_of[(size_t)item_of[(size_t)r]] = r;
  int32_t* d_item_of = ws_get<int32_t>(m->ws_itemof, item_of.size());
  int32_t* d_rank_of = ws_get<int32_t>(m->ws_rankof, rank_of.size());
  HIP_TRY(hipMemcpyAsync(d_item_of, item_of.data(), sizeof(int32_t) * item_of.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_rank_of, rank_of.data(), sizeof(int32_t) * rank_of.size(), hipMemcpyHostToDevice, st));
  int32_t* d_hik = ws_get<int32_t>(m->ws_hik, (size_t)ncols + 1);   // [ncols] + the flag word
  int32_t* d_hi2k = ws_get<int32_t>(m->ws_hi2k, (size_t)ncols);
  int32_t* d_flag = d_hik + ncols;
  HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int32_t), st));
  const float* dG = static_cast<const float*>(m->ws_G.p);
  // (SLIM_GPU_PACK_BASE=0: no per-chunk base bytes -- the first form of the planes, A/B runs)
  int use_base = 1;
  if (const char* e = std::getenv("SLIM_GPU_PACK_BASE")) use_base = std::atoi(e) != 0;
  hipLaunchKernelGGL(gram_pack_scan_fn(), dim3(ncols), dim3(256), 0, st, dG, m->G_ld, ncols, d_item_of, nchunks,
                     use_base, d_hik, d_hi2k, d_flag);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> hk((size_t)ncols + 1), h2k((size_t)ncols);
  HIP_TRY(hipMemcpyAsync(hk.data(), d_hik, sizeof(int32_t) * hk.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h2k.data(), d_hi2k, sizeof(int32_t) * h2k.size(), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (hk[(size_t)ncols] != 0) return false;  // not integers in [0, 2^24): stays on the float kernels
  // one pool: a row's hi groups, then its hi2 groups (the solver derives the second offset)
  std::vector<int64_t> off1((size_t)ncols), off2((size_t)ncols);
  int64_t n1 = 0, n2 = 0, npool = 0;
  for (int32_t i = 0; i < ncols; ++i) {
    off1[(size_t)i] = npool;
    npool += (int64_t)hk[(size_t)i] * kPackGroup;
    off2[(size_t)i] = npool;
    npool += (int64_t)h2k[(size_t)i] * kPackGroup;
    n1 += (int64_t)hk[(size_t)i] * kPackGroup;
    n2 += (int64_t)h2k[(size_t)i] * kPackGroup;
  }
  const int64_t ldb = (int64_t)nchunks * 16;
  // (a group of slack behind each pool: a lane outside a plane's prefix reads byte 0 of the plane)
  const size_t need = (size_t)ncols * ((size_t)ldb + kPackGroup) + (size_t)n1 + (size_t)n2 + 2 * (size_t)kPackGroup;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if (need + (size_t(4) << 30) > free_b + m->ws_Glo.bytes + m->ws_Ghi.bytes + m->ws_Gbase.bytes) return false;
  uint8_t* d_lo = ws_get<uint8_t>(m->ws_Glo, (size_t)ncols * (size_t)ldb);
  uint8_t* d_hi = ws_get<uint8_t>(m->ws_Ghi, (size_t)npool + kPackGroup);
  uint8_t* d_hi2 = d_hi;
  uint8_t* d_base = ws_get<uint8_t>(m->ws_Gbase, (size_t)ncols * kPackGroup);
  HIP_TRY(hipMemsetAsync(d_base, 0, (size_t)ncols * kPackGroup, st));
  float* d_diag = ws_get<float>(m->ws_Gdiag, (size_t)ncols);
  HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(float) * (size_t)ncols, st));
  int64_t* d_off1 = ws_get<int64_t>(m->ws_hioff, (size_t)ncols);
  int64_t* d_off2 = ws_get<int64_t>(m->ws_hi2off, (size_t)ncols);
  HIP_TRY(hipMemcpyAsync(d_off1, off1.data(), sizeof(int64_t) * off1.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_off2, off2.data(), sizeof(int64_t) * off2.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_hi + npool, 0, kPackGroup, st));
  hipLaunchKernelGGL(gram_pack_write_fn(), dim3(ncols), dim3(256), 0, st, dG, m->G_ld, ncols, d_item_of, nchunks,
                     d_lo, ldb, d_hi, d_off1, d_hik, d_hi2, d_off2, d_hi2k, d_base, d_diag);
  HIP_TRY(hipGetLastError());
  uint4* d_meta = ws_get<uint4>(m->ws_Gmeta, (size_t)ncols);
  hipLaunchKernelGGL(gram_pack_meta_fn(), dim3((ncols + 255) / 256), dim3(256), 0, st, ncols, d_rank_of, d_hik,
                     d_hi2k, d_off1, d_diag, m->d_colptr, m->d_csq, m->d_cnorm, d_meta, d_flag);
  HIP_TRY(hipGetLastError());
  int32_t meta_flag = 0;
  HIP_TRY(hipMemcpyAsync(&meta_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));  // (off1 / off2 are locals)
  if (meta_flag & 2) return false;    // |a_i|^2 of the planes is not the column view's: float kernels
  m->Gp_ldb = ldb;
  m->Gp_nchunks = nchunks;
  m->Gp_bytes_per_row = (double)ldb + 16.0 * std::min(nchunks, kGramrNT) + (double)(n1 + n2) / std::max(1, ncols);
  m->Gp_ready = true;
  if (const char* te = std::getenv("SLIM_GPU_TRACE"); te && std::atoi(te) >= 1)
    std::fprintf(stderr, "[trace] G packed: lo %.2f GB + base %.2f GB + hi %.2f GB + hi2 %.3f GB = %.3f bytes per entry, %.1f ms\n",
                 (double)ncols * ldb * 1e-9, (double)ncols * kPackGroup * 1e-9, n1 * 1e-9, n2 * 1e-9,
                 m->Gp_bytes_per_row / std::max(1, ncols), now_ms() - t0);
  return true;
}

// Once the byte planes stand and a kernel can consume them, nothing reads the floats of G: the
// on-chip kernel takes aTy of a problem (x's active set, the loss term) from the planes of row iC
// (cd_gramr.hpp, SLIM_GRAMR_FROM_PLANES).  Large ones are freed -- C4: 40 GB per handle --; a later
// solve that needs floats (SLIM_GPU_NO_GRAMR=1, SLIMGPU_MatrixGramView) forms them again.
// SLIM_GPU_KEEP_G=1 keeps them; SLIM_GPU_DROP_G_MIN_GB sets the size from which they go (default 8).
void drop_float_gram(slimgpu_matrix* m) {
  if (!m->Gp_ready || !m->ws_G.p || std::getenv("SLIM_GPU_KEEP_G") || std::getenv("SLIM_GPU_NO_GRAMR")) return;
  if ((m->Gp_nchunks + kGramrNT - 1) / kGramrNT > kGramrMaxGroups) return;
  double min_gb = 8.0;
  if (const char* e = std::getenv("SLIM_GPU_DROP_G_MIN_GB")) min_gb = std::atof(e);
  if ((double)m->ws_G.bytes < min_gb * 1073741824.0) return;
  (void)hipStreamSynchronize(m->stream);
  (void)hipFree(m->ws_G.p);
  m->ws_G.p = nullptr;
  m->ws_G.bytes = 0;
  m->Gf_dropped = true;
}

}  // namespace

slim_csr_t* learn_cd(slimgpu_matrix_t* m, const LearnOptions& opt, const slim_csr_t* imodel,
                     int32_t* status, const int32_t* columns, int32_t ncolumns, bool row_view,
                     ResidentIO* rio) {
  const double t_begin = now_ms();
  slimgpu_stats_t st;
  std::memset(&st, 0, sizeof(st));
  auto fail = [&](int32_t code) -> slim_csr_t* {
    if (status) *status = code;
    return nullptr;
  };
  if (!m) {
    set_error("SLIMGPU_Learn: null matrix");
    return fail(SLIM_ERROR_INPUT);
  }
  const int32_t ncols = m->ncols;
  int32_t cb = std::max(0, opt.col_begin);
  int32_t ce = opt.col_end < 0 ? ncols : std::min(opt.col_end, ncols);
  if (cb > ce) cb = ce;
  int32_t nwork = ce - cb;
  if (columns) {  // an explicit set of item columns instead of a range
    std::vector<char> seen((size_t)ncols, 0);
    for (int32_t k = 0; k < ncolumns; ++k) {
      if (columns[k] < 0 || columns[k] >= ncols || seen[(size_t)columns[k]]) {
        set_error("SLIMGPU_LearnColumns: column ids must be distinct and inside [0, ncols)");
        return fail(SLIM_ERROR_INPUT);
      }
      seen[(size_t)columns[k]] = 1;
    }
    nwork = ncolumns;
  }

  if (opt.shard_index < 0 || opt.shard_index >= opt.shard_count) {
    set_error("SLIMGPU_Learn: shard index outside [0, shard count)");
    return fail(SLIM_ERROR_INPUT);
  }

  try {
    (void)hipGetLastError();
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t stream = m->stream;
    // One solve at a time per device and process: the tile kernel sizes its grid to the whole
    // chip and its clusters need every member workgroup resident, which two concurrent
    // launches (two host threads calling SLIM_Learn on one GPU) would not guarantee.
    // (recursive: the first item-space solve builds G through a nested call of this function)
    static std::recursive_mutex device_lock[64];
    std::lock_guard<std::recursive_mutex> solve_guard(device_lock[m->device & 63]);

    // work list: most expensive columns first (longest-processing-time order)
    std::vector<int32_t> order((size_t)nwork);
    if (columns)
      std::copy(columns, columns + nwork, order.begin());
    else
      std::iota(order.begin(), order.end(), cb);
    std::stable_sort(order.begin(), order.end(),
                     [&](int32_t a, int32_t b) { return m->h_cost[a] > m->h_cost[b]; });
    if (opt.shard_count > 1) {  // granules of 32 work-list entries, dealt round-robin
      std::vector<int32_t> mine;
      for (int32_t t = 0; t < nwork; ++t)
        if ((t / 32) % opt.shard_count == opt.shard_index) mine.push_back(order[(size_t)t]);
      order.swap(mine);
      nwork = (int32_t)order.size();
    }
    // (G = R^T R by row blocks: the block's items come first in the list and only their tiles run
    // -- a tile forms the sums with every column at or behind its own position, so the first tiles
    // of a list form whole rows)
    int32_t G_block = 0;
    if (opt.build_G && opt.G_rows_end >= 0) {
      auto in_block = [&](int32_t c) { return c >= opt.G_rows_begin && c < opt.G_rows_end; };
      std::stable_partition(order.begin(), order.end(), in_block);
      G_block = (int32_t)std::count_if(order.begin(), order.end(), in_block);
    }
    const std::vector<int32_t> requested = order;

    // kernel flavour and geometry
    const int nrows_pad = round_up(std::max(m->nrows, 1), 64);
    const int ncols_pad = round_up(ncols, 64);
    const size_t vec_floats = (size_t)nrows_pad + 2 * (size_t)ncols_pad;
    const size_t lds_need = vec_floats * sizeof(float);
    int kernel = opt.kernel;
    // Item-space CD on G = R^T R (cd_gram.hpp): when asked for, or -- left to the engine -- when
    // this matrix is being solved repeatedly (G is there already; the caller announced a grid,
    // SLIMGPU_MatrixExpectSolves; the very work list of the previous call comes again), the
    // matrix is beyond the one-wavefront-per-item kernel, g fits the LDS of a CU and G the HBM.
    int gram_nw = 0, gram_v = 0;
    const bool gram_fits =
        gram_geometry(ncols_pad, &gram_nw, &gram_v) && opt.nnbrs == 0 && !opt.build_G && ncols > 0;
    const int64_t G_ld = round_up(ncols_pad, 64);
    const size_t G_bytes = sizeof(float) * (size_t)ncols * (size_t)G_ld;
    bool use_gram = false;
    if (kernel == SLIMGPU_KERNEL_GRAM) {
      if (!gram_fits) {
        set_error("SLIMGPU_Learn: the item-space kernel has no FSLIM form");
        return fail(SLIM_ERROR_INPUT);
      }
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      if (!m->G_ready && G_bytes + (size_t(4) << 30) > free_b + m->ws_gram.bytes) {
        set_error("SLIMGPU_Learn: G = R^T R (4 ncols^2 bytes) does not fit the free HBM");
        return fail(SLIM_ERROR_MEMORY);
      }
      use_gram = true;
    } else if (kernel == SLIMGPU_KERNEL_AUTO && gram_fits && lds_need > 64 * 1024 &&
               !std::getenv("SLIM_GPU_NO_GRAMCD")) {
      // The engine's own choice between the residual (tile) kernel and item space, by their byte
      // models per problem and sweep: the tile kernel moves ~5.4 bytes per nnz of
      // R (ids + one residual line per nnz shared by 32 problems, write-backs), item space one
      // row of G per update -- f ncols rows of 4 ncols bytes with f ~ 3 % of the coordinates
      // carrying a coefficient (C4 2.6 %, C4 at 0.1 % 2.7 %, ml100k 2.4 %).  rho = item / tile
      // = (ncols^2 / nnz) / 45; measured 0.22 on C4 (1.3 against 5.8 ms per column), 2.6 on C4 at
      // 0.1 % density.  G itself costs what ~ncols / 32 columns cost the tile kernel (one screen
      // pass over every column; measured ncols / 64 on C4, ncols / 36 on C5), so a FIRST solve
      // takes item space when the columns it solves -- times the number of solves the caller
      // announced (SLIMGPU_MatrixExpectSolves: a model-selection grid) -- save more than that;
      // with G already there the per-column figure decides alone.  A shard of a multi-GPU solve
      // applies the rule to its own columns (every replica builds its own G).  Deterministic in
      // the call's arguments and the handle's state (G built or not): the same call on a fresh
      // handle always takes the same kernel.
      const double rho = (double)ncols * (double)ncols / std::max(1.0, (double)m->nnz) / 45.0;
      const double solves = (double)std::max(1, m->expect_solves);
      // (explicit cluster / heavy-phase options describe a residual-kernel launch: honoured)
      const bool tile_geometry_asked = opt.cluster != 0 || opt.heavy_tiles >= 0 || opt.heavy_cluster != 0;
      bool item_space = rho < 1.0 && !tile_geometry_asked &&
                        (m->G_ready || (double)nwork * solves * (1.0 - rho) > (double)ncols / 32.0);
      if (const char* e = std::getenv("SLIM_GPU_GRAMCD"); e && std::strcmp(e, "never-first") == 0)
        item_space = item_space && (m->G_ready || m->expect_solves >= 2);  // (round-4 policy, A/B runs)
      if (item_space) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        use_gram = m->G_ready || G_bytes + (size_t(8) << 30) <= free_b + m->ws_gram.bytes;
      }
    }
    if (use_gram) kernel = SLIMGPU_KERNEL_GRAM;
    if (use_gram && m->G_ready && m->Gf_dropped && std::getenv("SLIM_GPU_NO_GRAMR")) {
      m->G_ready = false;  // (the float kernels were asked for: form the floats again; the planes stay)
      m->Gf_dropped = false;
    }
    if (use_gram && !m->G_ready) {
      // G by the tile kernel's screen pass over every column (S.gram_mode 3), once per handle
      const double tb = now_ms();
      drop_screen_cache(m);  // (G holds the same sums for every column)
      float* dG = ws_get<float>(m->ws_G, (size_t)ncols * (size_t)G_ld);
      const double t_alloc = now_ms();
      HIP_TRY(hipMemsetAsync(dG, 0, G_bytes, stream));
      m->G_ld = G_ld;
      LearnOptions bo = opt;
      bo.kernel = SLIMGPU_KERNEL_TILE;
      bo.build_G = true;
      bo.col_begin = 0;
      bo.col_end = -1;
      bo.shard_count = 1;
      bo.shard_index = 0;
      bo.nnbrs = 0;
      bo.cluster = 0;
      bo.heavy_tiles = 0;
      bo.dbglvl = 0;
      int32_t bst = SLIM_OK;
      slim_csr_t* none = learn_cd(m, bo, nullptr, &bst, nullptr, 0, false);
      if (!none) return fail(bst);
      csr_free(none);
      const double t_sums = now_ms();
      const double sums_kernel_ms = last_stats().kernel_ms;
      m->G_ready = true;
      m->Gf_dropped = false;
      if (!m->Gp_ready) {  // (planes of an earlier build of the same G are still right)
        m->Gp_tried = false;
        if (!pack_gram(m)) m->Gp_ready = false;
      }
      drop_float_gram(m);
      m->G_build_ms = now_ms() - tb;
      m->G_alloc_ms = t_alloc - tb;
      m->G_sums_ms = t_sums - t_alloc;
      m->G_sums_kernel_ms = sums_kernel_ms;
      m->G_pack_ms = now_ms() - t_sums;
      if (const char* te = std::getenv("SLIM_GPU_TRACE"); te && std::atoi(te) >= 1)
        std::fprintf(stderr, "[trace] G = R^T R (%d x %d, %.2f GB) built in %.1f ms: allocation %.1f, sums %.1f "
                     "(kernel %.1f), byte planes %.1f\n", ncols, ncols, G_bytes * 1e-9, m->G_build_ms,
                     t_alloc - tb, t_sums - t_alloc, sums_kernel_ms, now_ms() - t_sums);
    }
    if (kernel == SLIMGPU_KERNEL_AUTO)
      kernel = lds_need <= 64 * 1024 ? SLIMGPU_KERNEL_WAVE_LDS : SLIMGPU_KERNEL_TILE;
    if (kernel == SLIMGPU_KERNEL_WAVE_LDS && lds_need > 160 * 1024) {
      set_error("SLIMGPU_Learn: work vectors do not fit the 160 KiB LDS of a CU");
      return fail(SLIM_ERROR_INPUT);
    }
    if (kernel < SLIMGPU_KERNEL_WAVE_LDS || kernel > SLIMGPU_KERNEL_GRAM) {
      set_error("SLIMGPU_Learn: unknown kernel selection");
      return fail(SLIM_ERROR_INPUT);
    }
    const bool use_lds = kernel == SLIMGPU_KERNEL_WAVE_LDS;
    if (use_gram && gram_v == 0)  // g in HBM: 8 or 16 wavefronts per workgroup
      if (const char* e = std::getenv("SLIM_GPU_GRAM_NW")) gram_nw = std::atoi(e) == 8 ? 8 : 16;
    // item space with g on chip and G as byte planes (cd_gramr.hpp) whenever G could be packed and
    // the items fit the largest instantiation (106 496); SLIM_GPU_NO_GRAMR=1: the float kernels
    GramrFn fn_r = nullptr;
    int gramr_kr = 0, gramr_kl = 0;
    if (use_gram && m->G_ready && !m->Gp_tried && !m->Gf_dropped) {
      (void)pack_gram(m);
      drop_float_gram(m);
    }
    size_t gramr_lds = 0;
    // rows through the LDS ring (global_load_lds) where a row is many groups long -- measured
    // (profiles/r05/gramr_dma_ab.txt): C4, 13 groups, kernel 6.62 -> 5.37 s; C5, 3 groups, where
    // one round of register loads already holds the whole row, 1.32 -> 1.41 s.
    // SLIM_GPU_GRAMR_DMA=0 / 1 forces either.
    bool gramr_dma = (m->Gp_nchunks + kGramrNT - 1) / kGramrNT > 6;
    if (const char* e = std::getenv("SLIM_GPU_GRAMR_DMA")) gramr_dma = std::atoi(e) != 0;
    if (use_gram && m->Gp_ready && !std::getenv("SLIM_GPU_NO_GRAMR"))
      fn_r = gramr_kernel(m->Gp_nchunks, gramr_dma, &gramr_kr, &gramr_kl, &gramr_lds);
    const bool use_gramr = fn_r != nullptr;
    if (use_gram && !use_gramr && m->Gf_dropped) {
      set_error("SLIMGPU_Learn: the floats of G were dropped and the byte-plane kernel cannot run this solve");
      return fail(SLIM_ERROR);
    }
    if (use_gramr) {
      gram_nw = kGramrNT / 64;
      gram_v = 1;  // (x only in the slab: g is on chip)
    }
    const size_t gram_lds = use_gramr ? gramr_lds : (gram_v > 0 ? sizeof(float) * (size_t)ncols_pad : 0);
    // tile width: 32 item columns per workgroup (128-byte residual lines) unless the row
    // offsets would overflow the kernel's 32-bit byte offsets
    int tileP = kernel == SLIMGPU_KERNEL_TILE16 ? 16 : 32;
    if (kernel == SLIMGPU_KERNEL_TILE && ((int64_t)nrows_pad + 64) * 128 >= (int64_t(1) << 32)) tileP = 16;
    bool use_tile = kernel == SLIMGPU_KERNEL_TILE || kernel == SLIMGPU_KERNEL_TILE16;
    if (use_tile && ((int64_t)nrows_pad + 64) * 4 * tileP >= (int64_t(1) << 32)) {
      use_tile = false;  // > 67M users: fall back to one wavefront per item
      kernel = SLIMGPU_KERNEL_WAVE_HBM;
    }
    if (use_tile && tileP == 16 && opt.nnbrs > 0) {  // FSLIM exists for 32-wide tiles only
      use_tile = false;
      kernel = SLIMGPU_KERNEL_WAVE_HBM;
    }
    if (use_tile) kernel = tileP == 32 ? SLIMGPU_KERNEL_TILE : SLIMGPU_KERNEL_TILE16;
    if (opt.build_G && !use_tile) {
      set_error("SLIMGPU_Learn: G = R^T R is built by the tile kernel, which this matrix cannot use");
      return fail(SLIM_ERROR_INPUT);
    }
    const char* trace_env = std::getenv("SLIM_GPU_TRACE");
    const int trace_level = trace_env ? std::atoi(trace_env) : 0;
    KernelFn fn = use_gram ? gram_kernel(gram_nw, gram_v) : pick_kernel(use_lds, !m->binary);
    // tile workgroup geometry: 16 wavefronts (1 workgroup per CU) or 8 (2 per CU, phases of
    // the two overlap).  SLIM_GPU_TILE_NW overrides the default.
    // Measured on C4 (profiles/r01): with few tiles per cluster the launch is bound by the
    // slowest tile and one big workgroup per CU finishes it sooner (134 vs 109-116 col/s at
    // 256 tiles); with many tiles throughput matters and two small workgroups win (150 vs
    // ~141 col/s at 512 tiles).
    int tileNW = (nwork + tileP - 1) / tileP >= 2 * m->num_cus ? 8 : 16;
    // G = R^T R of a binary matrix: clusters of 32 whose members keep the y of a tile's 32 items as
    // one word per user of their range in LDS (cd_tile.hpp, gbits) -- when that range fits
    int req_cluster = opt.cluster;
    size_t gram_bits_lds = 0;
    int gram_passes = 1, gram_pass = 0;  // (user passes of the G builder, below)
    if (opt.build_G && use_tile && tileP == 32 && m->binary && !std::getenv("SLIM_GPU_NO_GBITS")) {
      ensure_cluster_split(m, 5);
      const size_t need = sizeof(uint32_t) * (size_t)(round_up(m->max_range_rows[5], 64) + 64);
      // (test hook: pretend a member holds only that many users, so that small matrices take the passes)
      size_t words_cap = 148 * 1024;
      if (test_hook("SLIM_GPU_TEST_GBITS_ROWS"))
        words_cap = sizeof(uint32_t) * (size_t)(round_up(std::max(64, std::atoi(std::getenv("SLIM_GPU_TEST_GBITS_ROWS"))), 64) + 64);
      if (need <= words_cap && m->num_cus >= 32) {
        gram_bits_lds = need;
        req_cluster = 32;
        tileNW = 16;
      } else if (m->num_cus >= 32 && !std::getenv("SLIM_GPU_NO_GPASSES")) {
        // More users than 32 members hold as words (C5: 10M users, 312K per member): the same
        // kernel in USER PASSES -- np launches over the whole work list, launch s with member k on
        // range 32 s + k of 32 np ranges of equal nnz, its sums added to G (exact: integer counts).
        // The id stream is the same 4 bytes per nnz and tile either way (a member reads its
        // ranges' slices of every column); what passes add is a bitmap build per tile and pass.
        const int64_t rows_cap = (int64_t)(words_cap / sizeof(uint32_t)) - 128;
        int np = (int)((m->max_range_rows[5] + rows_cap - 1) / rows_cap);
        for (; np <= 64; ++np) {
          ensure_gram_split(m, np);
          const size_t need_p = sizeof(uint32_t) * (size_t)(round_up(m->gsplit_max_rows, 64) + 64);
          if (need_p <= words_cap) {
            gram_bits_lds = need_p;
            req_cluster = 32;
            tileNW = 16;
            gram_passes = np;
            if (const char* te = std::getenv("SLIM_GPU_TRACE"); te && std::atoi(te) >= 1)
              std::fprintf(stderr, "[trace] G builder: %d user passes (32 members x %d users at most, %zu KB of words)\n",
                           np, m->gsplit_max_rows, need_p >> 10);
            break;
          }
        }
      }
    }
    // (four 4-wavefront workgroups per CU were measured too: no gain, even on columns of ~900 nnz)
    if (const char* e = std::getenv("SLIM_GPU_TILE_NW")) tileNW = std::atoi(e) == 16 ? 16 : 8;
    // warm start (estimate.c:453-464) on the tile path: how the previous coefficients are folded
    // into the residual -- "row" (default for 32-wide tiles: one pass over the member's rows,
    // x lines from one copy per cluster) or "col" (one pass per column of the union list)
    const slimgpu_model* warm_dev = rio ? rio->warm : nullptr;  // previous model already in HBM
    const bool resident = rio && rio->out;                      // the learned model stays in HBM
    if (warm_dev && warm_dev->device != m->device) {
      set_error("SLIMGPU_LearnResident: the warm-start model lives on another device");
      return fail(SLIM_ERROR_INPUT);
    }
    const bool has_imodel = warm_dev ? warm_dev->n > 0 : (imodel && imodel->colptr && imodel->ncols > 0);
    bool row_fold = true;
    if (const char* e = std::getenv("SLIM_GPU_FOLD")) row_fold = std::strcmp(e, "col") != 0;
    if (use_tile) {
      const bool prof = trace_level >= 2;
      const bool val = !m->binary;
      if (tileP == 32 && opt.nnbrs > 0)
        fn = tile_kernel_p32_fslim(val, tileNW == 16);
      else if (tileP == 32 && !prof && !has_imodel)
        fn = tile_kernel_p32_cold(val, tileNW == 16);
      else if (tileP == 32 && !prof && row_fold)
        fn = tile_kernel_p32_rowfold(val, tileNW == 16);
      else if (tileP == 32)
        fn = tileNW == 16 ? tile_kernel_p32_nw16(val, prof) : tile_kernel_p32_nw8(val, prof);
      else
        fn = tileNW == 16 ? tile_kernel_p16_nw16(val, prof) : tile_kernel_p16_nw8(val, prof);
    }
    int waves_per_cu;
    if (use_gram) {  // one workgroup per problem, as many per CU as g (LDS) and registers allow
      int per_cu = 0;
      const void* kfn = use_gramr ? reinterpret_cast<const void*>(fn_r) : reinterpret_cast<const void*>(fn);
      HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_lds));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * gram_nw, gram_lds));
      if (per_cu < 1) {
        set_error("SLIMGPU_Learn: the item-space kernel does not fit a compute unit of this device");
        return fail(SLIM_ERROR);
      }
      waves_per_cu = per_cu;
    } else if (use_lds) {
      waves_per_cu = (int)std::min<size_t>(16, (160 * 1024) / std::max<size_t>(lds_need, 1));
      if (waves_per_cu < 1) waves_per_cu = 1;
      if (lds_need > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    } else {
      waves_per_cu = 8;
    }
    // wave kernels: one block = one wavefront; tile kernel: one block = 16 wavefronts
    int nwaves = std::max(1, std::min(nwork, m->num_cus * waves_per_cu));
    size_t tile_r = 0, tile_x = 0, tile_u = 0;
    int clusterK = 1, cluster_lg = 0, nclusters = 0;
    int clusterHi = 0, hi_lg = 0, nheavy = 0, nclusters_hi = 0, auto_heavy = 0;
    // co-resident tile workgroups: what the occupancy calculator grants this instantiation
    // (1 x 16 or 2 x 8 wavefronts per CU by design; fewer if the register or LDS footprint
    // of a build ever grows), never more than the design assumes
    int wg_slots = m->num_cus * (16 / tileNW);
    if (use_tile) {
      int per_cu = 0;
      const size_t worst_lds = std::max<size_t>(kBitmapBytes, gram_bits_lds);
      // (static + dynamic LDS beyond 64 KB needs the attribute)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)worst_lds));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn),
                                                           64 * tileNW, worst_lds));
      if (per_cu < 1) {
        set_error("SLIMGPU_Learn: the tile kernel does not fit a compute unit of this device");
        return fail(SLIM_ERROR);
      }
      wg_slots = m->num_cus * std::min(per_cu, 16 / tileNW);
    }
    // force_k1: no clusters, no heavy phase -- the geometry that needs no co-residency at all
    // (fallback after a cluster timed out waiting for a member, e.g. under a CU mask)
    bool force_k1_now = false;
    auto plan_tiles = [&](const bool force_k1) {
      force_k1_now = force_k1;
      clusterK = 1; cluster_lg = 0; nclusters = 0;
      clusterHi = 0; hi_lg = 0; nheavy = 0; nclusters_hi = 0; auto_heavy = 0;
      const int ngroups_all = (nwork + tileP - 1) / tileP;
      // cluster size: share a tile among K workgroups when there are too few tiles to keep
      // every CU busy behind the slowest one (auto), or as requested
      if (force_k1) {
        clusterK = 1;
      } else if (req_cluster == 1 || req_cluster == 2 || req_cluster == 4 || req_cluster == 8 ||
                 req_cluster == 16 || req_cluster == 32) {
        clusterK = req_cluster;
      } else {
        // the heaviest tile runs ~7x the median (popular items need more sweeps): a
        // cluster should see >= ~8 tiles so the others fill in behind it; with fewer
        // tiles per cluster, larger clusters shorten that critical path instead
        // ... but a member's slice of a column should stay long enough (>= ~512 nnz on
        // average) for the gather to amortise the per-visit exchange
        int cap = 1;
        while (cap < 16 && (m->nnz / std::max(ncols, 1)) / (2 * cap) >= 512) cap *= 2;
        // heavy tiles (queue order = cost order): a few tiles of the most popular items run
        // 5-8x the median (measured on C4: 52 / 36 / 29 s against 6.4 s).  They go to big
        // clusters first (below), which lets everything else use small, efficient clusters:
        // clusters of 2-4 reach ~0.9 of the HBM roofline, clusters of 8 pay ~25 % for the
        // per-visit exchange.  Cost is only a proxy for time, so the test is generous (a
        // light tile solved by a big cluster wastes a few CU-seconds, a heavy one solved by
        // a small cluster is the critical path of the launch).
        if (opt.heavy_tiles < 0 && tileNW == 16 && ngroups_all >= 16) {
          auto tile_cost = [&](int gI) {
            int64_t c = 0;
            for (int t = gI * tileP; t < std::min((gI + 1) * tileP, nwork); ++t)
              c += m->h_cost[order[(size_t)t]];
            return c;
          };
          const int64_t med = tile_cost(ngroups_all / 2);
          while (auto_heavy < ngroups_all / 16 && tile_cost(auto_heavy) >= 12 * med) ++auto_heavy;
        }
        const int64_t fill = auto_heavy > 0 ? 4 : 8;  // tiles wanted per workgroup slot
        while (clusterK < cap && (int64_t)ngroups_all * clusterK < fill * (int64_t)wg_slots)
          clusterK *= 2;
      }
      while (clusterK > 1 && wg_slots / clusterK < 1) clusterK /= 2;
      for (cluster_lg = 0; (1 << cluster_lg) < clusterK; ++cluster_lg) {}
      ensure_cluster_split(m, cluster_lg);
      // (+ 64 rows: the spare residual line behind a member's user range, cd_tile.hpp)
      tile_r = (size_t)(round_up(m->max_range_rows[cluster_lg], 64) + 64) * tileP;
      // (the G builder's word-per-user form keeps no residual: C5 would reserve 10 GB of lines)
      if (opt.build_G && gram_bits_lds && clusterK == 32 && !force_k1) tile_r = (size_t)64 * tileP;
      tile_x = (size_t)ncols_pad * tileP;
      tile_u = (size_t)ncols_pad;
      nclusters = std::max(1, std::min(ngroups_all, wg_slots / clusterK));
      // heavy phase: the first nheavy tiles (most expensive) go to clusters of clusterHi
      nheavy = std::min(opt.heavy_tiles < 0 ? auto_heavy : opt.heavy_tiles, ngroups_all);
      clusterHi = opt.heavy_cluster;
      if (const char* e = std::getenv("SLIM_GPU_HEAVY")) {  // "tiles,cluster" (experiments)
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2) {
          nheavy = std::min(std::max(a, 0), ngroups_all);
          clusterHi = b;
        }
      }
      if (clusterHi != 2 && clusterHi != 4 && clusterHi != 8 && clusterHi != 16 && clusterHi != 32)
        clusterHi = std::max(16, std::min(4 * clusterK, kTileKMax));
      if (clusterHi <= clusterK || nclusters * clusterK < clusterHi || force_k1) nheavy = 0;
      if (nheavy > 0) {
        for (hi_lg = 0; (1 << hi_lg) < clusterHi; ++hi_lg) {}
        ensure_cluster_split(m, hi_lg);
        nclusters_hi = nclusters * clusterK / clusterHi;
        tile_r = std::max(tile_r, (size_t)(round_up(m->max_range_rows[hi_lg], 64) + 64) * tileP);
      }
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      const size_t per_cl = ((tile_r + 2 * tile_x) * sizeof(float) + tile_u * sizeof(int32_t)) * clusterK;
      // (the screen-sum cache is given up when a workspace does not fit: ws_get evicts it)
      const size_t have = free_b + m->ws_slab.bytes + m->ws_xslab.bytes + m->ws_ulist.bytes +
                          m->ws_part.bytes + m->ws_gram.bytes;
      const size_t budget = have > (size_t(6) << 30) ? have - (size_t(6) << 30) : have / 2;
      if ((size_t)nclusters * per_cl > budget) {
        nclusters = (int)std::max<size_t>(1, budget / per_cl);
        nclusters_hi = nclusters * clusterK / std::max(clusterHi, 1);
        if (nclusters_hi < 1) nheavy = 0;
      }
      nwaves = nclusters * clusterK;  // workgroups launched
    };
    if (use_tile) plan_tiles(false);

    // device buffers
    int32_t* d_order = ws_get<int32_t>(m->ws_order, (size_t)nwork);
    int32_t* d_cnt = ws_get<int32_t>(m->ws_cnt, (size_t)ncols);
    int64_t* d_off = ws_get<int64_t>(m->ws_off, (size_t)ncols);
    int32_t* d_sti = ws_get<int32_t>(m->ws_stat_i, 3 * (size_t)ncols);
    int64_t* d_stl = ws_get<int64_t>(m->ws_stat_l, 4 * (size_t)ncols);
    float* d_stf = ws_get<float>(m->ws_stat_f, 2 * (size_t)ncols);
    // misc: [0] queue (int32) [1] overflow (int32) [2..3] cursor (u64)
    int32_t* d_misc = ws_get<int32_t>(m->ws_misc, 16);  // [4] queue of the heavy phase
    float* d_slab = nullptr;
    float* d_xslab = nullptr;
    int32_t* d_ulist = nullptr;
    unsigned long long* d_mailbox = nullptr;
    float* d_part = nullptr;
    int bm_shift = 0, bm_words = 1;
    // dynamic LDS of a tile workgroup: the user bitmap of the screen pass (FSLIM: the select
    // histograms)
    size_t tile_lds = 0;
    const size_t mailbox_stride = 2 * (size_t)kTileKMax * (size_t)tileP + 8;
    size_t mailbox_words = 0;
    auto alloc_tiles = [&]() {
      mailbox_words = (size_t)(std::max(nclusters, 1) + nclusters_hi) * mailbox_stride;
      d_slab = ws_get<float>(m->ws_slab, tile_r * (size_t)nwaves, m);
      d_xslab = ws_get<float>(m->ws_xslab, tile_x * (size_t)nwaves, m);
      d_ulist = ws_get<int32_t>(m->ws_ulist, tile_u * (size_t)nwaves, m);
      d_mailbox = ws_get<unsigned long long>(m->ws_mailbox, mailbox_words, m);
      d_part = ws_get<float>(m->ws_part, tile_x * (size_t)nwaves, m);
      // LDS user bitmap of the screen pass: one bit per 2^shift users of a member's range,
      // at most kBitmapBytes
      const int64_t range = (int64_t)(tile_r / (size_t)tileP);
      auto words_at = [&](int sh) { return ((range >> sh) + 1 + 31) / 32; };
      bm_shift = 0;
      while (words_at(bm_shift) * 4 > kBitmapBytes) ++bm_shift;
      bm_words = (int)words_at(bm_shift);
      if (opt.nnbrs > 0) bm_words = std::max(bm_words, tileP * 256);  // FSLIM's select histograms
      // (the bitmap follows the member's user range, which grows when the fallback below
      // re-plans without clusters: the launch size must follow it)
      tile_lds = sizeof(uint32_t) * (size_t)bm_words;
      if (gram_bits_lds && clusterK == 32 && !force_k1_now) {  // one word per user of a member's range
        bm_shift = 0;
        bm_words = (int)(gram_bits_lds / sizeof(uint32_t));
        tile_lds = gram_bits_lds;
      }
    };
    int32_t* d_nunion = nullptr;
    if (use_gram) {
      const size_t ngroups0 = ((size_t)nwork + 31) / 32;
      d_xslab = ws_get<float>(m->ws_xslab, (size_t)ncols_pad * (size_t)nwaves, m);
      if (gram_v == 0) d_slab = ws_get<float>(m->ws_slab, (size_t)ncols_pad * (size_t)nwaves, m);
      d_ulist = ws_get<int32_t>(m->ws_ulist, (size_t)ncols_pad * ngroups0, m);
      d_nunion = ws_get<int32_t>(m->ws_nunion, ngroups0, m);
    } else if (use_tile) {
      alloc_tiles();
    } else if (!use_lds) {
      d_slab = ws_get<float>(m->ws_slab, vec_floats * (size_t)nwaves);
    }

    // output arena: a column of W holds at most ncols - 1 entries and, on the large
    // configurations, ~2.7K (C4, both densities) to ~4K (C5); columns that do not fit are
    // solved again with a larger arena (below), so the size is only a matter of cost
    int64_t arena_cap = std::max<int64_t>(1 << 20, (int64_t)nwork * std::min<int64_t>(ncols, 8192));
    const char* env_cap = std::getenv("SLIM_GPU_ARENA");
    if (env_cap) arena_cap = std::max<int64_t>(1, std::atoll(env_cap));

    // warm start: column view of imodel
    const int64_t* d_icolptr = nullptr;
    const int32_t* d_icolind = nullptr;
    const float* d_icolval = nullptr;
    int32_t incols = 0;
    const double t_prep_done = now_ms();  // (host phases, SLIM_GPU_TRACE: prep | launches + D2H | counters | columns | row view)
    double d2h_ms = 0.0;
    if (has_imodel && warm_dev) {  // no upload: the solver reads the resident column view
      incols = warm_dev->n;
      d_icolptr = warm_dev->d_colptr;
      d_icolind = warm_dev->d_colind;
      d_icolval = warm_dev->d_colval;
    } else if (has_imodel) {
      incols = imodel->ncols;
      const int64_t innz = imodel->colptr[incols];
      int64_t* p = ws_get<int64_t>(m->ws_icolptr, (size_t)incols + 1);
      int32_t* ci = ws_get<int32_t>(m->ws_icolind, (size_t)innz);
      float* cv = ws_get<float>(m->ws_icolval, (size_t)innz);
      HIP_TRY(hipMemcpyAsync(p, imodel->colptr, sizeof(int64_t) * ((size_t)incols + 1),
                             hipMemcpyHostToDevice, stream));
      if (innz > 0) {
        HIP_TRY(hipMemcpyAsync(ci, imodel->colind, sizeof(int32_t) * (size_t)innz,
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(cv, imodel->colval, sizeof(float) * (size_t)innz,
                               hipMemcpyHostToDevice, stream));
      }
      d_icolptr = p;
      d_icolind = ci;
      d_icolval = cv;
    }

    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)ncols, stream));
    HIP_TRY(hipMemsetAsync(d_off, 0, sizeof(int64_t) * (size_t)ncols, stream));
    HIP_TRY(hipMemsetAsync(d_sti, 0, sizeof(int32_t) * 3 * (size_t)ncols, stream));
    HIP_TRY(hipMemsetAsync(d_stl, 0, sizeof(int64_t) * 4 * (size_t)ncols, stream));
    HIP_TRY(hipMemsetAsync(d_stf, 0, sizeof(float) * 2 * (size_t)ncols, stream));

    DevMatrix A;
    A.nrows = m->nrows;
    A.ncols = ncols;
    A.nnz = m->nnz;
    A.rowptr = m->d_rowptr;
    A.rowind = m->d_rowind;
    A.rowval = m->d_rowval;
    A.colptr = m->d_colptr;
    A.colind = m->d_colind;
    A.colval = m->d_colval;
    A.cnorm = m->d_cnorm;
    A.csq = m->d_csq;

    struct EventPair {  // released on every exit path
      hipEvent_t a = nullptr, b = nullptr;
      ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
      }
    } events;
    HIP_TRY(hipEventCreate(&events.a));
    HIP_TRY(hipEventCreate(&events.b));
    const hipEvent_t ev0 = events.a, ev1 = events.b;

    std::vector<int32_t> h_cnt((size_t)ncols, 0);
    std::vector<int64_t> h_off((size_t)ncols, 0);
    std::vector<int32_t> h_ind;
    std::vector<float> h_val;
    // resident models: the arenas of the launches stay in HBM (one, unless a column overflowed)
    struct ArenaSeg { int32_t* ind; float* val; int64_t n; bool owned; };
    struct ArenaSegs : std::vector<ArenaSeg> {
      ~ArenaSegs() { for (auto& g : *this) if (g.owned) { (void)hipFree(g.ind); (void)hipFree(g.val); } }
    } arena_segs;
    std::vector<int32_t> pending = order;  // columns still to solve
    double kernel_ms = 0;
    // results per column (host), filled as launches complete
    std::vector<int32_t> fin_cnt((size_t)ncols, 0);
    std::vector<int64_t> fin_off((size_t)ncols, 0);
    int64_t fin_total = 0;

    // screen-sum cache: read when this very work list was solved last time in this geometry,
    // else record (if the [tiles][ncols][P] array fits comfortably next to everything else)
    int gram_mode = 0;
    float* d_gram = nullptr;
    const int gram_geom_now[6] = {tileP, clusterK, nheavy > 0 ? clusterHi : 0, nheavy,
                                  opt.shard_count, opt.shard_index};
    if (use_tile && !opt.build_G && !std::getenv("SLIM_GPU_NO_GRAM")) {
      const size_t ngroups0 = ((size_t)nwork + tileP - 1) / tileP;
      const size_t need = ngroups0 * tile_x * sizeof(float);
      if (!m->gram_order.empty() && m->gram_order == order && m->ws_gram.bytes >= need &&
          std::equal(gram_geom_now, gram_geom_now + 6, m->gram_geom)) {
        gram_mode = 2;
        d_gram = static_cast<float*>(m->ws_gram.p);
      } else {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // (a quarter of what is free, 32 GB at most: a second handle or a replica on the same
        // device must still find room -- and the cache is dropped whenever a workspace needs it)
        if (need <= (size_t(32) << 30) && need <= (free_b + m->ws_gram.bytes) / 4) {
          m->gram_order.clear();  // (invalid while it is being rewritten)
          d_gram = ws_get<float>(m->ws_gram, need / sizeof(float));
          gram_mode = 1;
        }
      }
    }

    bool cluster_fallback = false;
    // g carried between the solves of a grid (slimgpu_model::d_gsave): all columns of an unsharded
    // solve on the packed kernel, the model staying in HBM
    float* carry_buf = nullptr;
    int64_t carry_stride = 0;
    bool carry_from_warm = false;
    if (resident && use_gramr && std::max(gramr_kr, 1) + gramr_kl <= kGramrCarryMaxGroups && !columns && nwork == ncols &&
        opt.shard_count == 1 && !std::getenv("SLIM_GPU_NO_CARRY")) {
      carry_stride = (int64_t)(std::max(gramr_kr, 1) + gramr_kl) * 8192;
      const size_t carry_bytes = sizeof(float) * (size_t)ncols * (size_t)carry_stride;
      if (warm_dev && warm_dev->d_gsave && warm_dev->gsave_owner == m->uid && warm_dev->gsave_stride == carry_stride &&
          warm_dev->n == ncols) {
        // the previous model's buffer: read where l1 is the same, overwritten in place either way, and
        // no longer that model's from here on (a failure below leaves nothing half-valid behind)
        carry_from_warm = warm_dev->gsave_valid && warm_dev->gsave_l1 == opt.l1r;
        carry_buf = warm_dev->d_gsave;
        warm_dev->d_gsave = nullptr;
        warm_dev->gsave_valid = false;
      } else {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (carry_bytes <= (size_t(8) << 30) && carry_bytes * 4 <= free_b) carry_buf = dev_alloc<float>((size_t)ncols * (size_t)carry_stride);
      }
    }
    struct CarryGuard {  // (freed unless a model takes it over)
      float*& p;
      ~CarryGuard() { if (p) (void)hipFree(p); }
    } carry_guard{carry_buf};
    for (int attempt = 0; attempt < 8 && !pending.empty(); ++attempt) {
      const int32_t npend = (int32_t)pending.size();
      int32_t* d_ai = ws_get<int32_t>(m->ws_arena_i, (size_t)arena_cap, m);
      float* d_av = ws_get<float>(m->ws_arena_v, (size_t)arena_cap, m);
      if (gram_mode != 0 && m->ws_gram.p != d_gram) {
        // the arena did not fit next to the screen-sum cache and ws_get gave the cache up
        // (drop_screen_cache): this launch neither records nor reads it
        gram_mode = 0;
        d_gram = nullptr;
      }
      HIP_TRY(hipMemcpyAsync(d_order, pending.data(), sizeof(int32_t) * (size_t)npend,
                             hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemsetAsync(d_misc, 0, sizeof(int32_t) * 16, stream));
      if (attempt > 0) nheavy = 0;  // a retry regroups what is left: plain clusters

      SolveArgs S;
      S.l1 = (float)opt.l1r;
      S.l2 = (float)opt.l2r;
      S.opt_tol = (float)opt.optTol;
      S.maxniters = opt.maxniters;
      S.seed = opt.seed;
      S.nnbrs = opt.nnbrs;
      S.simtype = opt.simtype;
      S.order = d_order;
      S.nwork = npend;
      S.queue = d_misc;
      S.icolptr = d_icolptr;
      S.icolind = d_icolind;
      S.icolval = d_icolval;
      S.incols = incols;
      S.slab = d_slab;
      S.slab_stride = use_tile ? (int64_t)tile_r : (int64_t)vec_floats;
      S.nrows_pad = nrows_pad;
      S.ncols_pad = ncols_pad;
      S.xslab = d_xslab;
      S.x_stride = (int64_t)tile_x;
      S.ulist = d_ulist;
      S.u_stride = (int64_t)tile_u;
      S.ngroups = (npend + tileP - 1) / tileP;
      if (opt.build_G && opt.G_rows_end >= 0) S.ngroups = (G_block + tileP - 1) / tileP;
      S.cluster = clusterK;
      S.ubounds = use_tile ? m->d_ubounds[cluster_lg] : nullptr;
      S.csplit = use_tile ? m->d_csplit[cluster_lg] : nullptr;
      S.mailbox = d_mailbox;
      // (FSLIM on ratings that can cancel: the fixed-order pass also counts co-ratings, so that
      // a candidate whose sum is 0 stays a candidate -- neighbors.c:46-60 marks every co-rated item)
      S.exact_gram = (m->exact_gram || std::getenv("SLIM_GPU_EXACT_GRAM") ||
                      (opt.nnbrs > 0 && m->nonpositive)) ? 1 : 0;
      S.atypart = d_part;
      S.bm_shift = bm_shift;
      S.bm_words = bm_words;
      S.nheavy = use_tile ? nheavy : 0;
      S.cluster_hi = clusterHi;
      S.ubounds_hi = nheavy > 0 ? m->d_ubounds[hi_lg] : nullptr;
      S.csplit_hi = nheavy > 0 ? m->d_csplit[hi_lg] : nullptr;
      S.mailbox_hi = d_mailbox ? d_mailbox + (size_t)std::max(nclusters, 1) * mailbox_stride : nullptr;
      S.queue_hi = d_misc + 4;
      S.hi_prefetch = 1;
      S.shard_count = opt.shard_count;
      S.shard_index = opt.shard_index;
      S.nnz_last = m->nnz > 0 ? m->nnz - 1 : 0;
      S.xcd_swizzle = 0;
      // (valid for the first launch over the whole work list only: a retry solves a subset,
      // the fallback another geometry)
      S.gram_mode = (attempt == 0 && !cluster_fallback) ? gram_mode : 0;
      S.gram = d_gram;
      S.G = static_cast<float*>(m->ws_G.p);  // (nullptr once dropped: only the float kernels read it)
      S.G_ld = m->G_ld;
      S.tile_nunion = d_nunion;
      S.gram_pos = nullptr;
      // (2: lanes = columns, bit-sliced counters; SLIM_GPU_GBITS=1: the round-4 form, one column
      // per wavefront and 32 ballots per 64 nnz)
      S.gram_bits = (gram_bits_lds && clusterK == 32 && tile_lds == gram_bits_lds) ? 2 : 0;
      if (const char* e = std::getenv("SLIM_GPU_GBITS"); e && S.gram_bits) S.gram_bits = std::atoi(e) == 1 ? 1 : 2;
      S.gram_split_stride = clusterK + 1;
      S.gram_accum = 0;
      S.g_save = nullptr;
      S.g_load = nullptr;
      S.g_stride = 0;
      if (carry_buf) {  // (resident models on the packed kernel: g carried from pair to pair, see slimgpu_model)
        S.g_save = carry_buf;
        S.g_stride = carry_stride;
        // (a retry re-solves a column whose slot already holds this solve's result: it folds again)
        S.g_load = (carry_from_warm && attempt == 0) ? carry_buf : nullptr;
      }
      if (gram_passes > 1) {
        if (S.gram_bits) {  // pass gram_pass of gram_passes: this launch's 32 user ranges
          S.ubounds = m->d_gubounds + 32 * gram_pass;
          S.csplit = m->d_gcsplit + 32 * gram_pass;
          S.gram_split_stride = 32 * gram_passes + 1;
          S.gram_accum = gram_pass > 0;
        } else {  // (re-planned without clusters: one launch forms all of G from the top)
          gram_passes = 1;
          gram_pass = 0;
        }
      }
      if (opt.build_G) {
        if (attempt > 0 || cluster_fallback || npend != ncols) {
          // (the symmetric fill needs every column in ONE launch; a re-plan after a cluster
          // timeout starts the fill again from the top with the whole list)
          if (npend != ncols) {
            set_error("SLIMGPU_Learn: internal: G = R^T R must be built over all columns at once");
            return fail(SLIM_ERROR);
          }
        }
        std::vector<int32_t> pos((size_t)ncols, 0);
        for (int32_t t = 0; t < npend; ++t) pos[(size_t)pending[(size_t)t]] = t;
        int32_t* d_pos = ws_get<int32_t>(m->ws_nunion, (size_t)ncols, m);
        HIP_TRY(hipMemcpyAsync(d_pos, pos.data(), sizeof(int32_t) * (size_t)ncols,
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // (pos is a local)
        S.gram_pos = d_pos;
        S.gram_mode = 3;
      }
      if (use_gram) {
        S.slab_stride = (int64_t)ncols_pad;
        S.x_stride = (int64_t)ncols_pad;
        S.u_stride = (int64_t)ncols_pad;
        S.ngroups = (npend + 31) / 32;
      }
      if (const char* e = std::getenv("SLIM_GPU_HI_PREFETCH")) S.hi_prefetch = std::atoi(e);
      if (use_tile)
        HIP_TRY(hipMemsetAsync(d_mailbox, 0, sizeof(unsigned long long) * mailbox_words, stream));
      const bool trace = use_tile && trace_level >= 1;
      S.trace = nullptr;
      if (trace) {
        S.trace = ws_get<uint64_t>(m->ws_trace, 16 * (size_t)S.ngroups);
        HIP_TRY(hipMemsetAsync(S.trace, 0, sizeof(uint64_t) * 16 * (size_t)S.ngroups, stream));
      }
      S.out_cnt = d_cnt;
      S.out_off = d_off;
      S.out_ind = d_ai;
      S.out_val = d_av;
      S.out_cursor = reinterpret_cast<unsigned long long*>(d_misc + 2);
      S.out_cap = arena_cap;
      S.overflow = d_misc + 1;
      S.st_na = d_sti;
      S.st_sweeps = d_sti + ncols;
      S.st_conv = d_sti + 2 * (size_t)ncols;
      S.st_G = d_stl;
      S.st_D = d_stl + ncols;
      S.st_U = d_stl + 2 * (size_t)ncols;
      S.st_B = d_stl + 3 * (size_t)ncols;
      S.st_err = d_stf;
      S.st_obj = d_stf + ncols;

      // clustered tiles: always launch whole clusters (every member must be resident)
      const int launch_waves =
          use_tile ? std::max(1, std::min((npend + tileP - 1) / tileP, nclusters)) * clusterK
                   : std::max(1, std::min(npend, nwaves));
      HIP_TRY(hipEventRecord(ev0, stream));
      GramPacked P{};
      if (use_gramr) {
        P.lo = static_cast<const uint8_t*>(m->ws_Glo.p);
        P.ldb = m->Gp_ldb;
        P.hi = static_cast<const uint8_t*>(m->ws_Ghi.p);
        P.hi_off = static_cast<const int64_t*>(m->ws_hioff.p);
        P.hi_k = static_cast<const int32_t*>(m->ws_hik.p);
        P.hi2_off = static_cast<const int64_t*>(m->ws_hi2off.p);
        P.hi2_k = static_cast<const int32_t*>(m->ws_hi2k.p);
        P.base = static_cast<const uint8_t*>(m->ws_Gbase.p);
        P.diag = static_cast<const float*>(m->ws_Gdiag.p);
        P.meta = static_cast<const uint4*>(m->ws_Gmeta.p);
        P.rank_of = static_cast<const int32_t*>(m->ws_rankof.p);
        P.item_of = static_cast<const int32_t*>(m->ws_itemof.p);
        P.nchunks = m->Gp_nchunks;
      }
      // the union of the active sets of every tile, read off G (inside kernel_ms): off the byte
      // planes when the packed solver runs (the floats may be gone: drop_float_gram)
      if (use_gramr)
        hipLaunchKernelGGL(gramr_union_fn(), dim3(S.ngroups), dim3(gramr_union_threads()), 0, stream, A, S, P);
      else if (use_gram)
        hipLaunchKernelGGL(gram_union_fn(), dim3(S.ngroups), dim3(64), 0, stream, A, S);
      // the heavy phase needs at least one whole big cluster in the launch
      if (S.nheavy > 0 && launch_waves < clusterHi) S.nheavy = 0;
      // test hook: launch the last cluster one member short, which is what a CU mask or a
      // second tenant does to a cluster -- exercises the timeout + fallback path below
      int launch_now = launch_waves;
      // (acts only together with the master switch SLIM_GPU_TEST_HOOKS=1: an inherited
      // environment must not void production launches)
      if (use_tile && clusterK > 1 && !cluster_fallback && test_hook("SLIM_GPU_TEST_DROP_MEMBER")) {
        launch_now -= 1;
        S.nheavy = 0;
      }
      // members of a cluster on one XCD (one L2): matters for the row-wise fold, whose x lines
      // are shared by the cluster; placement only, never correctness
      // (8 XCDs of 32 CUs on this part; asked of the device as CUs / 32, so that a partition mode
      // or another part does not get a placement that straddles XCDs: clusters must tile an XCD's
      // share of the launch)
      const int nxcd = m->num_cus % 32 == 0 ? m->num_cus / 32 : 1;
      if (use_tile && clusterK > 1 && nxcd == 8 && launch_now % 8 == 0 &&
          (launch_now / 8) % clusterK == 0 && (S.nheavy == 0 || (launch_now / 8) % clusterHi == 0)) {
        S.xcd_swizzle = 1;
        if (const char* e = std::getenv("SLIM_GPU_XCD")) S.xcd_swizzle = std::atoi(e) != 0;
      }
      if (use_gramr) {
        hipLaunchKernelGGL(fn_r, dim3(launch_now), dim3(kGramrNT), gram_lds, stream, A, S, P);
      } else
      hipLaunchKernelGGL(fn, dim3(launch_now),
                         dim3(use_gram ? 64 * gram_nw : (use_tile ? 64 * tileNW : 64)),
                         use_gram ? gram_lds : (use_lds ? lds_need : (use_tile ? tile_lds : 0)),
                         stream, A, S);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(ev1, stream));

      int32_t h_misc[4];
      HIP_TRY(hipMemcpyAsync(h_misc, d_misc, sizeof(h_misc), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt, sizeof(int32_t) * (size_t)ncols,
                             hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipMemcpyAsync(h_off.data(), d_off, sizeof(int64_t) * (size_t)ncols,
                             hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
      kernel_ms += ms;
      if (S.trace) {  // per-tile timeline: where does the launch spend its time?
        std::vector<uint64_t> tr(16 * (size_t)S.ngroups);
        HIP_TRY(hipMemcpy(tr.data(), S.trace, sizeof(uint64_t) * tr.size(), hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull, t1 = 0;
        double busy = 0, setup = 0, sweeps = 0, fold = 0;
        std::vector<double> dur;
        for (int gI = 0; gI < S.ngroups; ++gI) {
          const uint64_t* e = &tr[8 * (size_t)gI];
          t0 = std::min(t0, e[0]);
          t1 = std::max(t1, e[3]);
          const double wk = double(e[6]) / clusterK;  // heavy tiles occupy more workgroups
          busy += double(e[3] - e[0]) * wk;
          setup += double(e[1] - e[0]) * wk;
          sweeps += double(e[2] - e[1]) * wk;
          fold += double(e[7] - e[1]) * wk;  // warm-start fold (part of "sweeps")
          dur.push_back(double(e[3] - e[0]) * 1e-5);
        }
        std::sort(dur.begin(), dur.end());
        const double span = double(t1 - t0);
        std::fprintf(stderr,
                     "[trace] tiles %d (%d heavy, clusters of %d) on %d workgroups (clusters of %d): span %.2f ms (event %.2f ms), busy/"
                     "(span*wgs) %.2f, setup %.1f%% sweeps %.1f%% (fold %.1f%%) of busy; tile ms min %.2f med "
                     "%.2f p90 %.2f max %.2f\n",
                     S.ngroups, S.nheavy, S.nheavy > 0 ? clusterHi : 0, launch_waves, clusterK, span * 1e-5, ms,
                     busy * clusterK / (span * launch_waves), 100 * setup / busy, 100 * sweeps / busy,
                     100 * fold / busy,
                     dur.front(), dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back());
        if (S.ngroups >= 16) {  // queue order = cost order: (estimated cost, measured ms)
          std::fprintf(stderr, "[trace] tile cost -> ms, queue order:");
          for (int k = 0; k < 19; ++k) {
            const int gI = k < 12 ? k : (int)((int64_t)S.ngroups * (k - 11) / 8) - (k == 19 ? 1 : 0);
            if (gI >= S.ngroups) break;
            double c = 0;
            for (int t = gI * tileP; t < std::min((gI + 1) * tileP, (int)npend); ++t)
              c += (double)m->h_cost[pending[(size_t)t]];
            std::fprintf(stderr, " [%d] %.3g -> %.0f", gI, c,
                         double(tr[8 * (size_t)gI + 3] - tr[8 * (size_t)gI]) * 1e-5);
          }
          std::fprintf(stderr, "\n");
        }
        if (trace_level >= 2) {
          double ph[7] = {0, 0, 0, 0, 0, 0, 0};
          for (int gI = 0; gI < S.ngroups; ++gI)
            for (int k = 0; k < 7; ++k) ph[k] += double(tr[8 * (size_t)S.ngroups + 8 * (size_t)gI + k]);
          const double tot = ph[0] + ph[1] + ph[2] + ph[3] + ph[4];
          std::fprintf(stderr,
                       "[trace] visit phases (shader clocks/visit): loads %.0f reduce+barrier %.0f "
                       "math %.0f stores %.0f closing barrier %.0f | total %.0f; visits %.0f, "
                       "%.1f%% with update\n",
                       ph[0] / ph[5], ph[1] / ph[5], ph[2] / ph[5], ph[3] / ph[5], ph[4] / ph[5],
                       tot / ph[5], ph[5], 100 * ph[6] / ph[5]);
        }
      }

      if (h_misc[1] == 2) {
        // A cluster waited ~10 s for a member that never published: not every workgroup of the
        // launch was resident (CU mask, another tenant on the device).  The launch is void;
        // solve everything that is pending again without clusters -- that geometry has no
        // inter-workgroup dependency, so it completes on any number of compute units.
        if (!cluster_fallback && (clusterK > 1 || nheavy > 0)) {
          cluster_fallback = true;
          std::fprintf(stderr, "[slim-gpu] tile cluster timed out (workgroups not co-resident); "
                               "re-solving %d columns without clusters\n", npend);
          plan_tiles(true);
          alloc_tiles();
          --attempt;  // the void launch does not count as an arena retry
          continue;
        }
        set_error("SLIMGPU_Learn: a tile cluster timed out waiting for a member workgroup "
                  "(were all workgroups resident?)");
        return fail(SLIM_ERROR);
      }
      if (gram_passes > 1 && ++gram_pass < gram_passes) {
        --attempt;  // the same work list again, over the next user ranges
        continue;
      }
      if (S.gram_mode == 1) {  // the launch completed: its screen sums are reusable
        m->gram_order = order;
        std::copy(gram_geom_now, gram_geom_now + 6, m->gram_geom);
      }
      unsigned long long cursor;
      std::memcpy(&cursor, h_misc + 2, sizeof(cursor));
      const int64_t used = std::min<int64_t>((int64_t)cursor, arena_cap);
      const int64_t base = fin_total;
      const double t_d2h = now_ms();
      if (resident) {  // the arena stays where it is; a retry (below) moves it aside first
        arena_segs.push_back({d_ai, d_av, used, false});
      } else {
        h_ind.resize((size_t)(base + used));
        h_val.resize((size_t)(base + used));
        if (used > 0) {
          HIP_TRY(hipMemcpyAsync(h_ind.data() + base, d_ai, sizeof(int32_t) * (size_t)used,
                                 hipMemcpyDeviceToHost, stream));
          HIP_TRY(hipMemcpyAsync(h_val.data() + base, d_av, sizeof(float) * (size_t)used,
                                 hipMemcpyDeviceToHost, stream));
          HIP_TRY(hipStreamSynchronize(stream));
        }
      }
      d2h_ms += now_ms() - t_d2h;
      fin_total += used;

      std::vector<int32_t> again;
      int64_t need = 0;
      for (int32_t c : pending) {
        if (h_cnt[c] >= 0) {
          fin_cnt[c] = h_cnt[c];
          fin_off[c] = base + h_off[c];
        } else {  // did not fit the arena: solve again with a larger one
          again.push_back(c);
          need += -(int64_t)h_cnt[c] - 1;
        }
      }
      pending.swap(again);
      if (!pending.empty()) arena_cap = std::max<int64_t>(arena_cap, need + 1024);
      if (resident && !pending.empty() && used > 0) {  // the next attempt overwrites the arena
        ArenaSeg& sg = arena_segs.back();
        int32_t* ki = dev_alloc<int32_t>((size_t)used);
        float* kv = dev_alloc<float>((size_t)used);
        HIP_TRY(hipMemcpyAsync(ki, sg.ind, sizeof(int32_t) * (size_t)used, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(kv, sg.val, sizeof(float) * (size_t)used, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        sg = {ki, kv, used, true};
      }
    }
    if (!pending.empty()) {
      set_error("SLIMGPU_Learn: output arena overflow persisted");
      return fail(SLIM_ERROR_MEMORY);
    }
    const double t_kernel_done = now_ms();

    // per-column counters
    ColumnStats& cs = g_colstats;
    cs.nacols.assign((size_t)ncols, 0);
    cs.sweeps.assign((size_t)ncols, 0);
    cs.conv.assign((size_t)ncols, 0);
    cs.G.assign((size_t)ncols, 0);
    cs.D.assign((size_t)ncols, 0);
    cs.U.assign((size_t)ncols, 0);
    std::vector<float> h_err((size_t)ncols), h_obj((size_t)ncols);
    HIP_TRY(hipMemcpy(cs.nacols.data(), d_sti, sizeof(int32_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs.sweeps.data(), d_sti + ncols, sizeof(int32_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs.conv.data(), d_sti + 2 * (size_t)ncols, sizeof(int32_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs.G.data(), d_stl, sizeof(int64_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    int64_t gram_rows = 0;  // item-space kernel: rows of G it read (its byte model)
    double gram_bytes = 0;
    if (use_gram) {
      for (int32_t c : requested) gram_rows += cs.G[(size_t)c];
      if (use_gramr) {  // packed rows: the bytes each column's updates streamed, counted on the device
        std::vector<int64_t> hb((size_t)ncols);
        HIP_TRY(hipMemcpy(hb.data(), d_stl + 3 * (size_t)ncols, sizeof(int64_t) * (size_t)ncols, hipMemcpyDeviceToHost));
        for (int32_t c : requested) gram_bytes += (double)hb[(size_t)c];
      } else {
        gram_bytes = (double)gram_rows * 4.0 * (double)ncols_pad;
      }
    }
    if (use_tile || use_gram)  // the Gram work of a column is the staging pass's cost figure
      for (int32_t c : requested) cs.G[(size_t)c] = m->h_cost[(size_t)c];
    HIP_TRY(hipMemcpy(cs.D.data(), d_stl + ncols, sizeof(int64_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cs.U.data(), d_stl + 2 * (size_t)ncols, sizeof(int64_t) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_err.data(), d_stf, sizeof(float) * (size_t)ncols, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_obj.data(), d_stf + ncols, sizeof(float) * (size_t)ncols, hipMemcpyDeviceToHost));

    // SaveModel (estimate.c:570-593): concatenate the columns, then the row view
    const double t_counters_done = now_ms();
    int64_t tnnz = 0;
    for (int32_t c = 0; c < ncols; ++c) tnnz += fin_cnt[c];
    slim_csr_t* model = nullptr;
    double t_columns_done = t_counters_done;
    ssize_t* colptr = nullptr;
    int32_t* colind = nullptr;
    float* colval = nullptr;
    if (resident) {
      // the same two steps on the device: the arena's columns gathered into column order, the row
      // view by the staging pass's stable sort (engine.hip: transpose_on_device) -- nothing crosses
      // PCIe unless the caller fetches the model (model_fetch)
      std::unique_ptr<slimgpu_model> dm(new slimgpu_model());
      dm->device = m->device;
      dm->n = ncols;
      dm->nnz = tnnz;
      std::vector<int64_t> h_colptr((size_t)ncols + 1, 0);
      for (int32_t c = 0; c < ncols; ++c) h_colptr[(size_t)c + 1] = h_colptr[(size_t)c] + fin_cnt[(size_t)c];
      const int32_t* src_i = arena_segs.empty() ? nullptr : arena_segs[0].ind;
      const float* src_v = arena_segs.empty() ? nullptr : arena_segs[0].val;
      int32_t* cat_i = nullptr;
      float* cat_v = nullptr;
      if (arena_segs.size() > 1) {  // (a column overflowed its arena: the launches' pieces, in order)
        cat_i = dev_alloc<int32_t>((size_t)std::max<int64_t>(fin_total, 1));
        cat_v = dev_alloc<float>((size_t)std::max<int64_t>(fin_total, 1));
        int64_t at = 0;
        for (const ArenaSeg& sg : arena_segs) {
          if (sg.n > 0) {
            HIP_TRY(hipMemcpyAsync(cat_i + at, sg.ind, sizeof(int32_t) * (size_t)sg.n, hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(cat_v + at, sg.val, sizeof(float) * (size_t)sg.n, hipMemcpyDeviceToDevice, stream));
          }
          at += sg.n;
        }
        src_i = cat_i;
        src_v = cat_v;
      }
      dm->d_colptr = dev_alloc<int64_t>((size_t)ncols + 1);
      dm->d_colind = dev_alloc<int32_t>((size_t)std::max<int64_t>(tnnz, 1));
      dm->d_colval = dev_alloc<float>((size_t)std::max<int64_t>(tnnz, 1));
      int64_t* d_src = ws_get<int64_t>(m->ws_off, (size_t)ncols, m);  // (the solver's own offsets: done with)
      HIP_TRY(hipMemcpyAsync(dm->d_colptr, h_colptr.data(), sizeof(int64_t) * ((size_t)ncols + 1),
                             hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(d_src, fin_off.data(), sizeof(int64_t) * (size_t)ncols, hipMemcpyHostToDevice, stream));
      if (tnnz > 0) {
        hipLaunchKernelGGL(k_gather_columns, dim3(grid_for((int64_t)ncols * 64, 256, m->num_cus * 16)), dim3(256),
                           0, stream, ncols, dm->d_colptr, d_src, src_i, src_v, dm->d_colind, dm->d_colval);
        HIP_TRY(hipGetLastError());
      }
      t_columns_done = now_ms();
      if (row_view) transpose_on_device(m, ncols, tnnz, dm->d_colptr, dm->d_colind, dm->d_colval, &dm->d_rowptr,
                                        &dm->d_rowind, &dm->d_rowval);
      HIP_TRY(hipStreamSynchronize(stream));
      if (cat_i) (void)hipFree(cat_i);
      if (cat_v) (void)hipFree(cat_v);
      if (carry_buf) {
        dm->d_gsave = carry_buf;
        dm->gsave_valid = true;
        dm->gsave_stride = carry_stride;
        dm->gsave_l1 = opt.l1r;
        dm->gsave_owner = m->uid;
        carry_buf = nullptr;  // (the model's now)
      }
      *rio->out = dm.release();
    } else {
    colptr = static_cast<ssize_t*>(std::malloc(sizeof(ssize_t) * ((size_t)ncols + 1)));
    colind = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(tnnz, 1)));
    colval = static_cast<float*>(std::malloc(sizeof(float) * (size_t)std::max<int64_t>(tnnz, 1)));
    if (!colptr || !colind || !colval) {
      std::free(colptr); std::free(colind); std::free(colval);
      set_error("SLIMGPU_Learn: out of host memory for the model");
      return fail(SLIM_ERROR_MEMORY);
    }
    colptr[0] = 0;
    for (int32_t c = 0; c < ncols; ++c) {
      const int64_t n = fin_cnt[c];
      if (n > 0) {
        std::memcpy(colind + colptr[c], h_ind.data() + fin_off[c], sizeof(int32_t) * (size_t)n);
        std::memcpy(colval + colptr[c], h_val.data() + fin_off[c], sizeof(float) * (size_t)n);
      }
      colptr[c + 1] = colptr[c] + n;
    }
    t_columns_done = now_ms();
    model = model_from_columns(ncols, colptr, colind, colval, row_view);
    }
    if (trace_level >= 1)
      std::fprintf(stderr, "[slim_gpu trace] host phases: prep %.0f ms, launches + D2H %.0f ms (kernel %.0f, D2H of "
                   "%lld entries %.0f), counters %.0f ms, columns %.0f ms, row view %.0f ms\n",
                   t_prep_done - t_begin, t_kernel_done - t_prep_done, kernel_ms, (long long)fin_total, d2h_ms,
                   t_counters_done - t_kernel_done, t_columns_done - t_counters_done, now_ms() - t_columns_done);

    if ((opt.dbglvl & SLIM_DBG_PROGRESS) && model) {
      // estimate.c:507-514: one line per solved column, in column order (the reference prints
      // them as its threads finish).  Everything but "a0s" comes from the counters the kernels
      // return; a0s (ComputeAvgZeroScore, estimate.c:627-662: the mean of the 10 largest
      // predicted scores among the users that did NOT rate the item) is a diagnostic that costs
      // one pass over R per column -- done here on the host, as the reference does, because
      // this switch is for eyeballing small runs.  tmr: the reference prints a timer it never
      // starts (estimate.c:377,514).
      std::vector<int64_t> hp((size_t)m->nrows + 1);
      std::vector<int32_t> hi((size_t)std::max<int64_t>(m->nnz, 1));
      std::vector<float> hv(m->binary ? 0 : (size_t)std::max<int64_t>(m->nnz, 1));
      std::vector<int64_t> hcp((size_t)ncols + 1);
      HIP_TRY(hipMemcpy(hp.data(), m->d_rowptr, sizeof(int64_t) * hp.size(), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hcp.data(), m->d_colptr, sizeof(int64_t) * hcp.size(), hipMemcpyDeviceToHost));
      if (m->nnz > 0) {
        HIP_TRY(hipMemcpy(hi.data(), m->d_rowind, sizeof(int32_t) * (size_t)m->nnz, hipMemcpyDeviceToHost));
        if (!m->binary)
          HIP_TRY(hipMemcpy(hv.data(), m->d_rowval, sizeof(float) * (size_t)m->nnz, hipMemcpyDeviceToHost));
      }
      std::vector<int32_t> sorted = requested;
      std::sort(sorted.begin(), sorted.end());
      std::vector<double> xd((size_t)ncols, 0.0);
      std::vector<char> rated((size_t)m->nrows, 0);
      std::vector<float> scores;
      for (int32_t c : sorted) {
        double nrm1 = 0.0;
        for (ssize_t k = colptr[c]; k < colptr[c + 1]; ++k) {
          xd[(size_t)colind[k]] = colval[k];
          nrm1 += colval[k];
        }
        scores.clear();
        for (int32_t u = 0; u < m->nrows; ++u) {
          bool has = false;
          double r = 0.0;
          for (int64_t e = hp[(size_t)u]; e < hp[(size_t)u + 1]; ++e) {
            if (hi[(size_t)e] == c) has = true;
            r += xd[(size_t)hi[(size_t)e]] * (m->binary ? 1.0 : (double)hv[(size_t)e]);
          }
          if (!has) scores.push_back((float)r);
        }
        const size_t ntop = std::min<size_t>(10, scores.size());
        std::partial_sort(scores.begin(), scores.begin() + (ptrdiff_t)ntop, scores.end(),
                          std::greater<float>());
        float a0 = 0.0f;
        for (size_t k = 0; k < ntop; ++k) a0 += scores[k];
        for (ssize_t k = colptr[c]; k < colptr[c + 1]; ++k) xd[(size_t)colind[k]] = 0.0;
        std::printf("Col: %5d %5zd rs: %3d nits: %4d nnz: %4d rsd: %.2le obj: %.2le ff: %.3lf nrm1: "
                    "%.3lf a0s: %.3lf tmr: %.2le\n",
                    c, (ssize_t)(hcp[(size_t)c + 1] - hcp[(size_t)c]), cs.conv[(size_t)c],
                    cs.sweeps[(size_t)c], (int)(colptr[c + 1] - colptr[c]), (double)h_err[(size_t)c],
                    (double)h_obj[(size_t)c],
                    h_obj[(size_t)c] != 0 ? (double)h_err[(size_t)c] / (double)h_obj[(size_t)c] : 0.0,
                    nrm1, ntop ? (double)a0 / (double)ntop : 0.0, 0.0);
      }
      std::fflush(stdout);
    }

    st.ncols_solved = nwork;
    st.kernel = kernel;
    st.nwaves = nwaves;
    st.lds_bytes = use_lds ? (int32_t)lds_need : 0;
    st.setup_ms = m->setup_ms;
    st.kernel_ms = kernel_ms;
    for (int32_t c : requested) {
      st.G += cs.G[c];
      st.D += cs.D[c];
      st.U += cs.U[c];
      st.sweeps += cs.sweeps[c];
      st.error += h_err[c];
      st.objval += h_obj[c];
    }
    st.nnzW = tnnz;
    st.alg_bytes = m->binary
                       ? 4.0 * st.G + 8.0 * st.D + 4.0 * st.U + 8.0 * st.nnzW
                       : 8.0 * st.G + 12.0 * st.D + 4.0 * st.U + 8.0 * st.nnzW;
    st.gather_ms = now_ms() - t_kernel_done;
    st.gram_build_ms = use_gram ? m->G_build_ms : 0.0;
    st.gram_alloc_ms = use_gram ? m->G_alloc_ms : 0.0;
    st.gram_sums_ms = use_gram ? m->G_sums_ms : 0.0;
    st.gram_sums_kernel_ms = use_gram ? m->G_sums_kernel_ms : 0.0;
    st.gram_pack_ms = use_gram ? m->G_pack_ms : 0.0;
    st.gram_rows = gram_rows;
    st.gram_bytes = gram_bytes;
    if (use_gram) m->G_build_ms = m->G_alloc_ms = m->G_sums_ms = m->G_sums_kernel_ms = m->G_pack_ms = 0.0;  // (charged to the solve that paid for it)
    if (!opt.build_G) m->last_order = requested;
    st.total_ms = now_ms() - t_begin;
    g_stats = st;
    if (opt.dbglvl & SLIM_DBG_INFO)  // estimate.c:552-555
      std::printf("Done estimation: loss: %.5le, fit: %.5le, ffrac: %.3lf,  #nzs: %zd\n", st.objval,
                  st.error, st.objval != 0 ? st.error / st.objval : 0.0, (ssize_t)tnnz);
    if (status) *status = SLIM_OK;
    return model;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_Learn");
    return fail(status_of(e));
  } catch (const std::bad_alloc&) {
    set_error("SLIMGPU_Learn: out of host memory");
    return fail(SLIM_ERROR_MEMORY);
  }
}

// -- models resident in HBM (engine.hpp) -----------------------------------------------------
slimgpu_model* learn_resident(slimgpu_matrix_t* m, const LearnOptions& opt, const slimgpu_model* warm,
                              int32_t* status) {
  if (m && !m->replicas.empty()) {
    set_error("SLIMGPU_LearnResident: a model resident in HBM belongs to one device (ngpus = 1)");
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  slimgpu_model* out = nullptr;
  ResidentIO rio;
  rio.warm = warm;
  rio.out = &out;
  int32_t st = SLIM_ERROR;
  (void)learn_cd(m, opt, nullptr, &st, nullptr, 0, /*row_view=*/true, &rio);
  if (status) *status = st;
  if (st != SLIM_OK && out) {
    model_free(out);
    out = nullptr;
  }
  return out;
}

int32_t model_row_view(const slimgpu_model* w, DeviceRowView* out) {
  if (!w || !w->d_rowptr || !out) {
    set_error("resident model: no row view");
    return SLIM_ERROR_INPUT;
  }
  try {
    HIP_TRY(hipSetDevice(w->device));
    std::vector<int64_t> rp((size_t)w->n + 1);
    HIP_TRY(hipMemcpy(rp.data(), w->d_rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost));
    out->nrows = out->ncols = w->n;
    out->nnz = w->nnz;
    out->max_row = 0;
    for (int32_t r = 0; r < w->n; ++r) out->max_row = std::max<int64_t>(out->max_row, rp[(size_t)r + 1] - rp[(size_t)r]);
    out->d_ptr = w->d_rowptr;
    out->d_ind = w->d_rowind;
    out->d_val = w->d_rowval;
    return SLIM_OK;
  } catch (const HipError& e) {
    report(e, "resident model");
    return status_of(e);
  }
}

int64_t model_nnz(const slimgpu_model* w) { return w ? w->nnz : -1; }
int32_t model_ncols(const slimgpu_model* w) { return w ? w->n : -1; }

namespace {
// D2H of both views into arrays the host model owns (csr_free releases them)
void fetch_now(slimgpu_model* w) {
  const double t0 = now_ms();
  ssize_t *cp = nullptr, *rp = nullptr;
  int32_t *ci = nullptr, *ri = nullptr;
  float *cv = nullptr, *rv = nullptr;
  hipStream_t cs = nullptr;
  try {
    HIP_TRY(hipSetDevice(w->device));
    static_assert(sizeof(ssize_t) == sizeof(int64_t), "offsets travel as they are");
    const size_t n1 = (size_t)w->n + 1, nz = (size_t)std::max<int64_t>(w->nnz, 1);
    cp = static_cast<ssize_t*>(std::malloc(sizeof(ssize_t) * n1));
    rp = static_cast<ssize_t*>(std::malloc(sizeof(ssize_t) * n1));
    ci = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * nz));
    ri = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * nz));
    cv = static_cast<float*>(std::malloc(sizeof(float) * nz));
    rv = static_cast<float*>(std::malloc(sizeof(float) * nz));
    if (!cp || !rp || !ci || !ri || !cv || !rv) throw std::bad_alloc();
    // its own stream: the copies run on the DMA engines beside whatever the solver's stream is doing
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(cp, w->d_colptr, sizeof(int64_t) * n1, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipMemcpyAsync(rp, w->d_rowptr, sizeof(int64_t) * n1, hipMemcpyDeviceToHost, cs));
    if (w->nnz > 0) {
      HIP_TRY(hipMemcpyAsync(ci, w->d_colind, sizeof(int32_t) * (size_t)w->nnz, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(cv, w->d_colval, sizeof(float) * (size_t)w->nnz, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(ri, w->d_rowind, sizeof(int32_t) * (size_t)w->nnz, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(rv, w->d_rowval, sizeof(float) * (size_t)w->nnz, hipMemcpyDeviceToHost, cs));
    }
    HIP_TRY(hipStreamSynchronize(cs));
    (void)hipStreamDestroy(cs);
    cs = nullptr;
    slim_csr_t* hm = model_from_columns(w->n, cp, ci, cv, /*row_view=*/false);
    if (!hm) throw std::bad_alloc();
    hm->rowptr = rp;
    hm->rowind = ri;
    hm->rowval = rv;
    w->fetched = hm;
    w->fetch_status = SLIM_OK;
  } catch (const HipError& e) {
    w->fetch_error = "SLIMGPU_ModelFetch: " + e.where + ": " + hipGetErrorString(e.code);
    w->fetch_status = SLIM_ERROR;
  } catch (const std::bad_alloc&) {
    w->fetch_error = "SLIMGPU_ModelFetch: out of host memory";
    w->fetch_status = SLIM_ERROR_MEMORY;
  }
  if (w->fetch_status != SLIM_OK) {
    if (cs) (void)hipStreamDestroy(cs);
    std::free(cp); std::free(rp); std::free(ci); std::free(ri); std::free(cv); std::free(rv);
  }
  w->fetch_ms = now_ms() - t0;
}
}  // namespace

int32_t model_fetch_begin(slimgpu_model* w) {
  if (!w || !w->d_rowptr) {
    set_error("SLIMGPU_ModelFetchBegin: null model");
    return SLIM_ERROR_INPUT;
  }
  if (w->fetch_begun) return SLIM_OK;
  w->fetch_begun = true;
  w->fetched = nullptr;
  w->fetch_status = SLIM_OK;
  try {
    w->fetcher = std::thread(fetch_now, w);
  } catch (const std::system_error&) {  // no thread: the fetch happens in model_fetch
    w->fetch_begun = false;
  }
  return SLIM_OK;
}

slim_csr_t* model_fetch(slimgpu_model* w, int32_t* status, double* ms) {
  if (!w || !w->d_rowptr) {
    set_error("SLIMGPU_ModelFetch: null model");
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  if (w->fetch_begun) {
    if (w->fetcher.joinable()) w->fetcher.join();
    w->fetch_begun = false;
  } else {
    fetch_now(w);
  }
  slim_csr_t* hm = w->fetched;  // the caller's from here on (SLIM_FreeModel); a later fetch copies again
  w->fetched = nullptr;
  if (w->fetch_status != SLIM_OK) set_error(w->fetch_error);
  if (status) *status = w->fetch_status;
  if (ms) *ms = w->fetch_ms;
  return hm;
}

void model_free(slimgpu_model* w) {
  if (!w) return;
  if (w->fetcher.joinable()) w->fetcher.join();
  if (w->fetched) csr_free(w->fetched);
  (void)hipSetDevice(w->device);
  (void)hipFree(w->d_colptr); (void)hipFree(w->d_colind); (void)hipFree(w->d_colval);
  (void)hipFree(w->d_rowptr); (void)hipFree(w->d_rowind); (void)hipFree(w->d_rowval);
  (void)hipFree(w->d_gsave);
  delete w;
}

// -- G = R^T R in row blocks (engine.hpp) ---------------------------------------------------
int32_t gram_build_rows(slimgpu_matrix_t* m, int32_t row_begin, int32_t row_end) {
  if (!m || row_begin < 0 || row_end > m->ncols || row_begin > row_end) {
    set_error("SLIMGPU_MatrixGramBuildRows: rows outside [0, ncols)");
    return SLIM_ERROR_INPUT;
  }
  try {
    HIP_TRY(hipSetDevice(m->device));
    const int32_t ncols = m->ncols;
    const int64_t G_ld = round_up(round_up(ncols, 64), 64);
    const size_t G_bytes = sizeof(float) * (size_t)ncols * (size_t)G_ld;
    if (m->G_ready || m->ws_G.bytes < G_bytes) {  // a fresh G: nothing of an earlier one is kept
      size_t free_b = 0, total_b = 0;
      HIP_TRY(hipMemGetInfo(&free_b, &total_b));
      if (G_bytes + (size_t(8) << 30) > free_b + m->ws_G.bytes + m->ws_gram.bytes) {
        set_error("SLIMGPU_MatrixGramBuildRows: G = R^T R (4 ncols^2 bytes) does not fit the free HBM");
        return SLIM_ERROR_MEMORY;
      }
      drop_screen_cache(m);
      float* dG = ws_get<float>(m->ws_G, (size_t)ncols * (size_t)G_ld);
      HIP_TRY(hipMemsetAsync(dG, 0, G_bytes, m->stream));
      m->G_ld = G_ld;
      m->G_ready = false;
      m->Gp_ready = false;
      m->Gp_tried = false;
    }
    if (row_begin == row_end) return SLIM_OK;
    LearnOptions bo;
    bo.kernel = SLIMGPU_KERNEL_TILE;
    bo.build_G = true;
    bo.G_rows_begin = row_begin;
    bo.G_rows_end = row_end;
    bo.heavy_tiles = 0;
    int32_t bst = SLIM_OK;
    slim_csr_t* none = learn_cd(m, bo, nullptr, &bst, nullptr, 0, false);
    if (!none) return bst;
    csr_free(none);
    return SLIM_OK;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixGramBuildRows");
    return status_of(e);
  }
}

int32_t gram_view(slimgpu_matrix_t* m, void** dptr, int64_t* ld, int32_t* nrows) {
  if (!m || !m->ws_G.p || m->G_ld <= 0) {
    set_error("SLIMGPU_MatrixGramView: no G on this handle (SLIMGPU_MatrixGramBuildRows first)");
    return SLIM_ERROR_INPUT;
  }
  if (dptr) *dptr = m->ws_G.p;
  if (ld) *ld = m->G_ld;
  if (nrows) *nrows = m->ncols;
  return SLIM_OK;
}

int32_t gram_commit(slimgpu_matrix_t* m) {
  if (!m || !m->ws_G.p || m->G_ld <= 0) {
    set_error("SLIMGPU_MatrixGramCommit: no G on this handle");
    return SLIM_ERROR_INPUT;
  }
  try {
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->G_ready = true;
    m->Gp_ready = false;
    m->Gp_tried = false;
    m->Gf_dropped = false;
    (void)pack_gram(m);  // (false: G is not integer-valued or the planes do not fit -- float kernels)
    drop_float_gram(m);
    return SLIM_OK;
  } catch (const HipError& e) {
    report(e, "SLIMGPU_MatrixGramCommit");
    return status_of(e);
  }
}

}  // namespace slimamd
