// admm.hip -- SLIM_Learn(algo = admm): the dense ADMM solver of the reference
// (/root/reference/src/libslim/estimate.c:38-304, built there only with MKL) on the GPU.
//
// The reference forms T = R^T R (m x m, m = number of items, fp64), P = (T + (l2 + rho) I)^-1 by
// a Cholesky factorisation and inversion, A = P T, and runs 30 iterations of
//     W <- rho W - C;  T <- P W + A;  gamma_j = T_jj / P_jj;  B = T - P diag(gamma)
//     W <- max(soft(B + C / rho, l1 / rho), 0);  B <- rho (B - W);  C <- C + B
// with rho = 1e4 (estimate.c:48-49, 166-213); imodel, optTol and niters are ignored there and
// here.  Everything is dense m x m, so this is the one GEMM-shaped corner of the library:
//   * R^T R: one wavefront per user row, fp64 atomics into T (hand-written; exact for integer
//     ratings, order-independent to 1e-16 otherwise);
//   * Cholesky + inverse (the reference: LAPACKE_dpotrf / dpotri, estimate.c:150-155): a blocked
//     right-looking factorisation -- the 64 x 64 diagonal blocks by a hand-written one-workgroup
//     kernel in LDS, the panel solve and the trailing update by rocBLAS dtrsm / dsyrk -- then
//     L^-1 by dtrsm and P^-1 = L^-T L^-1 by dgemm; the products P T and P W: rocBLAS dgemm (fp64
//     MFMA).  Plain library calls, rocBLAS only (loaded on demand with dlopen, so the CD path
//     does not pay for it; rocSOLVER's 888 MB shared object is not needed at all);
//   * everything elementwise (the whole iteration except the product) is one fused kernel with
//     the reference's rounding sequence (separate multiplies and adds, no contraction), plus a
//     m-thread kernel for gamma;
//   * the positive entries of W are compacted into the model's row view (estimate.c:215-262).
// Memory: 6 m x m doubles (m = 20 000: 19 GB; m = 100 000 does not fit any GPU, nor the
// reference's host).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "engine.hpp"
#include "host_csr.hpp"

namespace slimamd {

namespace {

struct HipFail {
  hipError_t code;
  const char* where;
};
#define ADMM_TRY(expr)                                          \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) throw HipFail{_e, #expr};             \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  explicit DevBuf(size_t n) { ADMM_TRY(hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * (n ? n : 1))); }
  ~DevBuf() { (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

// ---- the two libraries, resolved at first use ------------------------------------------
struct DenseLibs {
  void* blas = nullptr;
  int (*create_handle)(void**) = nullptr;
  int (*destroy_handle)(void*) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  int (*dgemm)(void*, int, int, int, int, int, const double*, const double*, int, const double*,
               int, const double*, double*, int) = nullptr;
  int (*dtrsm)(void*, int, int, int, int, int, int, const double*, const double*, int, double*,
               int) = nullptr;
  int (*dsyrk)(void*, int, int, int, int, const double*, const double*, int, const double*,
               double*, int) = nullptr;
  bool load(std::string* err) {
    blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!blas) blas = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
    if (!blas) {
      *err = "algo=admm needs librocblas.so (not found)";
      return false;
    }
    create_handle = reinterpret_cast<decltype(create_handle)>(dlsym(blas, "rocblas_create_handle"));
    destroy_handle = reinterpret_cast<decltype(destroy_handle)>(dlsym(blas, "rocblas_destroy_handle"));
    set_stream = reinterpret_cast<decltype(set_stream)>(dlsym(blas, "rocblas_set_stream"));
    dgemm = reinterpret_cast<decltype(dgemm)>(dlsym(blas, "rocblas_dgemm"));
    dtrsm = reinterpret_cast<decltype(dtrsm)>(dlsym(blas, "rocblas_dtrsm"));
    dsyrk = reinterpret_cast<decltype(dsyrk)>(dlsym(blas, "rocblas_dsyrk"));
    if (!create_handle || !destroy_handle || !set_stream || !dgemm || !dtrsm || !dsyrk) {
      *err = "algo=admm: rocBLAS entry points missing";
      return false;
    }
    return true;
  }
};
constexpr int kOpNone = 111, kOpTrans = 112;  // rocblas_operation_none / _transpose
constexpr int kFillLower = 122;               // rocblas_fill_lower
constexpr int kNonUnit = 131;                 // rocblas_diagonal_non_unit
constexpr int kSideLeft = 141, kSideRight = 142;
constexpr int kNB = 64;                       // Cholesky block

// ---- kernels ---------------------------------------------------------------------------

// T += row_u^T row_u, one wavefront per user (estimate.c:124-125, mkl_sparse_d_spmmd)
__global__ void k_gram_dense(int32_t nrows, int32_t m, const int64_t* __restrict__ rowptr,
                             const int32_t* __restrict__ rowind, const float* __restrict__ rowval,
                             double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < nrows; u += nwaves) {
    const int64_t s = rowptr[u], e = rowptr[u + 1];
    for (int64_t a = s; a < e; ++a) {
      const int64_t ia = rowind[a];
      const double va = rowval ? (double)rowval[a] : 1.0;
      for (int64_t b = s + lane; b < e; b += 64)
        atomicAdd(&T[ia * m + rowind[b]], va * (rowval ? (double)rowval[b] : 1.0));
    }
  }
}

__global__ void k_copy_add_diag(int64_t n2, int32_t m, const double* __restrict__ T,
                                double* __restrict__ P, double diag) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2;
       k += (int64_t)gridDim.x * blockDim.x)
    P[k] = T[k] + ((k / m == k % m) ? diag : 0.0);
}

// Cholesky factor of one kb x kb diagonal block (column-major, lower), in LDS, one workgroup of
// 64 threads: thread i owns row i.  *info becomes the 1-based global index of a non-positive pivot.
__global__ __launch_bounds__(kNB) void k_potf2(int32_t kb, int32_t k0, int32_t lda, double* __restrict__ Ablk,
                                               int32_t* __restrict__ info) {
  __shared__ double a[kNB][kNB + 1];
  const int i = threadIdx.x;
  for (int j = 0; j < kb; ++j) a[i][j] = (i < kb && j <= i) ? Ablk[(int64_t)j * lda + i] : 0.0;
  __syncthreads();
  for (int j = 0; j < kb; ++j) {
    const double d = a[j][j];
    if (!(d > 0.0)) {
      if (i == 0) atomicCAS(info, 0, k0 + j + 1);
      return;
    }
    const double piv = sqrt(d);
    __syncthreads();
    if (i == j) a[j][j] = piv;
    if (i > j && i < kb) a[i][j] = a[i][j] / piv;
    __syncthreads();
    if (i > j && i < kb)
      for (int c = j + 1; c <= i; ++c) a[i][c] -= a[i][j] * a[c][j];
    __syncthreads();
  }
  if (i < kb)
    for (int j = 0; j <= i; ++j) Ablk[(int64_t)j * lda + i] = a[i][j];
}

__global__ void k_identity(int64_t n2, int32_t m, double* __restrict__ X) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2;
       k += (int64_t)gridDim.x * blockDim.x)
    X[k] = (k / m == k % m) ? 1.0 : 0.0;
}

// estimate.c:160-163: the lower triangle (row-major) takes the computed upper one
__global__ void k_symmetrize(int32_t m, double* __restrict__ P) {
  const int64_t n2 = (int64_t)m * m;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = k / m, j = k % m;
    if (j < i) P[k] = P[j * m + i];
  }
}

// W <- rho W - C (estimate.c:168-172): scal, then axpy -- two roundings
__global__ void k_w_pre(int64_t n2, double rho, double* __restrict__ W, const double* __restrict__ C) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2;
       k += (int64_t)gridDim.x * blockDim.x)
    W[k] = __dadd_rn(__dmul_rn(rho, W[k]), -C[k]);
}

// gamma_j = (T_jj + A_jj) / P_jj (estimate.c:178-183)
__global__ void k_gamma(int32_t m, const double* __restrict__ T, const double* __restrict__ A,
                        const double* __restrict__ P, double* __restrict__ gamma) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) {
    const int64_t d = (int64_t)j * m + j;
    gamma[j] = __dadd_rn(T[d], A[d]) / P[d];
  }
}

// the rest of an iteration (estimate.c:178-213), elementwise, in the reference's rounding order
__global__ void k_iterate(int64_t n2, int32_t m, double rho, double irho, double kappa,
                          const double* __restrict__ T, const double* __restrict__ A,
                          const double* __restrict__ P, const double* __restrict__ gamma,
                          double* __restrict__ W, double* __restrict__ C) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n2;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(k % m);
    const double t = __dadd_rn(T[k], A[k]);                          // T := T + A
    double b = __dadd_rn(__dmul_rn(__dmul_rn(-1.0, P[k]), gamma[j]), t);  // B := -P diag(g) + T
    const double c = C[k];
    const double alpha = __dadd_rn(b, __dmul_rn(irho, c));
    const double hi = fmax(__dadd_rn(alpha, -kappa), 0.0);
    const double lo = fmax(__dadd_rn(-alpha, -kappa), 0.0);
    const double w = fmax(__dadd_rn(hi, -lo), 0.0);
    W[k] = w;
    b = __dmul_rn(rho, __dadd_rn(b, -w));                              // B := rho (B - W)
    C[k] = __dadd_rn(c, b);                                          // C := C + B
  }
}

__global__ void k_count_rows(int32_t m, const double* __restrict__ W, int64_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < m; i += nwaves) {
    int64_t n = 0;
    for (int j = lane; j < m; j += 64) n += W[i * m + j] > 0.0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (lane == 0) cnt[i] = n;
  }
}

__global__ void k_emit_rows(int32_t m, const double* __restrict__ W, const int64_t* __restrict__ ptr,
                            int32_t* __restrict__ ind, float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < m; i += nwaves) {
    int64_t at = ptr[i];
    for (int jb = 0; jb < m; jb += 64) {
      const int j = jb + lane;
      const double w = j < m ? W[i * m + j] : 0.0;
      const uint64_t mask = __ballot(w > 0.0);
      if (w > 0.0) {
        const int64_t dst = at + __popcll(mask & lane_lt);
        ind[dst] = j;
        val[dst] = (float)w;  // estimate.c:236
      }
      at += __popcll(mask);
    }
  }
}

int grid_for(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

slim_csr_t* learn_admm(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                       const float* rowval, const LearnOptions& opt, int32_t* status) {
  auto fail = [&](int32_t code, const std::string& msg) -> slim_csr_t* {
    set_error(msg);
    if (status) *status = code;
    return nullptr;
  };
  if (nrows < 0 || !rowptr || (rowptr[nrows] > 0 && !rowind))
    return fail(SLIM_ERROR_INPUT, "SLIM_Learn(admm): bad CSR arguments");
  const int64_t nnz = rowptr[nrows];
  int32_t m = max_index_plus_one(nnz, rowind);  // setup.c:117
  if (m <= 0) m = 1;
  for (int64_t k = 0; k < nnz; ++k)
    if (rowind[k] < 0) return fail(SLIM_ERROR_INPUT, "SLIM_Learn(admm): negative item id");
  const int64_t n2 = (int64_t)m * m;
  {
    // the same input rules as the CD path (engine.hip staging: offsets, id range, no repeated
    // (user, item) pair -- the reference would count a repeated pair twice in R^T R and once
    // per entry in the norms; neither solver defines a problem for it): a repeated item id
    // inside a row is found with one marker pass per row
    std::vector<int32_t> seen((size_t)m, -1);
    for (int32_t u = 0; u < nrows; ++u) {
      if (rowptr[u] > rowptr[u + 1] || rowptr[u] < 0 || rowptr[u + 1] > nnz)
        return fail(SLIM_ERROR_INPUT, "SLIM_Learn(admm): rowptr is not a non-decreasing offset array ending at nnz");
      for (ssize_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
        if (seen[(size_t)rowind[k]] == u)
          return fail(SLIM_ERROR_INPUT, "SLIM_Learn(admm): duplicate (user, item) entries in the rating matrix");
        seen[(size_t)rowind[k]] = u;
      }
    }
  }
  std::printf("Learning the model using ADMM... \n");  // estimate.c:41
  std::fflush(stdout);
  const bool trace = std::getenv("SLIM_GPU_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace)
      std::fprintf(stderr, "[trace] admm %-28s %8.1f ms\n", what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  try {
    (void)hipGetLastError();
    int ndev = 0;
    ADMM_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw HipFail{hipErrorNoDevice, "hipGetDeviceCount"};
    if (opt.device >= 0) ADMM_TRY(hipSetDevice(opt.device));
    size_t free_b = 0, total_b = 0;
    ADMM_TRY(hipMemGetInfo(&free_b, &total_b));
    if ((double)n2 * 8.0 * 6.5 > (double)free_b)
      return fail(SLIM_ERROR_MEMORY, "SLIM_Learn(admm): six " + std::to_string(m) + " x " +
                                         std::to_string(m) + " fp64 matrices do not fit this GPU");
    static DenseLibs libs;
    static std::once_flag once;
    static bool loaded = false;
    static std::string load_err;
    std::call_once(once, [&]() { loaded = libs.load(&load_err); });
    if (!loaded) return fail(SLIM_ERROR, "SLIM_Learn: " + load_err);
    mark("libraries resolved");
    hipDeviceProp_t prop;
    int dev = 0;
    ADMM_TRY(hipGetDevice(&dev));
    ADMM_TRY(hipGetDeviceProperties(&prop, dev));
    const int cap = prop.multiProcessorCount * 8;
    hipStream_t st = nullptr;
    ADMM_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void* handle = nullptr;
    if (libs.create_handle(&handle) != 0) {
      (void)hipStreamDestroy(st);
      return fail(SLIM_ERROR, "SLIM_Learn(admm): rocblas_create_handle failed");
    }
    libs.set_stream(handle, st);
    mark("rocblas handle");
    struct Cleanup {
      DenseLibs& l;
      void* h;
      hipStream_t s;
      ~Cleanup() {
        l.destroy_handle(h);
        (void)hipStreamDestroy(s);
      }
    } cleanup{libs, handle, st};

    // R on the device
    DevBuf<int64_t> d_ptr((size_t)nrows + 1), d_cnt((size_t)m + 1);
    DevBuf<int32_t> d_ind((size_t)nnz), d_info(1);
    DevBuf<float> d_val(rowval ? (size_t)nnz : 1);
    static_assert(sizeof(ssize_t) == sizeof(int64_t), "LP64 expected");
    ADMM_TRY(hipMemcpyAsync(d_ptr.p, rowptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, st));
    if (nnz) {
      ADMM_TRY(hipMemcpyAsync(d_ind.p, rowind, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, st));
      if (rowval)
        ADMM_TRY(hipMemcpyAsync(d_val.p, rowval, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, st));
    }
    DevBuf<double> T((size_t)n2), A((size_t)n2), P((size_t)n2), W((size_t)n2), C((size_t)n2), gam((size_t)m);
    ADMM_TRY(hipMemsetAsync(T.p, 0, sizeof(double) * (size_t)n2, st));
    ADMM_TRY(hipMemsetAsync(W.p, 0, sizeof(double) * (size_t)n2, st));
    ADMM_TRY(hipMemsetAsync(C.p, 0, sizeof(double) * (size_t)n2, st));

    const double rho = 10000.0;  // estimate.c:48
    const int maxiters = 30;     // estimate.c:49
    // T = R^T R
    hipLaunchKernelGGL(k_gram_dense, dim3(grid_for((int64_t)nrows * 64, 256, cap)), dim3(256), 0, st,
                       nrows, m, d_ptr.p, d_ind.p, rowval ? d_val.p : nullptr, T.p);
    ADMM_TRY(hipGetLastError());
    // P = (T + (l2 + rho) I)^-1 (estimate.c:139-163)
    hipLaunchKernelGGL(k_copy_add_diag, dim3(grid_for(n2, 256, cap)), dim3(256), 0, st, n2, m, T.p, P.p,
                       opt.l2r + rho);
    ADMM_TRY(hipGetLastError());
    // Blocked right-looking Cholesky of P in place (column-major lower == the row-major upper
    // triangle LAPACKE_dpotrf(ROW_MAJOR, 'U') fills, estimate.c:150), then the inverse.
    const double one = 1.0, zero = 0.0, minus_one = -1.0;
    ADMM_TRY(hipMemsetAsync(d_info.p, 0, sizeof(int32_t), st));
    for (int32_t k = 0; k < m; k += kNB) {
      const int32_t kb = std::min(kNB, m - k), rest = m - k - kb;
      double* Akk = P.p + (int64_t)k * m + k;
      hipLaunchKernelGGL(k_potf2, dim3(1), dim3(kNB), 0, st, kb, k, m, Akk, d_info.p);
      if (rest > 0) {
        double* A21 = Akk + kb;                    // rows k+kb.., columns k..k+kb
        double* A22 = Akk + (int64_t)kb * m + kb;  // trailing block
        if (libs.dtrsm(handle, kSideRight, kFillLower, kOpTrans, kNonUnit, rest, kb, &one, Akk, m, A21, m) != 0 ||
            libs.dsyrk(handle, kFillLower, kOpNone, rest, kb, &minus_one, A21, m, &one, A22, m) != 0)
          return fail(SLIM_ERROR, "SLIM_Learn(admm): rocBLAS dtrsm / dsyrk failed");
      }
    }
    ADMM_TRY(hipGetLastError());
    int32_t info = 0;
    ADMM_TRY(hipMemcpyAsync(&info, d_info.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ADMM_TRY(hipStreamSynchronize(st));
    if (info != 0) return fail(SLIM_ERROR, "SLIM_Learn(admm): R^T R + (l2 + rho) I is not positive definite");
    mark("R^T R + Cholesky");
    // X = L^-1 (W's storage is free until the iterations start), P^-1 = X^T X
    hipLaunchKernelGGL(k_identity, dim3(grid_for(n2, 256, cap)), dim3(256), 0, st, n2, m, W.p);
    if (libs.dtrsm(handle, kSideLeft, kFillLower, kOpNone, kNonUnit, m, m, &one, P.p, m, W.p, m) != 0 ||
        libs.dgemm(handle, kOpTrans, kOpNone, m, m, m, &one, W.p, m, W.p, m, &zero, P.p, m) != 0)
      return fail(SLIM_ERROR, "SLIM_Learn(admm): rocBLAS dtrsm / dgemm failed");
    ADMM_TRY(hipMemsetAsync(W.p, 0, sizeof(double) * (size_t)n2, st));
    hipLaunchKernelGGL(k_symmetrize, dim3(grid_for(n2, 256, cap)), dim3(256), 0, st, m, P.p);  // :160-163
    ADMM_TRY(hipGetLastError());
    // A = P T (row-major): column-major A^T = T^T P^T -> dgemm(T, P)
    if (libs.dgemm(handle, kOpNone, kOpNone, m, m, m, &one, T.p, m, P.p, m, &zero, A.p, m) != 0)
      return fail(SLIM_ERROR, "SLIM_Learn(admm): rocblas_dgemm failed");
    ADMM_TRY(hipStreamSynchronize(st));
    mark("inverse + A = P T");
    const double irho = 1.0 / rho, kappa = opt.l1r / rho;
    for (int it = 0; it < maxiters; ++it) {
      hipLaunchKernelGGL(k_w_pre, dim3(grid_for(n2, 256, cap)), dim3(256), 0, st, n2, rho, W.p, C.p);
      // T = P W
      if (libs.dgemm(handle, kOpNone, kOpNone, m, m, m, &one, W.p, m, P.p, m, &zero, T.p, m) != 0)
        return fail(SLIM_ERROR, "SLIM_Learn(admm): rocblas_dgemm failed");
      hipLaunchKernelGGL(k_gamma, dim3((m + 255) / 256), dim3(256), 0, st, m, T.p, A.p, P.p, gam.p);
      hipLaunchKernelGGL(k_iterate, dim3(grid_for(n2, 256, cap)), dim3(256), 0, st, n2, m, rho, irho,
                         kappa, T.p, A.p, P.p, gam.p, W.p, C.p);
      ADMM_TRY(hipGetLastError());
    }
    // the model's row view: positive entries of W, ascending j in every row
    hipLaunchKernelGGL(k_count_rows, dim3(grid_for((int64_t)m * 64, 256, cap)), dim3(256), 0, st, m, W.p,
                       d_cnt.p);
    ADMM_TRY(hipGetLastError());
    std::vector<int64_t> h_cnt((size_t)m + 1, 0);
    ADMM_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt.p, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, st));
    ADMM_TRY(hipStreamSynchronize(st));
    mark("30 iterations + row counts");
    std::vector<int64_t> h_ptr((size_t)m + 1, 0);
    for (int32_t i = 0; i < m; ++i) h_ptr[(size_t)i + 1] = h_ptr[(size_t)i] + h_cnt[(size_t)i];
    const int64_t wnnz = h_ptr[(size_t)m];
    DevBuf<int32_t> d_wind((size_t)wnnz);
    DevBuf<float> d_wval((size_t)wnnz);
    ADMM_TRY(hipMemcpyAsync(d_cnt.p, h_ptr.data(), sizeof(int64_t) * ((size_t)m + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_emit_rows, dim3(grid_for((int64_t)m * 64, 256, cap)), dim3(256), 0, st, m, W.p,
                       d_cnt.p, d_wind.p, d_wval.p);
    ADMM_TRY(hipGetLastError());
    slim_csr_t* model = csr_new();
    if (!model) return fail(SLIM_ERROR_MEMORY, "SLIM_Learn(admm): out of host memory");
    model->nrows = model->ncols = m;
    model->rowptr = static_cast<ssize_t*>(std::malloc(sizeof(ssize_t) * ((size_t)m + 1)));
    model->rowind = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(wnnz, 1)));
    model->rowval = static_cast<float*>(std::malloc(sizeof(float) * (size_t)std::max<int64_t>(wnnz, 1)));
    if (!model->rowptr || !model->rowind || !model->rowval) {
      csr_free(model);
      return fail(SLIM_ERROR_MEMORY, "SLIM_Learn(admm): out of host memory for the model");
    }
    for (int32_t i = 0; i <= m; ++i) model->rowptr[i] = (ssize_t)h_ptr[(size_t)i];
    if (wnnz) {
      ADMM_TRY(hipMemcpyAsync(model->rowind, d_wind.p, sizeof(int32_t) * (size_t)wnnz, hipMemcpyDeviceToHost, st));
      ADMM_TRY(hipMemcpyAsync(model->rowval, d_wval.p, sizeof(float) * (size_t)wnnz, hipMemcpyDeviceToHost, st));
    }
    ADMM_TRY(hipStreamSynchronize(st));
    csr_build_index(model, 0);  // + the column view every model handle of this library carries
    if (status) *status = SLIM_OK;
    return model;
  } catch (const HipFail& e) {
    return fail(e.code == hipErrorOutOfMemory ? SLIM_ERROR_MEMORY : SLIM_ERROR,
                std::string("SLIM_Learn(admm): HIP error '") + hipGetErrorString(e.code) + "' in " +
                    e.where + " -- training needs a gfx950 GPU; there is no CPU fallback");
  }
}

}  // namespace slimamd
