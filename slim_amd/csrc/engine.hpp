// engine.hpp -- device side of libslim.so: the training matrix staged in HBM and
// the driver that runs the CD kernels over a set of item columns.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/slim_gpu.h"

namespace slimamd {

// Decoded SLIM_Learn options (reference src/libslim/api.c:42-52 + slim_gpu.h).
struct LearnOptions {
  int32_t nthreads = 1, nnbrs = 0, simtype = 0, dbglvl = 0, algo = SLIM_ALGO_CD;
  int32_t ordered = 0, maxniters = 10000;
  double l1r = 1.0, l2r = 1.0, optTol = 1e-7;
  int32_t col_begin = 0, col_end = -1;  // -1: ncols
  uint32_t seed = 1;
  int32_t device = -1;  // -1: current
  int32_t kernel = SLIMGPU_KERNEL_AUTO;
  int32_t cluster = 0;  // tile-cluster size (0 = auto)
  int32_t heavy_tiles = -1;   // tiles solved by big clusters first (-1 = auto, 0 = none)
  int32_t heavy_cluster = 0;  // size of those clusters (0 = auto)
  int32_t ngpus = 1;          // SLIM_Learn & co: devices to shard over (multi_gpu.cpp)
  int32_t shard_count = 1, shard_index = 0;  // one shard of the cost-ordered work list
  bool build_G = false;  // internal: this call fills G = R^T R (engine.hip), it solves nothing
  int32_t G_rows_begin = 0, G_rows_end = -1;  // internal, with build_G: only rows [begin, end) of G (-1: all)
};
LearnOptions decode_options(const int32_t* ioptions, const double* doptions);

struct ColumnStats {
  std::vector<int32_t> nacols, sweeps, conv;
  std::vector<int64_t> G, D, U;
};

// Per-thread record of the most recent solve (SLIMGPU_LastStats & co).
slimgpu_stats_t& last_stats();
ColumnStats& last_column_stats();

// Implemented in engine.hip
slimgpu_matrix_t* matrix_from_host(int32_t nrows, const ssize_t* rowptr,
                                   const int32_t* rowind, const float* rowval,
                                   const LearnOptions& opt, int32_t* status);
slimgpu_matrix_t* matrix_from_device(int32_t nrows, int32_t ncols,
                                     const int64_t* d_rowptr, const int32_t* d_rowind,
                                     const float* d_rowval, const LearnOptions& opt,
                                     int32_t* status);
void matrix_free(slimgpu_matrix_t* m);
// copy of a staged matrix (all views, no re-sort) on another device, device to device
slimgpu_matrix_t* matrix_clone_to_device(const slimgpu_matrix_t* src, int32_t device,
                                         int32_t* status);
int32_t matrix_info(const slimgpu_matrix_t* m, int32_t* nrows, int32_t* ncols, int64_t* nnz);
int32_t matrix_get_column_view(const slimgpu_matrix_t* m, int64_t* colptr, int32_t* colind,
                               float* colval, float* cnorms);
double matrix_setup_ms(const slimgpu_matrix_t* m);
int32_t matrix_column_cost(const slimgpu_matrix_t* m, int64_t* cost);
// the caller is about to solve this matrix n times (a model-selection grid): lets the engine pay
// for G = R^T R up front (item-space CD, cd_gram.hpp)
void matrix_expect_solves(slimgpu_matrix_t* m, int32_t n);
// G = R^T R in pieces (multi-GPU: every rank forms the rows of a block of items, the blocks are
// exchanged by the caller, e.g. an RCCL broadcast per block into the view below, then committed):
// gram_build_rows fills rows [row_begin, row_end) of the handle's G (all of its ncols entries each),
// gram_view gives the device buffer (floats, ld per row), gram_commit declares every row present and
// forms the byte planes.  SLIM_OK or an error code (set_error).
int32_t gram_build_rows(slimgpu_matrix_t* m, int32_t row_begin, int32_t row_end);
int32_t gram_view(slimgpu_matrix_t* m, void** dptr, int64_t* ld, int32_t* nrows);
int32_t gram_commit(slimgpu_matrix_t* m);

// EstimateModelCD + SaveModel on the device matrix.  Returns a host model
// (slim_csr_t with both views) or nullptr with *status set.
// columns/ncolumns (optional): solve exactly these item columns (distinct ids) instead of
// the range [opt.col_begin, opt.col_end); every other column of the model comes back empty.
// rio (optional): warm start from / result into a model resident in HBM (below); with rio->out set
// the host model is not formed and nullptr comes back with *status = SLIM_OK.
struct ResidentIO {
  const slimgpu_model* warm = nullptr;
  slimgpu_model** out = nullptr;
};
slim_csr_t* learn_cd(slimgpu_matrix_t* m, const LearnOptions& opt, const slim_csr_t* imodel,
                     int32_t* status, const int32_t* columns = nullptr, int32_t ncolumns = 0,
                     bool row_view = true, ResidentIO* rio = nullptr);

// Models resident in HBM (model-selection grids: slim_mselect.c:94-113 learns 45 models one from the
// other and needs each on the host only to print its nnz): the solve leaves both views of W on the
// device, the next solve warm-starts from them without an upload, and a fetch to the host (both views,
// the arrays SaveModel would have formed) can run on the DMA engines beside the next solve.
slimgpu_model* learn_resident(slimgpu_matrix_t* m, const LearnOptions& opt, const slimgpu_model* warm,
                              int32_t* status);
int64_t model_nnz(const slimgpu_model* w);
int32_t model_ncols(const slimgpu_model* w);
int32_t model_fetch_begin(slimgpu_model* w);                      // starts the D2H on a host thread + copy stream
slim_csr_t* model_fetch(slimgpu_model* w, int32_t* status, double* ms = nullptr);  // joins it (or copies now)
void model_free(slimgpu_model* w);

// Replicas: copies of a staged matrix on other devices, owned by (and freed with) the
// primary handle; matrix_adopt_csr hands the borrowed device CSR of a FromDevice matrix over
// to the handle.
void matrix_add_replica(slimgpu_matrix_t* m, slimgpu_matrix_t* replica);
const std::vector<slimgpu_matrix_t*>& matrix_replicas(const slimgpu_matrix_t* m);
void matrix_adopt_csr(slimgpu_matrix_t* m);
int32_t matrix_device(const slimgpu_matrix_t* m);
void matrix_set_setup_ms(slimgpu_matrix_t* m, double ms);

// multi_gpu.cpp: the training matrix replicated on opt.ngpus GPUs of the node (the returned
// handle is the copy on the first device and owns the others) and a solve sharded over all
// copies of a handle, one host thread + stream per device.  With one device these are
// matrix_from_host / learn_cd.
slimgpu_matrix_t* multi_from_host(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                                  const float* rowval, const LearnOptions& opt, int32_t* status);
slim_csr_t* multi_learn(slimgpu_matrix_t* m, const LearnOptions& opt, const slim_csr_t* imodel,
                        int32_t* status, const int32_t* columns = nullptr, int32_t ncolumns = 0);

int32_t device_count();

// topn.hip: top-N lists of every row of `hist` through model W, on the GPU, bit-identical
// to host_csr.cpp::top_n.  counts (optional): list length per user.
int32_t predict_device(const slim_csr_t* W, const slim_csr_t* hist, int32_t nrcmds,
                       int32_t* output, float* scores, int32_t* counts);
// the same with the row view of W already on the device (a resident model: model_row_view)
struct DeviceRowView {
  int32_t nrows = 0, ncols = 0;
  int64_t nnz = 0, max_row = 0;  // max_row: entries of the longest row
  const int64_t* d_ptr = nullptr;
  const int32_t* d_ind = nullptr;
  const float* d_val = nullptr;
};
int32_t predict_device_view(const DeviceRowView& W, const slim_csr_t* hist, int32_t nrcmds,
                            int32_t* output, float* scores, int32_t* counts);
int32_t model_row_view(const slimgpu_model* w, DeviceRowView* out);

// admm.hip: SLIM_Learn(algo = admm), the reference's dense ADMM solver (estimate.c:38-304) with
// rocBLAS for the panel solves, updates and products and HIP kernels for the rest.
slim_csr_t* learn_admm(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                       const float* rowval, const LearnOptions& opt, int32_t* status);

// eval.hip: HR / ARHR of top-N lists against a test matrix on the GPU (the host loop's figures,
// bit for bit), and the 1-vs-k protocol (every user ranks its own nnegs candidates).
struct EvalResult;
int32_t evaluate_device(int32_t nusers, int32_t nrcmds, const int32_t* lists, const int32_t* counts,
                        const slim_csr_t* tst, const int32_t* fmarker, int32_t fm_ncols,
                        EvalResult* out);
int32_t predict_1vsk_device(const slim_csr_t* W, const slim_csr_t* hist, int32_t nrcmds,
                            int32_t nnegs, const int32_t* negitems, int32_t* output,
                            float* scores);

}  // namespace slimamd
