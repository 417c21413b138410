// host_csr.hpp -- host-side slim_csr_t utilities of libslim.so (C++17).
//
// Everything here is O(nnz) bookkeeping around the solver: building and
// releasing the handle objects, the row<->column index of a learned model, the
// top-N scorer and the HR/ARHR evaluation that consume W, and the model file
// formats.  None of it is on the training hot path (that is cd_wave.hpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/slim_gpu.h"

namespace slimamd {

constexpr double kEpsilon = 1e-7;  // reference src/libslim/def.h:14

// Thread-local error text behind SLIMGPU_LastError().
void set_error(const std::string& msg);
const char* last_error();

// Allocate an all-NULL handle / release a handle and every array it owns.
slim_csr_t* csr_new();
void csr_free(slim_csr_t* m);

// max id + 1 (reference src/libslim/setup.c:117).
int32_t max_index_plus_one(int64_t nnz, const int32_t* ind);

// Deep copy of a caller's CSR into a new handle (row view only); val may be NULL.
slim_csr_t* csr_from_rows(int32_t nrows, const ssize_t* ptr, const int32_t* ind,
                          const float* val);

// Build the missing view of `m` from the other one by a counting-sort transpose
// (ascending ids inside each output row/column).  what: 0 = build columns from
// rows, 1 = build rows from columns.  Replaces the GKlib call sites
// src/libslim/setup.c:128 and src/libslim/estimate.c:591.
void csr_build_index(slim_csr_t* m, int what);

// Assemble a trained model from its column view (takes ownership of the three
// malloc'd arrays) and add the row view (reference estimate.c:570-593).
// row_view = false: the piece of a sharded solve, columns only (multi_gpu.cpp merges them).
slim_csr_t* model_from_columns(int32_t n, ssize_t* colptr, int32_t* colind,
                               float* colval, bool row_view = true);

// ---- consumers of W (host; reference predict.c, api.c:215-245) -------------

// Scratch reused across users: marker (preset -1) and candidate list.
struct TopNScratch {
  std::vector<int32_t> marker;
  std::vector<float> key;
  std::vector<int32_t> val;
  explicit TopNScratch(int32_t ncols) : marker(ncols, -1), key(ncols), val(ncols) {}
};

// predict.c:15-71.  Returns the list length.
int32_t top_n(const slim_csr_t* W, int32_t nratings, const int32_t* itemids,
              const float* ratings, int32_t nrcmds, int32_t* rids, float* rscores,
              TopNScratch& ws);
// predict.c:77-133: rank a fixed negative-candidate list.
int32_t top_n_1vsk(const slim_csr_t* W, int32_t nratings, const int32_t* itemids,
                   const float* ratings, int32_t nrcmds, int32_t* rids,
                   float* rscores, int32_t nnegs, const int32_t* negitems);

// api.c:215-245.  malloc'd marker array (0 head / 1 tail).
int32_t* head_tail_split(int32_t nrows, int32_t ncols, const ssize_t* rowptr,
                         const int32_t* rowind);

struct EvalResult {
  float hr = 0, hr_head = 0, hr_tail = 0, arhr = 0;
  int32_t nvalid = 0, nvalid_head = 0, nvalid_tail = 0;
};
// pyapi.c:309-366: HR/ARHR of `model` for users with a non-empty test row.
// lists/counts (optional): top-N ids [nusers][nrcmds] and list lengths computed elsewhere
// (the GPU scorer); when null the host scorer is used.
EvalResult evaluate(const slim_csr_t* model, const slim_csr_t* trn,
                    const slim_csr_t* tst, int32_t nrcmds, const int32_t* fmarker,
                    int32_t fm_ncols, const int32_t* lists = nullptr,
                    const int32_t* counts = nullptr);

// ---- files ------------------------------------------------------------------
bool write_binrow(const slim_csr_t* m, const char* path);   // api.c:174-177
slim_csr_t* read_binrow(const char* path);                  // api.c:187-194
bool write_text_csr(const slim_csr_t* m, const char* path); // pyapi.c:47-51
slim_csr_t* read_text_csr(const char* path);                // pyapi.c:59-64

}  // namespace slimamd
