/*
 * slim_gpu.h -- the rest of libslim.so's C ABI:
 *   (1) slim_csr_t, the concrete object behind slim_t handles;
 *   (2) the Py_* entry points the reference's Python wrapper resolves by name
 *       (they are declared in no reference header; bodies in
 *       /root/reference/src/libslim/pyapi.c);
 *   (3) SLIMGPU_* engine extensions: matrices resident in HBM, column-sharded
 *       solves for one-process-per-GPU jobs, and the counters bench.py reports.
 * Plain C types only: pointers and sizes, no torch/HIP types in any signature
 * (streams and device buffers are passed as void* / typed raw pointers).
 */
#ifndef SLIM_AMD_SLIM_GPU_H_
#define SLIM_AMD_SLIM_GPU_H_

#include "slim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 * (1) slim_csr_t: field-for-field the layout of GKlib's gk_csr_t (GKlib is an
 * empty submodule in the reference; layout recalled from upstream
 * gk_struct.h and consistent with every use in the reference's src/libslim sources), so that
 * callers compiled against GKlib can keep dereferencing model handles
 * (src/programs/slim_learn.c:83, slim_mselect.c:189-195, slim_predict.c:34).
 * Every array is malloc'd by the library or NULL.  A trained model carries both
 * views: col* (column iC = regressors of item iC; read by warm start,
 * src/libslim/estimate.c:455-458) and row* (read by prediction,
 * src/libslim/predict.c:46-57).
 * ------------------------------------------------------------------------- */
typedef struct slim_csr_t {
  int32_t nrows, ncols;
  ssize_t *rowptr, *colptr;
  int32_t *rowind, *colind;
  int32_t *rowids, *colids;
  int32_t *rlabels, *clabels;
  int32_t *rmap, *cmap;
  float *rowval, *colval;
  float *rnorms, *cnorms;
  float *rsums, *csums;
  float *rsizes, *csizes;
  float *rvols, *cvols;
  float *rwgts, *cwgts;
} slim_csr_t;

/* ---------------------------------------------------------------------------
 * (2) Python-facing entry points.  All return SLIM_OK / SLIM_ERROR*.
 * ------------------------------------------------------------------------- */

/* Deep-copy a CSR matrix into a library-owned handle (ncols = max id + 1).
 * Replaces Py_csr_wrapper, src/libslim/pyapi.c:22-38. */
int32_t Py_csr_wrapper(int32_t nrows, ssize_t *rowptr, int32_t *rowind,
                       float *rowval, slim_t **matrix_out);
/* Text CSR save/load of a handle's row view (pyapi.c:47-64). */
int32_t Py_csr_save(slim_t *mathandle, char *fname);
int32_t Py_csr_load(slim_t **mathandle, char *fname);
/* Release a handle (pyapi.c:72-76). */
int32_t Py_csr_free(slim_t *mathandle);
/* nnz of the row view, narrowed to int32 as in the reference (pyapi.c:85-89). */
int32_t Py_csr_stat(slim_t *mathandle, int32_t *nnz);
/* Copy the row view out as int32 indptr/indices + float data (pyapi.c:101-123). */
int32_t Py_csr_export(slim_t *mathandle, int32_t *indptr, int32_t *indices,
                      float *data);
/* SLIM_Learn on a wrapped training matrix, no warm start (pyapi.c:134-199). */
int32_t Py_SLIM_Learn(slim_t *trnhandle, int32_t *ioptions, double *doptions,
                      slim_t **model_out);
/* l1 x l2 grid with warm start from the previous cell and HR/ARHR evaluation;
 * the 8 outputs are the best-HR and best-ARHR cells (pyapi.c:214-412). */
int32_t Py_SLIM_Mselect(slim_t *trnhandle, slim_t *tsthandle, int32_t *ioptions,
                        double *doptions, double *arrayl1, double *arrayl2,
                        int32_t nl1, int32_t nl2, double *bestl1HR,
                        double *bestl2HR, double *bestHRHR, double *bestARHR,
                        double *bestl1AR, double *bestl2AR, double *bestHRAR,
                        double *bestARAR);
/* Single-profile top-N (pyapi.c:414-469). Return the list length or SLIM_ERROR. */
int32_t Py_SLIM_GetTopN(slim_t *model, int32_t nratings, int32_t *itemids,
                        float *ratings, int32_t nrcmds, int32_t *rids,
                        float *rscores, int32_t dbglvl);
int32_t Py_SLIM_GetTopN_1vsk(slim_t *model, int32_t nratings, int32_t *itemids,
                             float *ratings, int32_t nrcmds, int32_t *rids,
                             float *rscores, int32_t nnegs, int32_t *negitems,
                             int32_t dbglvl);
/* Top-N for every row of trnhandle; output[u*nrcmds + r] (pyapi.c:483-563). */
int32_t Py_SLIM_Predict(int32_t nrcmds, slim_t *slimhandle, slim_t *trnhandle,
                        int32_t *output, float *scores);
int32_t Py_SLIM_Predict_1vsk(int32_t nrcmds, int32_t nnegs, slim_t *slimhandle,
                             slim_t *trnhandle, int32_t *negitems,
                             int32_t *output, float *scores);

/* ---------------------------------------------------------------------------
 * (3) Engine extensions.
 * ------------------------------------------------------------------------- */

/* Extra option slots (reference leaves 11..39 unused, include/slim.h:215-230).
 * -1 selects the default, like every other slot. */
enum {
  SLIM_OPTION_GPU_COLBEGIN = 11, /* first item column solved by this call [0]  */
  SLIM_OPTION_GPU_COLEND = 12,   /* one past the last column [ncols]; columns
                                    outside the range come back empty          */
  SLIM_OPTION_GPU_SEED = 13,     /* seed of the visiting permutation [1]       */
  SLIM_OPTION_GPU_DEVICE = 14,   /* HIP device ordinal [current device]        */
  SLIM_OPTION_GPU_KERNEL = 15,   /* slimgpu_kernel_et [SLIMGPU_KERNEL_AUTO]    */
  SLIM_OPTION_GPU_CLUSTER = 16,  /* tile kernels: workgroups sharing one tile,
                                    1/2/4/8/16/32 [auto: by tiles per cluster] */
  SLIM_OPTION_GPU_HEAVYTILES = 17,   /* tile kernels: the N most expensive tiles are
                                        solved first by larger clusters [auto]; 0 = off */
  SLIM_OPTION_GPU_HEAVYCLUSTER = 18, /* size of those clusters, 2..32 [auto]          */
  SLIM_OPTION_GPU_NGPUS = 19,      /* SLIM_Learn / Py_SLIM_Learn / Py_SLIM_Mselect: number of
                                      GPUs of this node to shard the item columns over, one
                                      host thread + stream per device, R replicated [1]   */
  SLIM_OPTION_GPU_SHARDCOUNT = 20, /* SLIMGPU_Learn*: solve shard SHARDINDEX of SHARDCOUNT of  */
  SLIM_OPTION_GPU_SHARDINDEX = 21  /* the requested columns: granules (32 columns of the
                                      cost-ordered work list) INDEX, INDEX + COUNT, ...; a
                                      column's visiting order does not depend on COUNT [1, 0] */
};

typedef enum {
  SLIMGPU_KERNEL_AUTO = 0,
  SLIMGPU_KERNEL_WAVE_LDS = 1, /* one wavefront per item, work vectors in LDS  */
  SLIMGPU_KERNEL_WAVE_HBM = 2, /* one wavefront per item, work vectors in HBM  */
  SLIMGPU_KERNEL_TILE = 3,     /* one workgroup per 32 items, residuals
                                  interleaved r[user][32] in HBM (large matrices) */
  SLIMGPU_KERNEL_TILE16 = 4,   /* same with 16 items per workgroup             */
  SLIMGPU_KERNEL_GRAM = 5      /* item-space CD: one workgroup per item, g = a_i.r over the
                                  ITEMS kept on chip, an update reads one row of G = R^T R
                                  (built on the first such solve and kept with the handle: floats,
                                  and byte planes of ~1-2 bytes per entry when G is integer-valued).
                                  AUTO takes it when its byte model beats the residual kernel's
                                  (ncols^2 / nnz < 45) and the call's columns -- times the solves
                                  announced -- pay for G; no FSLIM form                         */
} slimgpu_kernel_et;

/* A training matrix staged in HBM: CSR as given + the column view (CSC, rows
 * ascending inside each column) + column norms, i.e. the device form of
 * CreateTrainingMatrix (src/libslim/setup.c:109-135). */
typedef struct slimgpu_matrix slimgpu_matrix_t;

/* Stage a host CSR (H2D once) and build the column view on the device. */
slimgpu_matrix_t *SLIMGPU_MatrixFromHost(int32_t nrows, const ssize_t *rowptr,
                                         const int32_t *rowind,
                                         const float *rowval,
                                         int32_t *ioptions, int32_t *r_status);
/* Adopt CSR arrays that already live in HBM (int64 rowptr[nrows+1], int32
 * rowind[nnz], float rowval[nnz] or NULL); they are borrowed and must outlive
 * the handle.  ncols <= 0: computed as max id + 1.  Work queued on other
 * streams that produces these arrays must be complete before the call. */
slimgpu_matrix_t *SLIMGPU_MatrixFromDevice(int32_t nrows, int32_t ncols,
                                           const int64_t *d_rowptr,
                                           const int32_t *d_rowind,
                                           const float *d_rowval,
                                           int32_t *ioptions, int32_t *r_status);
void SLIMGPU_MatrixFree(slimgpu_matrix_t **mat);
int32_t SLIMGPU_MatrixInfo(const slimgpu_matrix_t *mat, int32_t *nrows,
                           int32_t *ncols, int64_t *nnz);
/* Copy the device column view back (any pointer may be NULL): tests compare it
 * with the oracle's transpose. */
int32_t SLIMGPU_MatrixGetColumnView(const slimgpu_matrix_t *mat, int64_t *colptr,
                                    int32_t *colind, float *colval,
                                    float *cnorms);

/* Announce that the matrix is about to be solved nsolves times (a model-selection grid,
 * src/programs/slim_mselect.c:94-113): with nsolves >= 2 SLIMGPU_KERNEL_AUTO may build
 * G = R^T R for a grid whose single solves would not pay for it (SLIMGPU_KERNEL_GRAM: the
 * automatic choice multiplies the columns of a call by nsolves).  0 withdraws it. */
void SLIMGPU_MatrixExpectSolves(slimgpu_matrix_t *mat, int32_t nsolves);
/* The HIP device the handle's buffers live on (-1: null handle). */
int32_t SLIMGPU_MatrixDevice(const slimgpu_matrix_t *mat);

/* G = R^T R of item-space CD in row blocks, for drivers that run one process per GPU: every rank
 * forms the rows of its block of items (all ncols entries of each: BuildRows), the ranks exchange
 * their blocks into each other's buffers (View: device floats, `ld` per row -- e.g. one RCCL
 * broadcast per block), and Commit declares every row present (and forms the byte planes where G is
 * integer-valued).  Without these every rank builds all of G itself (what SLIMGPU_Learn does when
 * it needs G).  No counterpart in the reference (its CD keeps the residual, src/libslim/cd.c:86-153);
 * the sums are the a_i . a_j of estimate.c:412-421 for every pair of items.  SLIM_OK or an error
 * code (SLIMGPU_LastError). */
int32_t SLIMGPU_MatrixGramBuildRows(slimgpu_matrix_t *mat, int32_t row_begin, int32_t row_end);
int32_t SLIMGPU_MatrixGramView(slimgpu_matrix_t *mat, void **dptr, int64_t *ld, int32_t *nrows);
int32_t SLIMGPU_MatrixGramCommit(slimgpu_matrix_t *mat);

/* Scheduling cost proxy per item column (the Gram work G = sum over the column's
 * users of nnz(row u)); multi-GPU drivers balance their column blocks with it. */
int32_t SLIMGPU_MatrixColumnCost(const slimgpu_matrix_t *mat, int64_t *cost);

/* The estimate step of SLIM_Learn (EstimateModelCD + SaveModel,
 * src/libslim/estimate.c:328-593) on a staged matrix.  Same options, imodel and
 * result conventions as SLIM_Learn; can be called repeatedly on one matrix
 * (model-selection grids keep R resident).  A call that solves the same columns
 * as the previous call on this matrix (the next (l1, l2) pair of a grid,
 * src/programs/slim_mselect.c:99-113) reuses that call's screen sums a_i.y --
 * they depend on R only -- instead of recomputing them (kept in HBM with the
 * matrix, at most half of the free memory; SLIM_GPU_NO_GRAM=1 disables it);
 * the model is the same bit for bit.  Input rules (checked while R is staged,
 * SLIM_ERROR_INPUT otherwise): non-decreasing row offsets ending at nnz, item
 * ids inside [0, ncols), no (user, item) pair twice. */
slim_t *SLIMGPU_Learn(slimgpu_matrix_t *mat, int32_t *ioptions,
                      double *doptions, slim_t *imodel, int32_t *r_status);

/* Same for an explicit set of item columns (distinct ids, any order) instead of the range in
 * option slots 11/12: the unit of work of a shard whose columns are not contiguous, and of
 * the parity tests that solve one tile of a full-size matrix.  Columns not listed come back
 * empty.  The engine's work list is the given list stably sorted by descending cost. */
slim_t *SLIMGPU_LearnColumns(slimgpu_matrix_t *mat, int32_t ncolumns,
                             const int32_t *columns, int32_t *ioptions,
                             double *doptions, slim_t *imodel, int32_t *r_status);

/* Models resident in HBM.  A model-selection grid (src/programs/slim_mselect.c:94-113,
 * src/libslim/pyapi.c:283-300) learns its models one from the other over the same R: SLIM_Learn's
 * host model (SaveModel, estimate.c:570-593) is then 1.2 GB that cross PCIe twice per pair -- down as
 * the result, up again as the next pair's warm start.  SLIMGPU_LearnResident solves like
 * SLIMGPU_Learn but leaves the model (both views, formed on the device) in HBM and takes the warm
 * start from such a model without an upload; SLIMGPU_ModelFetch forms the host model of SLIM_Learn
 * (the same arrays, bit for bit; SLIM_FreeModel releases it) when the caller wants it, and
 * SLIMGPU_ModelFetchBegin starts that copy on its own stream and host thread so that it runs beside
 * the next solve (ModelFetch then joins it).  One device per model (ngpus = 1).
 * A warm start from a resident model that keeps l1 (a grid step along l2: the same active sets) on the
 * packed item-space kernel does not fold the model into g again: it starts from the g the previous
 * solve left (the same quantity, cd.c:108-110, at fp32-rounding distance; SLIM_GPU_NO_CARRY=1 folds). */
typedef struct slimgpu_model slimgpu_model_t;
slimgpu_model_t *SLIMGPU_LearnResident(slimgpu_matrix_t *mat, int32_t *ioptions, double *doptions,
                                       const slimgpu_model_t *warm, int32_t *r_status);
int64_t SLIMGPU_ModelNnz(const slimgpu_model_t *model);    /* -1: null */
int32_t SLIMGPU_ModelFetchBegin(slimgpu_model_t *model);
slim_t *SLIMGPU_ModelFetch(slimgpu_model_t *model, int32_t *r_status);
void SLIMGPU_ModelFree(slimgpu_model_t **model);
/* SLIMGPU_Predict through a resident model: its row view is read where it lies (lists and scores
 * bit-identical to Py_SLIM_Predict on the fetched model).  1 <= nrcmds <= 128. */
int32_t SLIMGPU_ModelPredict(int32_t nrcmds, const slimgpu_model_t *model, slim_t *trnhandle,
                             int32_t *output, float *scores);

/* Py_SLIM_Predict on the GPU (one wavefront per user; lists and scores are bit-identical
 * to the host scorer, ties included).  1 <= nrcmds <= 128.  Fails without a device. */
int32_t SLIMGPU_Predict(int32_t nrcmds, slim_t *slimhandle, slim_t *trnhandle,
                        int32_t *output, float *scores);

/* Py_SLIM_Predict_1vsk on the GPU (src/libslim/predict.c:77-133, pyapi.c:483-528): user u
 * ranks negitems[u*nnegs .. +nnegs); scores and tie order are the host scorer's.  Model rows
 * must be ascending by item id (the engine's models are); 1 <= nnegs <= 1024. */
int32_t SLIMGPU_Predict1vsK(int32_t nrcmds, int32_t nnegs, slim_t *slimhandle,
                            slim_t *trnhandle, int32_t *negitems, int32_t *output,
                            float *scores);

/* Leave-k-out evaluation of top-N lists on the GPU (src/programs/slim_predict.c:181-236,
 * src/libslim/pyapi.c:309-366): lists[u*nrcmds .. +counts[u]) against row u of tsthandle;
 * fmarker = SLIM_DetermineHeadAndTail over fm_ncols items.  metrics = {HR, HR_head,
 * HR_tail, ARHR}, nvalid = {users with a test item, ... with a head item, ... with a tail
 * item}: the figures of the reference's host loop (float accumulators, user order). */
int32_t SLIMGPU_Evaluate(int32_t nusers, int32_t nrcmds, const int32_t *lists,
                         const int32_t *counts, slim_t *tsthandle,
                         const int32_t *fmarker, int32_t fm_ncols, double *metrics,
                         int32_t *nvalid);

/* Counters of the most recent solve on this thread. */
typedef struct slimgpu_stats_t {
  int32_t ncols_solved;
  int32_t kernel;          /* slimgpu_kernel_et actually used                 */
  int32_t nwaves;          /* persistent wavefronts launched                  */
  int32_t lds_bytes;       /* dynamic LDS per wavefront (0 for the HBM kernel) */
  double setup_ms;         /* host->HBM staging + column view (FromHost only) */
  double kernel_ms;        /* solver kernel, HIP events on the engine stream  */
  double gather_ms;        /* D2H of W + host assembly of the model           */
  double total_ms;         /* wall time of the call                           */
  int64_t G, D, U, nnzW;   /* sums of the per-column traffic terms below      */
  int64_t sweeps;          /* total CD sweeps                                 */
  int64_t visits;          /* total coordinate visits                         */
  double alg_bytes;        /* 8G + 12D + 4U + 8 nnzW (4G + 8D + 4U + 8 nnzW
                              for a binary matrix): SURVEY.md 8(d)            */
  double error, objval;    /* sum of 1/2||r||^2 and of the objective          */
  double gram_build_ms;    /* SLIMGPU_KERNEL_GRAM: time spent building G = R^T R in
                              this call (0 when it was there already)          */
  int64_t gram_rows;       /* SLIMGPU_KERNEL_GRAM: rows of G read (one per update and per
                              folded warm-start coefficient)                   */
  double gram_bytes;       /* bytes of G those rows streamed -- gram_rows x 4 ncols for the
                              float kernels, the packed rows' bytes (counted on the device)
                              for the byte-plane kernel: the item-space byte model (it does
                              not move what SURVEY.md 8(d)'s alg_bytes prices)  */
  /* gram_build_ms split (round 6; appended: older callers read a prefix): the allocation of
     G (a first 40 GB hipMalloc costs seconds on some boxes and nothing on others), the
     sums -- the whole nested solve that forms them, and its kernel alone -- and the byte
     planes                                                                            */
  double gram_alloc_ms, gram_sums_ms, gram_sums_kernel_ms, gram_pack_ms;
} slimgpu_stats_t;
int32_t SLIMGPU_LastStats(slimgpu_stats_t *out);

/* Per-column counters of the most recent solve (arrays of ncols entries; any
 * may be NULL): active-set size, sweeps (wspace->niters), convergence flag,
 * G = sum over the column's users of nnz(row u), D = sum over sweeps and active
 * columns of nnz(col i), U = same restricted to visits that changed x. */
int32_t SLIMGPU_LastColumnStats(int32_t ncols, int32_t *nacols, int32_t *sweeps,
                                int32_t *conv, int64_t *G, int64_t *D,
                                int64_t *U);

int32_t SLIMGPU_DeviceCount(void);
/* Human-readable description of the last failure on this thread ("" if none). */
const char *SLIMGPU_LastError(void);

#ifdef __cplusplus
}
#endif
#endif /* SLIM_AMD_SLIM_GPU_H_ */
