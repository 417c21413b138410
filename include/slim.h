/*
 * slim.h -- public C API of the MI355X-native SLIM engine (libslim.so).
 *
 * This header is source-compatible with the reference's include/slim.h: same
 * function names, argument order/meaning, enum names and values, so programs
 * written against KarypisLab/SLIM recompile unchanged.  Each declaration cites
 * the reference interface it replaces (paths relative to /root/reference).
 * The implementation behind it is new: SLIM_Learn(algo=cd) runs the
 * coordinate-descent solver as HIP kernels on gfx950 (slim_amd/csrc).
 *
 * Engine-only extensions (device-resident matrices, column sharding, counters)
 * are in slim_gpu.h; nothing here depends on them.
 */
#ifndef SLIM_AMD_SLIM_H_
#define SLIM_AMD_SLIM_H_

#include <inttypes.h>
#include <stdint.h>
#include <sys/types.h> /* ssize_t */

/* Opaque handle for models and matrices (reference: include/slim.h:48).  The
 * object behind it keeps the field layout of GKlib's gk_csr_t, which the
 * reference's CLI programs dereference (src/programs/slim_learn.c:83); see
 * slim_csr_t in slim_gpu.h. */
typedef void slim_t;

#define SLIM_VERSION "2.0" /* include/slim.h:53 */
#define SLIM_NOPTIONS 40   /* include/slim.h:56: length of ioptions[]/doptions[] */

#ifdef __cplusplus
extern "C" {
#endif

/* Fill all SLIM_NOPTIONS slots with -1 ("use the default").  Returns SLIM_OK.
 * Replaces include/slim.h:79-89 (src/libslim/api.c:149-165). */
int32_t SLIM_iSetDefaults(int32_t *options);
int32_t SLIM_dSetDefaults(double *options);

/* Learn the item-item model W from a user x item CSR matrix.
 *   nrows            number of users
 *   rowptr[nrows+1]  row offsets (ssize_t, as in the reference)
 *   rowind[nnz]      item ids, used as given; the model is ncols x ncols with
 *                    ncols = max id + 1
 *   rowval[nnz]      ratings, or NULL for a binary matrix
 *   ioptions/doptions  option arrays indexed by slim_options_et; -1 = default
 *                    (nthreads 1, nnbrs 0, algo cd, maxniters 10000, l1r 1,
 *                    l2r 1, optTol 1e-7); NULL = all defaults
 *   imodel           previous model used as warm start, or NULL (borrowed)
 *   r_status         SLIM_OK on success; on failure the return value is NULL
 *                    and *r_status is SLIM_ERROR_INPUT / _MEMORY / SLIM_ERROR
 *                    (the reference exits the process instead)
 * Inputs are borrowed for the duration of the call.  The result is owned by
 * the library; release it with SLIM_FreeModel.
 * Replaces include/slim.h:107-110 (src/libslim/api.c:33-96). */
slim_t *SLIM_Learn(int32_t nrows, ssize_t *rowptr, int32_t *rowind,
                   float *rowval, int32_t *ioptions, double *doptions,
                   slim_t *imodel, int32_t *r_status);

/* Top-N recommendation for one user profile: scores every item reachable from
 * the history through W, drops the history itself, returns the nrcmds best in
 * rids/rscores (descending score).  Returns the list length (may be < nrcmds)
 * or SLIM_ERROR.  Replaces include/slim.h:125-127 (src/libslim/api.c:111-141,
 * src/libslim/predict.c:15-71). */
int32_t SLIM_GetTopN(slim_t *model, int32_t nratings, int32_t *itemids,
                     float *ratings, int32_t *ioptions, int32_t nrcmds,
                     int32_t *rids, float *rscores);

/* Binary model file (row view): int32 nrows, int32 ncols, ssize_t
 * rowptr[nrows+1], int32 rowind[nnz], float rowval[nnz].
 * Replaces include/slim.h:136-146 (src/libslim/api.c:174-194). */
int32_t SLIM_WriteModel(slim_t *model, char *filename);
slim_t *SLIM_ReadModel(char *filename);

/* Release a model and NULL the caller's pointer.
 * Replaces include/slim.h:154 (src/libslim/api.c:204). */
void SLIM_FreeModel(slim_t **model);

/* Popularity split used by the evaluation programs: returns a malloc'd array of
 * ncols markers, 0 = head (most popular items covering half of the ratings),
 * 1 = tail.  The caller frees it with free().
 * Replaces include/slim.h:166-167 (src/libslim/api.c:215-245). */
int32_t *SLIM_DetermineHeadAndTail(int32_t nrows, int32_t ncols,
                                   ssize_t *rowptr, int32_t *rowind);

#ifdef __cplusplus
}
#endif

/* ---- enums: names and values of include/slim.h:177-239 ------------------- */

typedef enum {
  SLIM_OK = 1,
  SLIM_ERROR_INPUT = -2,
  SLIM_ERROR_MEMORY = -3,
  SLIM_ERROR = -4
} slim_rstatus_et;

typedef enum {
  SLIM_MTYPE_SLIM = 0,
  SLIM_MTYPE_FSLIM = 1,
  SLIM_MTYPE_OSLIM = 2,
  SLIM_MTYPE_OFSLIM = 3
} slim_mtype_et;

typedef enum {
  SLIM_SIMTYPE_COS = 0,
  SLIM_SIMTYPE_JAC = 1,
  SLIM_SIMTYPE_DOTP = 2
} slim_simtype_et;

typedef enum { SLIM_ALGO_ADMM = 0, SLIM_ALGO_CD = 1 } slim_algo_et;

typedef enum {
  SLIM_OPTION_DBGLVL = 0,
  SLIM_OPTION_NNBRS = 1,
  SLIM_OPTION_SIMTYPE = 2,
  SLIM_OPTION_NTHREADS = 3,
  SLIM_OPTION_MAXNITERS = 4,
  SLIM_OPTION_ALGO = 5,
  SLIM_OPTION_ORDERED = 6,
  SLIM_OPTION_L1R = 7,
  SLIM_OPTION_L2R = 8,
  SLIM_OPTION_OPTTOL = 9,
  SLIM_OPTION_NRCMDS = 10
  /* slots 11..39: unused by the reference; see slim_gpu.h */
} slim_options_et;

typedef enum {
  SLIM_DBG_INFO = 1,
  SLIM_DBG_TIME = 2,
  SLIM_DBG_PROGRESS = 4,
  SLIM_DBG_PROGRESS2 = 16,
  SLIM_DBG_MEMORY = 2048
} slim_dbglvl_et;

/* text labels, as the reference exposes them (include/slim.h:193,203,212) */
static const char slim_mtypenames[][10] = {"SLIM", "FSLIM", "OSLIM", "OFSLIM", ""};
static const char slim_simtypenames[][10] = {"cos", "jac", "dotp", ""};
static const char slim_algonames[][10] = {"admm", "cd", ""};

#endif /* SLIM_AMD_SLIM_H_ */
