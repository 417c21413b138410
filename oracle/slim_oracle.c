/*
 * slim_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's
 * SLIM coordinate-descent training path (and of the top-N / HR evaluation that
 * defines the parity metric).  Nothing in the shipped product (slim_amd/,
 * libslim.so) may include, link, import or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Provenance.  Own code, written from the behaviour of the reference at
 * /root/reference (KarypisLab/SLIM); each function cites the reference
 * file:line it restates.  The reference itself is UNBUILDABLE in this image:
 * src/libslim/slimlib.h:14 includes <GKlib.h> and lib/GKlib is an empty,
 * un-vendored submodule (.gitmodules:1-3).  GKlib (github.com/KarypisLab/GKlib,
 * version unpinned by the reference) owns three pieces of arithmetic on the
 * path; they are restated here from their published algorithm and from the
 * reference's call sites:
 *   gk_csr_CreateIndex   counting-sort transpose, rows ascending inside each
 *                        column           (call sites setup.c:128, estimate.c:591)
 *   gk_csr_ComputeNorms  cnorm = (float)sqrt(fp32 sum of val^2) (setup.c:130)
 *   gk_fkvsortd          descending sort by float key; tie order undefined
 *                        upstream (unstable quicksort) -- stable here
 *                                                         (predict.c:60,123)
 *
 * Pinning (see tests/test_oracle_pins.py): the reference ships no tests; its
 * only recorded outputs are python-package/UserGuide.ipynb:275-277 (Automotive
 * 9x9 model selection: best-HR and best-AR lines), which this oracle
 * reproduces exactly, plus the SURVEY.md 8(c) probe values.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_EPS 1e-7 /* def.h:14 EPSILON */

/* float key + index pair: GKlib gk_fkv_t {float key; ssize_t val;} */
typedef struct {
  float key;
  int64_t val;
} fkv_t;

/* ------------------------------------------------------------------------ */
/* visiting-order generators                                                 */
/* ------------------------------------------------------------------------ */

/* cd.c:76-86 ShuffleList: "swap element i with a random index in [0,n)",
 * libc rand(), never seeded by the reference.                               */
static void shuffle_glibc(fkv_t *list, int32_t n) {
  for (int32_t i = 0; i < n; i++) {
    fkv_t t = list[i];
    int32_t j = rand() % n;
    list[i] = list[j];
    list[j] = t;
  }
}

/* thread-local xorshift variant of the same swap shuffle (the
 * "thread-local PRNG" CPU-baseline mode; glibc rand() serialises threads).   */
static inline uint32_t xs32(uint32_t *s) {
  uint32_t x = *s;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return *s = x;
}
static void shuffle_local(fkv_t *list, int32_t n, uint32_t *state) {
  for (int32_t i = 0; i < n; i++) {
    fkv_t t = list[i];
    int32_t j = (int32_t)(xs32(state) % (uint32_t)n);
    list[i] = list[j];
    list[j] = t;
  }
}

/* Stateless keyed permutation of [0,n): the visiting order the HIP engine
 * uses (slim_amd/csrc/cd_perm.h is the device twin; both are integer-exact,
 * so engine and oracle can walk identical orders).  A bijection on b =
 * ceil(log2 n) bits (odd multiply, add, xorshift-right are each invertible
 * mod 2^b) followed by cycle walking into [0,n).                             */
uint32_t oracle_perm_key(uint32_t seed, uint32_t item, uint32_t sweep) {
  uint32_t h = seed * 0x9E3779B1u + 0x7F4A7C15u;
  h ^= item + 0x85EBCA6Bu + (h << 6) + (h >> 2);
  h *= 0xC2B2AE35u;
  h ^= h >> 15;
  h ^= sweep * 0x27D4EB2Fu + 0x165667B1u + (h << 6) + (h >> 2);
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
uint32_t oracle_perm_index(uint32_t p, uint32_t n, uint32_t key) {
  if (n <= 1) return 0;
  uint32_t b = 32u - (uint32_t)__builtin_clz(n - 1); /* ceil(log2 n), n>=2 */
  uint32_t mask = (b >= 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  uint32_t s1 = (b + 1) / 2, s2 = (b + 2) / 3;
  if (s2 == 0) s2 = 1;
  uint32_t a1 = (key | 1u), c1 = key >> 7;
  uint32_t a2 = ((key * 0x9E3779B1u) >> 3) | 1u, c2 = (key * 0x85EBCA6Bu) >> 11;
  uint32_t a3 = ((key * 0xC2B2AE35u) >> 5) | 1u, c3 = key >> 17;
  uint32_t x = p;
  do {
    x = (x * a1 + c1) & mask;
    x ^= x >> s1;
    x = (x * a2 + c2) & mask;
    x ^= x >> s2;
    x = (x * a3 + c3) & mask;
    x ^= x >> s1;
  } while (x >= n);
  return x;
}

/* ------------------------------------------------------------------------ */
/* setup: setup.c:109-135 CreateTrainingMatrix                                */
/* ------------------------------------------------------------------------ */

/* setup.c:117  ncols = gk_i32max(rowind)+1 */
int32_t oracle_ncols(int64_t nnz, const int32_t *rowind) {
  int32_t m = -1;
  for (int64_t i = 0; i < nnz; i++)
    if (rowind[i] > m) m = rowind[i];
  return m + 1;
}

/* gk_csr_CreateIndex(GK_CSR_COL) [GKlib; call site setup.c:128]: counting-sort
 * transpose.  Walking rows in ascending order makes every column's row ids
 * ascending, which slim_csr_SortIndices (setup.c:19-94) then leaves untouched.
 * val may be NULL (binary matrix, setup.c:122-126).                          */
void oracle_transpose(int32_t nrows, int32_t ncols, const int64_t *ptr,
                      const int32_t *ind, const float *val, int64_t *tptr,
                      int32_t *tind, float *tval) {
  memset(tptr, 0, sizeof(int64_t) * ((size_t)ncols + 1));
  for (int64_t j = 0; j < ptr[nrows]; j++) tptr[ind[j] + 1]++;
  for (int32_t c = 0; c < ncols; c++) tptr[c + 1] += tptr[c];
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * ((size_t)ncols + 1));
  memcpy(cur, tptr, sizeof(int64_t) * ((size_t)ncols + 1));
  for (int32_t r = 0; r < nrows; r++)
    for (int64_t j = ptr[r]; j < ptr[r + 1]; j++) {
      int64_t d = cur[ind[j]]++;
      tind[d] = r;
      if (val && tval) tval[d] = val[j];
    }
  free(cur);
}

/* gk_csr_ComputeNorms(GK_CSR_COL) [GKlib; call site setup.c:130]:
 * cnorms[i] = sqrt(sum val^2) with a float accumulator; sqrt(nnz) if binary. */
void oracle_col_norms(int32_t ncols, const int64_t *colptr, const float *colval,
                      float *cnorms) {
  for (int32_t c = 0; c < ncols; c++) {
    float s = 0.0f;
    if (colval)
      for (int64_t j = colptr[c]; j < colptr[c + 1]; j++)
        s += colval[j] * colval[j];
    else
      s = (float)(colptr[c + 1] - colptr[c]);
    cnorms[c] = (float)sqrt((double)s);
  }
}

/* ------------------------------------------------------------------------ */
/* cd.c:24-65 sparse axpy / dot on the column view                            */
/* ------------------------------------------------------------------------ */
typedef struct {
  const int64_t *colptr;
  const int32_t *colind;
  const float *colval; /* NULL => binary */
  const float *cnorms;
} cview_t;

/* cd.c:24-38 AddSpVec: skipped entirely when |xi| <= EPSILON */
static inline int64_t add_spvec(const cview_t *A, int32_t i, double xi,
                                double *yhat) {
  if (xi > ORACLE_EPS || xi < -ORACLE_EPS) {
    if (A->colval)
      for (int64_t j = A->colptr[i]; j < A->colptr[i + 1]; j++)
        yhat[A->colind[j]] += xi * A->colval[j];
    else
      for (int64_t j = A->colptr[i]; j < A->colptr[i + 1]; j++)
        yhat[A->colind[j]] += xi;
    return A->colptr[i + 1] - A->colptr[i];
  }
  return 0;
}

/* cd.c:51-65 SpVecInnerProduct */
static inline double spvec_dot(const cview_t *A, int32_t i, const double *yhat) {
  double res = 0.0;
  if (A->colval)
    for (int64_t j = A->colptr[i]; j < A->colptr[i + 1]; j++)
      res += A->colval[j] * yhat[A->colind[j]];
  else
    for (int64_t j = A->colptr[i]; j < A->colptr[i + 1]; j++)
      res += yhat[A->colind[j]];
  return res;
}

/* ------------------------------------------------------------------------ */
/* configuration                                                             */
/* ------------------------------------------------------------------------ */
enum { ORDER_GLIBC = 0, ORDER_PERM = 1, ORDER_LOCAL = 2, ORDER_NONE = 3 };
enum { ATY_FULLSCAN = 0, ATY_GRAM = 1 };

typedef struct {
  double l1r, l2r, optTol; /* api.c:50-52 */
  int32_t maxniters;       /* api.c:48 */
  int32_t nthreads;        /* api.c:42 */
  int32_t order;           /* ORDER_* */
  uint32_t seed;           /* ORDER_PERM / ORDER_LOCAL */
  int32_t aty;             /* ATY_* : estimate.c:412-421 vs Gram-column */
  int32_t fp32;            /* 0: reference arithmetic (fp64 x/yhat, 3-pass)
                              1: engine-style arithmetic (fp32 residual, fused) */
  int32_t nnbrs;           /* > 0: FSLIM, api.c:43,55-56 */
  int32_t simtype;         /* 0 cos, 1 jac, 2 dotp (slim.h:196-200) */
  int32_t chunk;           /* omp dynamic chunk; 0 = 32, the reference's (estimate.c:402).
                              Timing a SAMPLE of columns wants 1 (one column per thread) */
  int32_t tile_first;      /* oracle_learn_cd_tile: walk only tiles [tile_first,           */
  int32_t tile_count;      /*   tile_first + tile_count) of the work list (0 = all)         */
} oracle_cfg_t;

/* wall time of the estimate phase (the reference's LearnTmr, api.c:68-85) of the last
 * oracle_learn_cd / oracle_learn_cd_tile call, without the setup (transpose, norms) */
static double g_learn_seconds = 0.0;
double oracle_learn_seconds(void) { return g_learn_seconds; }
/* Timing aid (bench.py's bounded CPU sample): a wall-clock budget for the estimate phase of the
 * NEXT oracle_learn_cd call.  Past it no new sweep starts: a column that was cut off reports
 * conv = -1 and the D it had reached, a column that never started conv = -2 -- the caller turns
 * partial columns into fractions of a column.  0 (the default) = no budget; results of a
 * budgeted call are not models and are never compared with anything. */
static double g_time_budget = 0.0, g_deadline = 0.0;
void oracle_set_time_budget(double seconds) { g_time_budget = seconds > 0.0 ? seconds : 0.0; }
static double now_seconds(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

static void fkv_sortd_stable(fkv_t *a, int64_t n);

/* CreateTrainingMatrix (setup.c:109-135): column view + norms of the caller's CSR.  With the
 * cache switched on (oracle_cache_setup(1): bench.py's CPU leg, which calls the oracle several
 * times on one 1e9-nnz matrix) the view of the last matrix is kept and reused while the caller
 * passes the same rowind array; the caller owns that guarantee and clears the cache afterwards. */
typedef struct {
  int64_t *colptr;
  int32_t *colind;
  float *colval;
  float *cnorms;
} colview_t;
static int g_cache_on = 0;
static struct {
  const int32_t *rowind;
  const float *rowval;
  int64_t nnz;
  int32_t nrows, ncols;
  colview_t v;
} g_cache = {0};

static void colview_free(colview_t *v) {
  free(v->colptr); free(v->colind); free(v->colval); free(v->cnorms);
  memset(v, 0, sizeof(*v));
}
void oracle_cache_setup(int32_t on) {
  g_cache_on = on;
  if (!on && g_cache.v.colptr) {
    colview_free(&g_cache.v);
    memset(&g_cache, 0, sizeof(g_cache));
  }
}
static colview_t colview_get(int32_t nrows, int32_t ncols, const int64_t *rowptr,
                             const int32_t *rowind, const float *rowval) {
  const int64_t nnz = rowptr[nrows];
  if (g_cache_on && g_cache.v.colptr && g_cache.rowind == rowind && g_cache.rowval == rowval &&
      g_cache.nnz == nnz && g_cache.nrows == nrows && g_cache.ncols == ncols)
    return g_cache.v;
  colview_t v;
  v.colptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)ncols + 1));
  v.colind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  v.colval = rowval ? (float *)malloc(sizeof(float) * (size_t)(nnz ? nnz : 1)) : NULL;
  v.cnorms = (float *)malloc(sizeof(float) * (size_t)ncols);
  oracle_transpose(nrows, ncols, rowptr, rowind, rowval, v.colptr, v.colind, v.colval);
  oracle_col_norms(ncols, v.colptr, v.colval, v.cnorms);
  if (g_cache_on) {
    if (g_cache.v.colptr) colview_free(&g_cache.v);
    g_cache.rowind = rowind; g_cache.rowval = rowval; g_cache.nnz = nnz;
    g_cache.nrows = nrows; g_cache.ncols = ncols; g_cache.v = v;
  }
  return v;
}
static void colview_put(colview_t *v) {
  if (g_cache_on && g_cache.v.colptr == v->colptr) return; /* kept */
  colview_free(v);
}

/* neighbors.c:16-125 FindColumnNeighbors: candidates = items co-rated with iC,
 * similarity accumulated in FLOAT over (users of iC ascending, row order)
 * (neighbors.c:46-60), cos: / cnorm[k] (:82-83), jac: / (cnorm[k] + cnorm[iC] - key)
 * (:107-109, norms not squared, as in the reference), dotp: as is; the nnbrs
 * best are kept.  gk_dfkvkselect/gk_fkvsortd leave ties undefined upstream:
 * here (similarity descending, item id ascending).  marker/cand: ncols scratch. */
static int32_t find_neighbors(const oracle_cfg_t *cfg, int32_t nrows,
                              const int64_t *rowptr, const int32_t *rowind,
                              const float *rowval, const cview_t *A, int32_t iC,
                              int32_t *marker, fkv_t *cand) {
  (void)nrows;
  if (A->colptr[iC] == A->colptr[iC + 1]) return 0; /* neighbors.c:31-32 */
  int32_t ncand = 0;
  for (int64_t ii = A->colptr[iC]; ii < A->colptr[iC + 1]; ii++) {
    const int32_t u = A->colind[ii];
    const float cval = A->colval ? A->colval[ii] : 1.0f;
    for (int64_t j = rowptr[u]; j < rowptr[u + 1]; j++) {
      const int32_t k = rowind[j];
      if (k == iC) continue;
      if (marker[k] == -1) {
        cand[ncand].val = k;
        cand[ncand].key = 0;
        marker[k] = ncand++;
      }
      if (rowval)
        cand[marker[k]].key += rowval[j] * cval;
      else
        cand[marker[k]].key += cval;
    }
  }
  for (int32_t i = 0; i < ncand; i++) {
    const int64_t k = cand[i].val;
    if (cfg->simtype == 0)
      cand[i].key = cand[i].key / A->cnorms[k];
    else if (cfg->simtype == 1)
      cand[i].key = cand[i].key / (A->cnorms[k] + A->cnorms[iC] - cand[i].key);
    marker[k] = -1;
  }
  /* top nnbrs: similarity descending, then item id ascending */
  for (int32_t a = 1; a < ncand; a++) { /* order by id first (insertion sort is fine: */
    fkv_t t = cand[a];                   /* candidates arrive nearly sorted per row)   */
    int32_t b = a;
    while (b > 0 && cand[b - 1].val > t.val) {
      cand[b] = cand[b - 1];
      b--;
    }
    cand[b] = t;
  }
  fkv_sortd_stable(cand, ncand);
  return ncand < cfg->nnbrs ? ncand : cfg->nnbrs;
}

/* per-column counters (SURVEY 8(d) algorithmic-bytes terms) */
typedef struct {
  int32_t nacols; /* active-set size */
  int32_t sweeps; /* wspace->niters, cd.c:140 */
  int32_t conv;   /* rstatus, cd.c:136 */
  int32_t nnzw;   /* kept entries */
  int64_t G;      /* sum over users of col of nnz(row u) */
  int64_t D;      /* sum over sweeps, active cols of nnz(col i) */
  int64_t U;      /* sum over visits with changed coefficient of nnz(col i) */
  double err;     /* 1/2 ||y - yhat||^2 */
  double obj;     /* err + l2/2 ||x||^2 + l1 ||x||_1 */
} oracle_colstat_t;

/* cd.c:101-142 CoordinateDescent, reference arithmetic.
 * x, y, yhat dense doubles; act[] = (float aTy, column id).                  */
static int32_t cd_reference(const cview_t *A, const oracle_cfg_t *cfg,
                            fkv_t *act, int32_t na, int32_t maxniters,
                            double *x, double *yhat, int32_t item,
                            uint32_t *lstate, fkv_t *tmp, int32_t *r_niters,
                            int64_t *D, int64_t *U) {
  int32_t t, rstatus = 0;
  for (int32_t i = 0; i < na; i++) /* cd.c:108-110 */
    add_spvec(A, (int32_t)act[i].val, x[act[i].val], yhat);

  for (t = 0; t < maxniters; t++) {
    double dltx = 0.0;
    const fkv_t *visit = act;
    if (g_deadline > 0.0 && now_seconds() > g_deadline) { /* oracle_set_time_budget */
      rstatus = -1;
      break;
    }
    if (cfg->order == ORDER_GLIBC)
      shuffle_glibc(act, na); /* cd.c:115 */
    else if (cfg->order == ORDER_LOCAL)
      shuffle_local(act, na, lstate);
    else if (cfg->order == ORDER_PERM) {
      uint32_t key = oracle_perm_key(cfg->seed, (uint32_t)item, (uint32_t)t);
      for (int32_t p = 0; p < na; p++)
        tmp[p] = act[oracle_perm_index((uint32_t)p, (uint32_t)na, key)];
      visit = tmp;
    }
    for (int32_t i = 0; i < na; i++) { /* cd.c:116-133 */
      int32_t iI = (int32_t)visit[i].val;
      double aTy = visit[i].key;
      double aTa = A->cnorms[iI];
      double xi = x[iI];
      int64_t len = A->colptr[iI + 1] - A->colptr[iI];
      int64_t touched = add_spvec(A, iI, -xi, yhat);
      double ip = spvec_dot(A, iI, yhat);
      double num = aTy - ip;
      double newxi =
          num > cfg->l1r ? (num - cfg->l1r) / ((aTa * aTa) + cfg->l2r) : 0.0;
      touched += add_spvec(A, iI, newxi, yhat);
      x[iI] = newxi;
      dltx += (newxi - xi) * (newxi - xi);
      *D += len;
      if (touched) *U += len;
    }
    if (dltx < cfg->optTol) { /* cd.c:135-138 */
      rstatus = 1;
      break;
    }
  }
  *r_niters = t + 1; /* cd.c:140 (also when maxniters == 0 or loop exhausted) */
  return rstatus;
}

/* Engine-style arithmetic on the same algorithm: fp32 residual r = y - yhat,
 * one fused pass per visit (num = a_i.r + x_i*sum(a_i^2)), same epsilon rule
 * for which coefficients enter the residual (cd.c:27).  Used to bound what
 * fp32 alone does to W; the GPU adds only a different summation order.        */
static int32_t cd_fp32(const cview_t *A, const oracle_cfg_t *cfg, fkv_t *act,
                       int32_t na, int32_t maxniters, float *x, float *r,
                       int32_t item, uint32_t *lstate, fkv_t *tmp,
                       int32_t *r_niters, int64_t *D, int64_t *U) {
  int32_t t, rstatus = 0;
  const float l1 = (float)cfg->l1r, l2 = (float)cfg->l2r;
  const float eps = (float)ORACLE_EPS;
  for (int32_t i = 0; i < na; i++) {
    int32_t iI = (int32_t)act[i].val;
    float xi = x[iI];
    if (xi > eps || xi < -eps)
      for (int64_t j = A->colptr[iI]; j < A->colptr[iI + 1]; j++)
        r[A->colind[j]] -= xi * (A->colval ? A->colval[j] : 1.0f);
  }
  for (t = 0; t < maxniters; t++) {
    float dltx = 0.0f;
    const fkv_t *visit = act;
    if (cfg->order == ORDER_GLIBC)
      shuffle_glibc(act, na);
    else if (cfg->order == ORDER_LOCAL)
      shuffle_local(act, na, lstate);
    else if (cfg->order == ORDER_PERM) {
      uint32_t key = oracle_perm_key(cfg->seed, (uint32_t)item, (uint32_t)t);
      for (int32_t p = 0; p < na; p++)
        tmp[p] = act[oracle_perm_index((uint32_t)p, (uint32_t)na, key)];
      visit = tmp;
    }
    for (int32_t i = 0; i < na; i++) {
      int32_t iI = (int32_t)visit[i].val;
      float cn = A->cnorms[iI];
      float xi = x[iI];
      float xeff = (xi > eps || xi < -eps) ? xi : 0.0f;
      float dot = 0.0f, ss = 0.0f;
      for (int64_t j = A->colptr[iI]; j < A->colptr[iI + 1]; j++) {
        float v = A->colval ? A->colval[j] : 1.0f;
        dot += v * r[A->colind[j]];
        ss += v * v;
      }
      float num = dot + xeff * ss;
      float newxi = num > l1 ? (num - l1) / (cn * cn + l2) : 0.0f;
      float neff = (newxi > eps || newxi < -eps) ? newxi : 0.0f;
      float d = neff - xeff;
      int64_t len = A->colptr[iI + 1] - A->colptr[iI];
      if (d != 0.0f) {
        for (int64_t j = A->colptr[iI]; j < A->colptr[iI + 1]; j++)
          r[A->colind[j]] -= d * (A->colval ? A->colval[j] : 1.0f);
        *U += len;
      }
      *D += len;
      x[iI] = newxi;
      dltx += (newxi - xi) * (newxi - xi);
    }
    if (dltx < (float)cfg->optTol) {
      rstatus = 1;
      break;
    }
  }
  *r_niters = t + 1;
  return rstatus;
}

/* ------------------------------------------------------------------------ */
/* estimate.c:328-558 EstimateModelCD (+ SaveModel's column view, :570-589)   */
/* ------------------------------------------------------------------------ */
/* Inputs: CSR (rowval may be NULL).  imodel_* : column view of a previous model
 * (warm start, estimate.c:453-464) or NULL.  colsel: optional list of the
 * columns to solve (others produce empty columns) -- the reference always
 * solves all; the subset exists for bounded CPU-baseline samples.
 * Outputs (malloc'd, release with oracle_free): colptr[ncols+1], colind, colval
 * = column iC holds the regressors of item iC, ascending ids, float values.
 * stats: optional array of ncols entries.  Returns ncols, or <0 on error.     */
int32_t oracle_learn_cd(int32_t nrows, const int64_t *rowptr,
                        const int32_t *rowind, const float *rowval,
                        const oracle_cfg_t *cfg, const int64_t *imodel_colptr,
                        const int32_t *imodel_colind,
                        const float *imodel_colval, int32_t imodel_ncols,
                        int32_t ncolsel, const int32_t *colsel,
                        int64_t **r_colptr, int32_t **r_colind,
                        float **r_colval, oracle_colstat_t *stats,
                        double *r_error, double *r_objval) {
  const int64_t nnz = rowptr[nrows];
  const int32_t ncols = oracle_ncols(nnz, rowind);
  if (ncols <= 0) return -1;

  /* CreateTrainingMatrix: column view + norms (setup.c:128-132) */
  colview_t cvw = colview_get(nrows, ncols, rowptr, rowind, rowval);
  int64_t *colptr = cvw.colptr;
  int32_t *colind = cvw.colind;
  float *colval = cvw.colval;
  float *cnorms = cvw.cnorms;
  cview_t A = {colptr, colind, colval, cnorms};

  int32_t *nnzs = (int32_t *)calloc((size_t)ncols, sizeof(int32_t));
  fkv_t **lists = (fkv_t **)calloc((size_t)ncols, sizeof(fkv_t *));
  double error = 0.0, objval = 0.0;
  const int32_t nwork = colsel ? ncolsel : ncols;
  int nthreads = cfg->nthreads > 0 ? cfg->nthreads : 1;
  if (stats) memset(stats, 0, sizeof(oracle_colstat_t) * (size_t)ncols);

  const int chunk = cfg->chunk > 0 ? cfg->chunk : 32;
  const double t_learn0 = now_seconds();
  g_deadline = g_time_budget > 0.0 ? t_learn0 + g_time_budget : 0.0;
  g_time_budget = 0.0; /* one call only */
#pragma omp parallel num_threads(nthreads) reduction(+ : error, objval)
  {
    /* per-thread dense work vectors, estimate.c:382-385 */
    double *x = (double *)calloc((size_t)ncols, sizeof(double));
    double *y = (double *)calloc((size_t)nrows, sizeof(double));
    double *yhat = (double *)calloc((size_t)nrows, sizeof(double));
    double *ATy = (double *)calloc((size_t)ncols, sizeof(double));
    float *xf = cfg->fp32 ? (float *)calloc((size_t)ncols, sizeof(float)) : NULL;
    float *rf = cfg->fp32 ? (float *)calloc((size_t)nrows, sizeof(float)) : NULL;
    fkv_t *act = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
    fkv_t *tmp = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
    int32_t *nmark = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
    fkv_t *ncand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
    for (int32_t i = 0; i < ncols; i++) nmark[i] = -1;
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    uint32_t lstate = 0x9E3779B9u ^ (cfg->seed * 2654435761u) ^
                      ((uint32_t)tid * 0x85EBCA6Bu + 1u);
    if (lstate == 0) lstate = 1;

#pragma omp for schedule(dynamic, chunk) /* estimate.c:402: chunk = 32 */
    for (int32_t w = 0; w < nwork; w++) {
      const int32_t iC = colsel ? colsel[w] : w;
      const int64_t cs = colptr[iC], ce = colptr[iC + 1];
      int64_t G = 0, D = 0, U = 0;
      if (g_deadline > 0.0 && now_seconds() > g_deadline) { /* budget spent: not started */
        if (stats) stats[iC].conv = -2;
        continue;
      }

      /* estimate.c:406-408 target vector */
      for (int64_t j = cs; j < ce; j++)
        y[colind[j]] = colval ? colval[j] : 1.0;

      /* estimate.c:412-421: ATy[i] = a_i . y for ALL columns i.  ATY_GRAM
       * computes the same vector through the users of column iC (sum over
       * u in col iC of val * row_u); identical in exact arithmetic.           */
      if (cfg->aty == ATY_FULLSCAN) {
        for (int32_t i = 0; i < ncols; i++) {
          double ip = 0.0;
          if (colval)
            for (int64_t j = colptr[i]; j < colptr[i + 1]; j++)
              ip += colval[j] * y[colind[j]];
          else
            for (int64_t j = colptr[i]; j < colptr[i + 1]; j++)
              ip += y[colind[j]];
          ATy[i] = ip;
        }
        for (int64_t j = cs; j < ce; j++)
          G += rowptr[colind[j] + 1] - rowptr[colind[j]];
      } else {
        for (int64_t j = cs; j < ce; j++) {
          const int32_t u = colind[j];
          const double v = colval ? colval[j] : 1.0;
          for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++)
            ATy[rowind[e]] += v * (rowval ? rowval[e] : 1.0);
          G += rowptr[u + 1] - rowptr[u];
        }
      }

      /* estimate.c:433-444 active set: strict '>' against l1r, diag excluded,
       * key stored as float, x flagged -0.1 for the warm-start test          */
      int32_t na = 0;
      if (cfg->nnbrs > 0) {
        /* estimate.c:424-431 FSLIM: the active set is the neighbour list, with no
         * l1 screen and without the -0.1 flags (warm start is a no-op there)    */
        na = find_neighbors(cfg, nrows, rowptr, rowind, rowval, &A, iC, nmark, ncand);
        if (cfg->order == ORDER_PERM) {
          /* the engine keeps every active list in ascending id order and permutes
           * positions; the reference keeps similarity order and shuffles it.  Either
           * only seeds the visiting order. */
          for (int32_t a = 1; a < na; a++) {
            fkv_t t = ncand[a];
            int32_t b = a;
            while (b > 0 && ncand[b - 1].val > t.val) {
              ncand[b] = ncand[b - 1];
              b--;
            }
            ncand[b] = t;
          }
        }
        for (int32_t i = 0; i < na; i++) {
          act[i].val = ncand[i].val;
          act[i].key = (float)ATy[ncand[i].val];
        }
      } else {
        for (int32_t i = 0; i < ncols; i++) {
          if (ATy[i] > cfg->l1r && i != iC) {
            act[na].val = i;
            act[na].key = (float)ATy[i];
            na++;
            x[i] = -0.1;
          }
        }
      }

      /* estimate.c:448-449 adaptive sweep cap */
      int64_t cap = 50 * (ce - cs);
      int32_t maxit = cap < cfg->maxniters ? (int32_t)cap : cfg->maxniters;

      /* estimate.c:453-471 initial solution */
      if (imodel_colptr && iC < imodel_ncols) {
        for (int64_t j = imodel_colptr[iC]; j < imodel_colptr[iC + 1]; j++) {
          int32_t k = imodel_colind[j];
          if (k < ncols) x[k] = x[k] < 0. ? imodel_colval[j] : 0.0;
        }
        for (int32_t i = 0; i < na; i++) {
          int64_t k = act[i].val;
          x[k] = x[k] < 0. ? 0.0 : x[k];
        }
      } else {
        for (int32_t i = 0; i < na; i++) x[act[i].val] = 0.0;
      }

      int32_t niters = 0, rstatus;
      double soln_rNorm = 0.0;
      if (!cfg->fp32) {
        rstatus = cd_reference(&A, cfg, act, na, maxit, x, yhat, iC, &lstate,
                               tmp, &niters, &D, &U);
        /* estimate.c:477-481 */
        for (int32_t i = 0; i < nrows; i++)
          soln_rNorm += (y[i] - yhat[i]) * (y[i] - yhat[i]);
      } else {
        for (int64_t j = cs; j < ce; j++)
          rf[colind[j]] = colval ? colval[j] : 1.0f;
        for (int32_t i = 0; i < na; i++) xf[act[i].val] = (float)x[act[i].val];
        rstatus = cd_fp32(&A, cfg, act, na, maxit, xf, rf, iC, &lstate, tmp,
                          &niters, &D, &U);
        for (int32_t i = 0; i < na; i++) x[act[i].val] = xf[act[i].val];
        for (int32_t i = 0; i < nrows; i++)
          soln_rNorm += (double)rf[i] * (double)rf[i];
      }
      soln_rNorm *= 0.5;
      error += soln_rNorm;

      /* estimate.c:483-489 objective */
      double soln_obj = soln_rNorm;
      for (int32_t i = 0; i < ncols; i++)
        soln_obj += 0.5 * cfg->l2r * (x[i] * x[i]) + cfg->l1r * fabs(x[i]);
      objval += soln_obj;

      /* estimate.c:492-505 keep |x| > EPSILON, ascending i, float values */
      int32_t nz = 0;
      for (int32_t i = 0; i < ncols; i++)
        if (fabs(x[i]) > ORACLE_EPS) nz++;
      fkv_t *list = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)(nz ? nz : 1));
      nz = 0;
      for (int32_t i = 0; i < ncols; i++)
        if (fabs(x[i]) > ORACLE_EPS) {
          list[nz].key = (float)x[i];
          list[nz].val = i;
          nz++;
        }
      nnzs[iC] = nz;
      lists[iC] = list;

      if (stats) {
        stats[iC].nacols = na;
        stats[iC].sweeps = niters;
        stats[iC].conv = rstatus;
        stats[iC].nnzw = nz;
        stats[iC].G = G;
        stats[iC].D = D;
        stats[iC].U = U;
        stats[iC].err = soln_rNorm;
        stats[iC].obj = soln_obj;
      }

      /* estimate.c:517-530 restore the work vectors (sparsely where the
       * support is known; same end state as the reference's dense resets)    */
      for (int64_t j = cs; j < ce; j++) y[colind[j]] = 0.0;
      for (int32_t i = 0; i < na; i++) x[act[i].val] = 0.0;
      if (imodel_colptr && iC < imodel_ncols)
        for (int64_t j = imodel_colptr[iC]; j < imodel_colptr[iC + 1]; j++)
          if (imodel_colind[j] < ncols) x[imodel_colind[j]] = 0.0;
      if (cfg->aty == ATY_GRAM) {
        for (int64_t j = cs; j < ce; j++) {
          const int32_t u = colind[j];
          for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++) ATy[rowind[e]] = 0.0;
        }
      }
      if (!cfg->fp32)
        memset(yhat, 0, sizeof(double) * (size_t)nrows);
      else {
        memset(rf, 0, sizeof(float) * (size_t)nrows);
        for (int32_t i = 0; i < na; i++) xf[act[i].val] = 0.0f;
      }
    }
    free(x);
    free(y);
    free(yhat);
    free(ATy);
    free(xf);
    free(rf);
    free(act);
    free(tmp);
    free(nmark);
    free(ncand);
  }
  g_learn_seconds = now_seconds() - t_learn0;
  g_deadline = 0.0;

  /* estimate.c:570-589 SaveModel, column view */
  int64_t tnnz = 0;
  for (int32_t c = 0; c < ncols; c++) tnnz += nnzs[c];
  int64_t *wptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)ncols + 1));
  int32_t *wind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tnnz ? tnnz : 1));
  float *wval = (float *)malloc(sizeof(float) * (size_t)(tnnz ? tnnz : 1));
  wptr[0] = 0;
  tnnz = 0;
  for (int32_t c = 0; c < ncols; c++) {
    for (int32_t k = 0; k < nnzs[c]; k++, tnnz++) {
      wind[tnnz] = (int32_t)lists[c][k].val;
      wval[tnnz] = lists[c][k].key;
    }
    wptr[c + 1] = tnnz;
    free(lists[c]);
  }
  free(lists);
  free(nnzs);
  colview_put(&cvw);
  *r_colptr = wptr;
  *r_colind = wind;
  *r_colval = wval;
  if (r_error) *r_error = error;
  if (r_objval) *r_objval = objval;
  return ncols;
}

/* ------------------------------------------------------------------------ */
/* EstimateModelCD in the visiting order of the engine's tile kernel          */
/* ------------------------------------------------------------------------ */
/* slim_amd/csrc/cd_tile.hpp solves `tileP` item columns per workgroup in
 * lock-step: tile g = entries [g*tileP, (g+1)*tileP) of the work list `order`
 * (the engine sorts the requested columns by descending Gram work G, stable);
 * sweep t of the tile walks u[perm(p; key(seed, g, t))], p = 0..|u|-1, where u
 * is the ascending union of the members' active sets, and each member skips
 * the coordinates outside its own active set.  Per member the algorithm is
 * EstimateModelCD/CoordinateDescent unchanged (reference arithmetic: fp64,
 * 3-pass), so this is the same restatement with a different ShuffleList.     */
/* imodel_* : column view of a previous model (warm start, estimate.c:453-464: the
 * previous coefficients of the coordinates that are active now, folded into yhat by
 * cd.c:108-110 before the first sweep) or NULL.                                  */
int32_t oracle_learn_cd_tile_warm(int32_t nrows, const int64_t *rowptr,
                                  const int32_t *rowind, const float *rowval,
                                  const oracle_cfg_t *cfg, int32_t tileP,
                                  int32_t nwork, const int32_t *order,
                                  const int64_t *imodel_colptr,
                                  const int32_t *imodel_colind,
                                  const float *imodel_colval, int32_t imodel_ncols,
                                  int64_t **r_colptr, int32_t **r_colind,
                                  float **r_colval, oracle_colstat_t *stats,
                                  double *r_error, double *r_objval) {
  const int64_t nnz = rowptr[nrows];
  const int32_t ncols = oracle_ncols(nnz, rowind);
  if (ncols <= 0 || tileP <= 0) return -1;
  colview_t cvw = colview_get(nrows, ncols, rowptr, rowind, rowval);
  int64_t *colptr = cvw.colptr;
  int32_t *colind = cvw.colind;
  float *colval = cvw.colval;
  float *cnorms = cvw.cnorms;
  cview_t A = {colptr, colind, colval, cnorms};
  int32_t *nnzs = (int32_t *)calloc((size_t)ncols, sizeof(int32_t));
  fkv_t **lists = (fkv_t **)calloc((size_t)ncols, sizeof(fkv_t *));
  double error = 0.0, objval = 0.0;
  if (stats) memset(stats, 0, sizeof(oracle_colstat_t) * (size_t)ncols);
  const int32_t ntiles = (nwork + tileP - 1) / tileP;
  int nthreads = cfg->nthreads > 0 ? cfg->nthreads : 1;

  /* shared per-tile state; the threads work on the MEMBERS of one tile at a time (a
   * single tile of a 1M x 100K matrix is minutes of one core: the full-size parity tests
   * solve one tile on all cores), which gives the same result for any thread count      */
  float *key = (float *)malloc(sizeof(float) * (size_t)ncols * (size_t)tileP);
  uint8_t *act = (uint8_t *)malloc((size_t)ncols * (size_t)tileP);
  int32_t *uni = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
  int64_t *Gm = (int64_t *)malloc(sizeof(int64_t) * (size_t)tileP);
  int32_t nu = 0;
  const int32_t g_begin = cfg->tile_count > 0 ? cfg->tile_first : 0;
  const int32_t g_end =
      cfg->tile_count > 0 && g_begin + cfg->tile_count < ntiles ? g_begin + cfg->tile_count : ntiles;
  const double t_learn0 = now_seconds();

#pragma omp parallel num_threads(nthreads) reduction(+ : error, objval)
  {
    double *x = (double *)calloc((size_t)ncols, sizeof(double));
    double *y = (double *)calloc((size_t)nrows, sizeof(double));
    double *yhat = (double *)calloc((size_t)nrows, sizeof(double));
    double *ATy = (double *)calloc((size_t)ncols, sizeof(double));
    int32_t *nmark = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
    fkv_t *ncand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
    for (int32_t i = 0; i < ncols; i++) nmark[i] = -1;

    for (int32_t g = g_begin; g < g_end; g++) {
      const int32_t base = g * tileP;
      const int32_t np = (nwork - base) < tileP ? (nwork - base) : tileP;
      /* active sets of the members (estimate.c:406-444, Gram-column aTy) */
#pragma omp for schedule(dynamic, 1)
      for (int32_t m = 0; m < np; m++) {
        const int32_t iC = order[base + m];
        memset(act + (size_t)m * ncols, 0, (size_t)ncols);
        Gm[m] = 0;
        for (int64_t j = colptr[iC]; j < colptr[iC + 1]; j++) {
          const int32_t u = colind[j];
          const double v = colval ? colval[j] : 1.0;
          for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++)
            ATy[rowind[e]] += v * (rowval ? rowval[e] : 1.0);
          Gm[m] += rowptr[u + 1] - rowptr[u];
        }
        if (cfg->nnbrs > 0) { /* estimate.c:424-431 FSLIM: the neighbour list, no l1 screen */
          const int32_t nn =
              find_neighbors(cfg, nrows, rowptr, rowind, rowval, &A, iC, nmark, ncand);
          for (int32_t i = 0; i < nn; i++) {
            act[(size_t)m * ncols + ncand[i].val] = 1;
            key[(size_t)m * ncols + ncand[i].val] = (float)ATy[ncand[i].val];
          }
        } else
        for (int32_t i = 0; i < ncols; i++) {
          if (ATy[i] > cfg->l1r && i != iC) {
            act[(size_t)m * ncols + i] = 1;
            key[(size_t)m * ncols + i] = (float)ATy[i];
          }
        }
        for (int64_t j = colptr[iC]; j < colptr[iC + 1]; j++) {
          const int32_t u = colind[j];
          for (int64_t e = rowptr[u]; e < rowptr[u + 1]; e++) ATy[rowind[e]] = 0.0;
        }
      }
#pragma omp single
      {
        nu = 0;
        for (int32_t i = 0; i < ncols; i++) {
          int any = 0;
          for (int32_t m = 0; m < np; m++) any |= act[(size_t)m * ncols + i];
          if (any) uni[nu++] = i;
        }
      }
      /* every member: CoordinateDescent (cd.c:101-142) in the tile's order */
#pragma omp for schedule(dynamic, 1)
      for (int32_t m = 0; m < np; m++) {
        const int32_t iC = order[base + m];
        const uint8_t *am = act + (size_t)m * ncols;
        const float *km = key + (size_t)m * ncols;
        const int64_t cs = colptr[iC], ce = colptr[iC + 1];
        int64_t D = 0, U = 0;
        int32_t na = 0;
        for (int32_t k = 0; k < nu; k++) na += am[uni[k]];
        for (int64_t j = cs; j < ce; j++) y[colind[j]] = colval ? colval[j] : 1.0;
        int64_t cap = 50 * (ce - cs);
        int32_t maxit = cap < cfg->maxniters ? (int32_t)cap : cfg->maxniters;
        /* estimate.c:453-464 initial solution (in the FSLIM branch the reference never sets
         * its -0.1 flags, estimate.c:424-431: warm start is a no-op there) */
        if (imodel_colptr && iC < imodel_ncols && cfg->nnbrs <= 0) {
          for (int64_t j = imodel_colptr[iC]; j < imodel_colptr[iC + 1]; j++) {
            const int32_t k = imodel_colind[j];
            /* (a negative previous value ends up 0: estimate.c:456-457 copies it, the
             * flag-clearing loop :461-464 then resets every x < 0) */
            if (k < ncols && am[k]) x[k] = imodel_colval[j] < 0.f ? 0.0 : imodel_colval[j];
          }
          for (int32_t k = 0; k < nu; k++) /* cd.c:108-110 */
            if (am[uni[k]]) add_spvec(&A, uni[k], x[uni[k]], yhat);
        }
        int32_t t, rstatus = 0;
        for (t = 0; t < maxit; t++) {
          double dltx = 0.0;
          uint32_t pk = oracle_perm_key(cfg->seed, (uint32_t)g, (uint32_t)t);
          for (int32_t p = 0; p < nu; p++) {
            const int32_t iI = uni[oracle_perm_index((uint32_t)p, (uint32_t)nu, pk)];
            if (!am[iI]) continue;
            const double aTy = km[iI], aTa = cnorms[iI], xi = x[iI];
            const int64_t len = colptr[iI + 1] - colptr[iI];
            int64_t touched = add_spvec(&A, iI, -xi, yhat);
            const double ip = spvec_dot(&A, iI, yhat);
            const double num = aTy - ip;
            const double newxi =
                num > cfg->l1r ? (num - cfg->l1r) / ((aTa * aTa) + cfg->l2r) : 0.0;
            touched += add_spvec(&A, iI, newxi, yhat);
            x[iI] = newxi;
            dltx += (newxi - xi) * (newxi - xi);
            D += len;
            if (touched) U += len;
          }
          if (dltx < cfg->optTol) {
            rstatus = 1;
            break;
          }
        }
        const int32_t niters = t + 1;
        double rn = 0.0;
        for (int32_t i = 0; i < nrows; i++) rn += (y[i] - yhat[i]) * (y[i] - yhat[i]);
        rn *= 0.5;
        double ob = rn;
        int32_t nz = 0;
        for (int32_t k = 0; k < nu; k++) {
          const double xv = x[uni[k]];
          ob += 0.5 * cfg->l2r * xv * xv + cfg->l1r * fabs(xv);
          if (fabs(xv) > ORACLE_EPS) nz++;
        }
        error += rn;
        objval += ob;
        fkv_t *list = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)(nz ? nz : 1));
        nz = 0;
        for (int32_t k = 0; k < nu; k++)
          if (fabs(x[uni[k]]) > ORACLE_EPS) {
            list[nz].key = (float)x[uni[k]];
            list[nz].val = uni[k];
            nz++;
          }
        nnzs[iC] = nz;
        lists[iC] = list;
        if (stats) {
          stats[iC].nacols = na;
          stats[iC].sweeps = niters;
          stats[iC].conv = rstatus;
          stats[iC].nnzw = nz;
          stats[iC].G = Gm[m];
          stats[iC].D = D;
          stats[iC].U = U;
          stats[iC].err = rn;
          stats[iC].obj = ob;
        }
        for (int64_t j = cs; j < ce; j++) y[colind[j]] = 0.0;
        for (int32_t k = 0; k < nu; k++) x[uni[k]] = 0.0;
        memset(yhat, 0, sizeof(double) * (size_t)nrows);
      } /* (implicit barrier: the next tile rewrites act / key / uni) */
    }
    free(x); free(y); free(yhat); free(ATy); free(nmark); free(ncand);
  }
  g_learn_seconds = now_seconds() - t_learn0;
  g_deadline = 0.0;
  free(key); free(act); free(uni); free(Gm);

  int64_t tnnz = 0;
  for (int32_t c = 0; c < ncols; c++) tnnz += nnzs[c];
  int64_t *wptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)ncols + 1));
  int32_t *wind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tnnz ? tnnz : 1));
  float *wval = (float *)malloc(sizeof(float) * (size_t)(tnnz ? tnnz : 1));
  wptr[0] = 0;
  tnnz = 0;
  for (int32_t c = 0; c < ncols; c++) {
    for (int32_t k = 0; k < nnzs[c]; k++, tnnz++) {
      wind[tnnz] = (int32_t)lists[c][k].val;
      wval[tnnz] = lists[c][k].key;
    }
    wptr[c + 1] = tnnz;
    free(lists[c]);
  }
  free(lists); free(nnzs);
  colview_put(&cvw);
  *r_colptr = wptr;
  *r_colind = wind;
  *r_colval = wval;
  if (r_error) *r_error = error;
  if (r_objval) *r_objval = objval;
  return ncols;
}

int32_t oracle_learn_cd_tile(int32_t nrows, const int64_t *rowptr,
                             const int32_t *rowind, const float *rowval,
                             const oracle_cfg_t *cfg, int32_t tileP,
                             int32_t nwork, const int32_t *order,
                             int64_t **r_colptr, int32_t **r_colind,
                             float **r_colval, oracle_colstat_t *stats,
                             double *r_error, double *r_objval) {
  return oracle_learn_cd_tile_warm(nrows, rowptr, rowind, rowval, cfg, tileP, nwork, order,
                                   NULL, NULL, NULL, 0, r_colptr, r_colind, r_colval, stats,
                                   r_error, r_objval);
}

/* ------------------------------------------------------------------------ */
/* EstimateModelADMM (estimate.c:38-304; MKL-only in the reference)           */
/* ------------------------------------------------------------------------ */
/* Plain-loop restatement in the reference's precision and operation order where it is
 * observable: T = R^T R, P = (T + (l2 + rho) I)^-1 by Cholesky (dpotrf / dpotri,
 * estimate.c:150-163), A = P T, then 30 iterations of estimate.c:166-213 with rho = 1e4.
 * The products are naive triple loops (k ascending), so they differ from a BLAS by the
 * summation order only.  Returns W's row view (positive entries, float values).
 * Parity note: the reference ships no output of this path (it needs MKL) -- unpinned.      */
int32_t oracle_learn_admm(int32_t nrows, const int64_t *rowptr, const int32_t *rowind,
                          const float *rowval, double l1r, double l2r, int32_t nthreads,
                          int64_t **r_rowptr, int32_t **r_rowind, float **r_rowval) {
  const int64_t nnz = rowptr[nrows];
  const int32_t m = oracle_ncols(nnz, rowind);
  if (m <= 0) return -1;
  const double RHO = 10000.0;
  const int MAXITERS = 30;
  const size_t n2 = (size_t)m * (size_t)m;
  double *T = (double *)calloc(n2, sizeof(double)), *A = (double *)calloc(n2, sizeof(double));
  double *B = (double *)calloc(n2, sizeof(double)), *W = (double *)calloc(n2, sizeof(double));
  double *C = (double *)calloc(n2, sizeof(double)), *P = (double *)calloc(n2, sizeof(double));
  double *L = (double *)calloc(n2, sizeof(double)), *Li = (double *)calloc(n2, sizeof(double));
  double *gamma = (double *)calloc((size_t)m, sizeof(double));
  if (nthreads < 1) nthreads = 1;
  /* T = Rt R (estimate.c:124-125) */
  for (int32_t u = 0; u < nrows; u++)
    for (int64_t a = rowptr[u]; a < rowptr[u + 1]; a++)
      for (int64_t b = rowptr[u]; b < rowptr[u + 1]; b++)
        T[(size_t)rowind[a] * m + rowind[b]] +=
            (double)(rowval ? rowval[a] : 1.0f) * (double)(rowval ? rowval[b] : 1.0f);
  /* P = T + (l2 + rho) I, Cholesky P = L L^T, P^-1 = L^-T L^-1 */
  for (size_t k = 0; k < n2; k++) P[k] = T[k];
  for (int32_t i = 0; i < m; i++) P[(size_t)i * m + i] += l2r + RHO;
  for (int32_t j = 0; j < m; j++) {
    double d = P[(size_t)j * m + j];
    for (int32_t k = 0; k < j; k++) d -= L[(size_t)j * m + k] * L[(size_t)j * m + k];
    if (d <= 0.0) return -2;
    d = sqrt(d);
    L[(size_t)j * m + j] = d;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int32_t i = j + 1; i < m; i++) {
      double v = P[(size_t)i * m + j];
      for (int32_t k = 0; k < j; k++) v -= L[(size_t)i * m + k] * L[(size_t)j * m + k];
      L[(size_t)i * m + j] = v / d;
    }
  }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
  for (int32_t c = 0; c < m; c++) { /* column c of L^-1 by forward substitution */
    for (int32_t i = c; i < m; i++) {
      double v = i == c ? 1.0 : 0.0;
      for (int32_t k = c; k < i; k++) v -= L[(size_t)i * m + k] * Li[(size_t)k * m + c];
      Li[(size_t)i * m + c] = v / L[(size_t)i * m + i];
    }
  }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
  for (int32_t i = 0; i < m; i++) /* P^-1 = Li^T Li */
    for (int32_t j = 0; j <= i; j++) {
      double v = 0.0;
      for (int32_t k = i; k < m; k++) v += Li[(size_t)k * m + i] * Li[(size_t)k * m + j];
      P[(size_t)i * m + j] = P[(size_t)j * m + i] = v;
    }
  /* A = P T (estimate.c:166-167) */
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int32_t i = 0; i < m; i++)
    for (int32_t k = 0; k < m; k++) {
      const double p = P[(size_t)i * m + k];
      for (int32_t j = 0; j < m; j++) A[(size_t)i * m + j] += p * T[(size_t)k * m + j];
    }
  const double irho = 1.0 / RHO, kappa = l1r / RHO;
  for (int iter = 0; iter < MAXITERS; iter++) { /* estimate.c:169-213 */
    for (size_t k = 0; k < n2; k++) W[k] = RHO * W[k];
    for (size_t k = 0; k < n2; k++) W[k] = W[k] - C[k];
    memset(T, 0, sizeof(double) * n2);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int32_t i = 0; i < m; i++)
      for (int32_t k = 0; k < m; k++) {
        const double p = P[(size_t)i * m + k];
        for (int32_t j = 0; j < m; j++) T[(size_t)i * m + j] += p * W[(size_t)k * m + j];
      }
    for (size_t k = 0; k < n2; k++) T[k] = T[k] + A[k];
    for (int32_t i = 0; i < m; i++) gamma[i] = T[(size_t)i * m + i] / P[(size_t)i * m + i];
    for (int32_t i = 0; i < m; i++)
      for (int32_t j = 0; j < m; j++) B[(size_t)i * m + j] = -1.0 * P[(size_t)i * m + j] * gamma[j];
    for (size_t k = 0; k < n2; k++) B[k] = B[k] + T[k];
    for (size_t k = 0; k < n2; k++) {
      const double alpha = B[k] + irho * C[k];
      const double hi = alpha - kappa > 0.0 ? alpha - kappa : 0.0;
      const double lo = -alpha - kappa > 0.0 ? -alpha - kappa : 0.0;
      const double temp = hi - lo;
      W[k] = temp > 0.0 ? temp : 0.0;
    }
    for (size_t k = 0; k < n2; k++) B[k] = B[k] - W[k];
    for (size_t k = 0; k < n2; k++) B[k] = RHO * B[k];
    for (size_t k = 0; k < n2; k++) C[k] = C[k] + B[k];
  }
  int64_t cnt = 0;
  for (size_t k = 0; k < n2; k++) cnt += W[k] > 0.0;
  int64_t *wptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)m + 1));
  int32_t *wind = (int32_t *)malloc(sizeof(int32_t) * (size_t)(cnt ? cnt : 1));
  float *wval = (float *)malloc(sizeof(float) * (size_t)(cnt ? cnt : 1));
  cnt = 0;
  wptr[0] = 0;
  for (int32_t i = 0; i < m; i++) { /* estimate.c:228-262 */
    for (int32_t j = 0; j < m; j++)
      if (W[(size_t)i * m + j] > 0.0) {
        wind[cnt] = j;
        wval[cnt] = (float)W[(size_t)i * m + j];
        cnt++;
      }
    wptr[i + 1] = cnt;
  }
  free(T); free(A); free(B); free(W); free(C); free(P); free(L); free(Li); free(gamma);
  *r_rowptr = wptr;
  *r_rowind = wind;
  *r_rowval = wval;
  return m;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* predict.c:15-71 GetRecommendations                                         */
/* ------------------------------------------------------------------------ */
static int fkv_desc(const void *a, const void *b) {
  const fkv_t *x = (const fkv_t *)a, *y = (const fkv_t *)b;
  if (x->key > y->key) return -1;
  if (x->key < y->key) return 1;
  return 0;
}
/* stable descending sort by key (merge sort): ties keep insertion order.    */
static void fkv_sortd_stable(fkv_t *a, int64_t n) {
  if (n < 2) return;
  fkv_t *b = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)n);
  for (int64_t w = 1; w < n; w *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) b[k++] = (fkv_desc(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
      while (i < mid) b[k++] = a[i++];
      while (j < hi) b[k++] = a[j++];
    }
    memcpy(a, b, sizeof(fkv_t) * (size_t)n);
  }
  free(b);
}

/* W given by its ROW view (rowptr/rowind/rowval over ncols rows): the score of
 * candidate k is sum over history items i of rating_i * W[i,k].  float
 * accumulation in history order, history excluded, sort desc, take N.
 * marker/cand are caller scratch of ncols entries (marker preset to -1);
 * both are restored on return.                                                */
static int32_t get_recs(int32_t ncols, const int64_t *rowptr,
                        const int32_t *rowind, const float *rowval,
                        int32_t nratings, const int32_t *itemids,
                        const float *ratings, int32_t nrcmds, int32_t *rids,
                        float *rscores, int32_t *marker, fkv_t *cand) {
  for (int32_t r = 0; r < nratings; r++) /* predict.c:35-38 */
    if (itemids[r] < ncols && itemids[r] >= 0) marker[itemids[r]] = -2;
  int32_t ncand = 0;
  for (int32_t r = 0; r < nratings; r++) { /* predict.c:40-58 */
    int32_t i = itemids[r];
    if (i >= ncols || i < 0) continue; /* the reference's guard (&&) never
                       fires; out-of-range ids are undefined behaviour there */
    float rating = ratings ? ratings[r] : 1.0f;
    for (int64_t j = rowptr[i]; j < rowptr[i + 1]; j++) {
      int32_t k = rowind[j];
      if (marker[k] == -2) continue;
      if (marker[k] == -1) {
        cand[ncand].val = k;
        cand[ncand].key = 0.0f;
        marker[k] = ncand++;
      }
      cand[marker[k]].key += rating * rowval[j];
    }
  }
  fkv_sortd_stable(cand, ncand); /* predict.c:60 gk_fkvsortd */
  int32_t n = ncand < nrcmds ? ncand : nrcmds;
  for (int32_t r = 0; r < n; r++) {
    rids[r] = (int32_t)cand[r].val;
    rscores[r] = cand[r].key;
  }
  for (int32_t r = 0; r < ncand; r++) marker[cand[r].val] = -1;
  for (int32_t r = 0; r < nratings; r++)
    if (itemids[r] < ncols && itemids[r] >= 0) marker[itemids[r]] = -1;
  return n;
}

int32_t oracle_get_topn(int32_t ncols, const int64_t *wrowptr,
                        const int32_t *wrowind, const float *wrowval,
                        int32_t nratings, const int32_t *itemids,
                        const float *ratings, int32_t nrcmds, int32_t *rids,
                        float *rscores) {
  int32_t *marker = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
  fkv_t *cand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
  for (int32_t i = 0; i < ncols; i++) marker[i] = -1;
  int32_t n = get_recs(ncols, wrowptr, wrowind, wrowval, nratings, itemids,
                       ratings, nrcmds, rids, rscores, marker, cand);
  free(marker);
  free(cand);
  return n;
}

/* predict.c:77-133 GetRec_1vsk + pyapi.c:483-528 Py_SLIM_Predict_1vsk: user u ranks the
 * nnegs candidates negitems[u*nnegs ..]; a candidate id repeated in the list scores through
 * its LAST position (predict.c:94-96), ids outside [0, ncols) stay candidates with score 0;
 * float accumulation in history order; sort descending (stable here: upstream leaves ties
 * undefined), take nrcmds.                                                              */
int32_t oracle_predict_1vsk(int32_t ncols, const int64_t *wrowptr,
                            const int32_t *wrowind, const float *wrowval,
                            int32_t nusers, const int64_t *hptr, const int32_t *hind,
                            const float *hval, int32_t nrcmds, int32_t nnegs,
                            const int32_t *negitems, int32_t *out, float *scores) {
  int32_t *marker = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
  fkv_t *cand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)(nnegs > 0 ? nnegs : 1));
  for (int32_t i = 0; i < ncols; i++) marker[i] = -2;
  for (int32_t u = 0; u < nusers; u++) {
    const int32_t *neg = negitems + (int64_t)u * nnegs;
    int32_t ncand = 0;
    for (int32_t c = 0; c < nnegs; c++) { /* predict.c:91-101 */
      cand[ncand].val = neg[c];
      cand[ncand].key = 0.0f;
      if (neg[c] >= 0 && neg[c] < ncols) marker[neg[c]] = ncand;
      ncand++;
    }
    for (int64_t e = hptr[u]; e < hptr[u + 1]; e++) { /* predict.c:103-117 */
      const int32_t i = hind[e];
      if (i >= ncols || i < 0) continue;
      const float rating = hval ? hval[e] : 1.0f;
      for (int64_t j = wrowptr[i]; j < wrowptr[i + 1]; j++) {
        const int32_t k = wrowind[j];
        if (marker[k] == -2) continue;
        cand[marker[k]].key += rating * wrowval[j];
      }
    }
    fkv_sortd_stable(cand, ncand); /* predict.c:119 */
    const int32_t n = ncand < nrcmds ? ncand : nrcmds;
    for (int32_t r = 0; r < n; r++) {
      out[(int64_t)u * nrcmds + r] = (int32_t)cand[r].val;
      scores[(int64_t)u * nrcmds + r] = cand[r].key;
    }
    for (int32_t c = 0; c < nnegs; c++)
      if (neg[c] >= 0 && neg[c] < ncols) marker[neg[c]] = -2;
  }
  free(marker);
  free(cand);
  return 1;
}

/* pyapi.c:530-563 Py_SLIM_Predict: top-N for every row of the history matrix;
 * out[u*nrcmds + r], untouched slots keep the caller's fill.                  */
int32_t oracle_predict(int32_t ncols, const int64_t *wrowptr,
                       const int32_t *wrowind, const float *wrowval,
                       int32_t nusers, const int64_t *hptr, const int32_t *hind,
                       const float *hval, int32_t nrcmds, int32_t *out,
                       float *scores) {
  int32_t *marker = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
  fkv_t *cand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
  int32_t *rids = (int32_t *)malloc(sizeof(int32_t) * (size_t)nrcmds);
  float *rsc = (float *)malloc(sizeof(float) * (size_t)nrcmds);
  for (int32_t i = 0; i < ncols; i++) marker[i] = -1;
  for (int32_t u = 0; u < nusers; u++) {
    int32_t n = get_recs(ncols, wrowptr, wrowind, wrowval,
                         (int32_t)(hptr[u + 1] - hptr[u]), hind + hptr[u],
                         hval ? hval + hptr[u] : NULL, nrcmds, rids, rsc, marker,
                         cand);
    for (int32_t r = 0; r < n; r++) {
      out[(int64_t)u * nrcmds + r] = rids[r];
      scores[(int64_t)u * nrcmds + r] = rsc[r];
    }
  }
  free(marker);
  free(cand);
  free(rids);
  free(rsc);
  return 1;
}

/* api.c:215-245 SLIM_DetermineHeadAndTail: 0 = head (most popular items that
 * cover the first half of the ratings), 1 = tail.  gk_ikvsortd tie order is
 * undefined upstream; stable (ascending id among equal counts) here.          */
typedef struct {
  int32_t key;
  int32_t val;
} ikv_t;
static int ikv_desc(const void *a, const void *b) {
  const ikv_t *x = (const ikv_t *)a, *y = (const ikv_t *)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->val < y->val ? -1 : (x->val > y->val);
}
void oracle_head_tail(int32_t nrows, int32_t ncols, const int64_t *rowptr,
                      const int32_t *rowind, int32_t *fmarker) {
  ikv_t *cand = (ikv_t *)malloc(sizeof(ikv_t) * (size_t)ncols);
  for (int32_t c = 0; c < ncols; c++) {
    cand[c].key = 0;
    cand[c].val = c;
    fmarker[c] = 1;
  }
  for (int64_t j = 0; j < rowptr[nrows]; j++) cand[rowind[j]].key++;
  qsort(cand, (size_t)ncols, sizeof(ikv_t), ikv_desc);
  int64_t cnnz = rowptr[nrows] / 2;
  for (int32_t c = 0; c < ncols && cnnz > 0; c++) {
    fmarker[cand[c].val] = 0;
    cnnz -= cand[c].key;
  }
  free(cand);
}

/* HR / ARHR evaluation: pyapi.c:309-366 (== slim_mselect.c:122-187; users with
 * an empty test row are skipped, unlike slim_predict.c:226).
 * res[0]=HR res[1]=HR_head res[2]=HR_tail res[3]=ARHR; counts[0]=nvalid
 * counts[1]=nvalid_head counts[2]=nvalid_tail.                                */
void oracle_eval(int32_t ncols, const int64_t *wrowptr, const int32_t *wrowind,
                 const float *wrowval, int32_t nusers, const int64_t *trnptr,
                 const int32_t *trnind, const float *trnval,
                 const int64_t *tstptr, const int32_t *tstind, int32_t nrcmds,
                 int32_t fm_ncols, double *res, int32_t *counts) {
  int32_t mcols = ncols > fm_ncols ? ncols : fm_ncols;
  int32_t *marker = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncols);
  fkv_t *cand = (fkv_t *)malloc(sizeof(fkv_t) * (size_t)ncols);
  int32_t *rids = (int32_t *)malloc(sizeof(int32_t) * (size_t)nrcmds);
  float *rsc = (float *)malloc(sizeof(float) * (size_t)nrcmds);
  int32_t *rmarker = (int32_t *)malloc(sizeof(int32_t) * (size_t)mcols);
  int32_t *fmarker = (int32_t *)malloc(sizeof(int32_t) * (size_t)mcols);
  for (int32_t i = 0; i < ncols; i++) marker[i] = -1;
  for (int32_t i = 0; i < mcols; i++) rmarker[i] = -1;
  oracle_head_tail(nusers, mcols, trnptr, trnind, fmarker);

  float hr[3] = {0, 0, 0}, arhr = 0;
  int32_t nvalid = 0, nvh = 0, nvt = 0;
  for (int32_t u = 0; u < nusers; u++) {
    if (tstptr[u + 1] - tstptr[u] < 1) continue;
    int32_t n = get_recs(ncols, wrowptr, wrowind, wrowval,
                         (int32_t)(trnptr[u + 1] - trnptr[u]), trnind + trnptr[u],
                         trnval ? trnval + trnptr[u] : NULL, nrcmds, rids, rsc,
                         marker, cand);
    nvalid++;
    int is_t = 0, is_h = 0;
    int32_t ntrue[2] = {0, 0}, nhits[3] = {0, 0, 0};
    float larhr = 0, baseline = 0;
    for (int64_t z = tstptr[u]; z < tstptr[u + 1]; z++) {
      rmarker[tstind[z]] = u;
      ntrue[fmarker[tstind[z]]]++;
      if (fmarker[tstind[z]]) is_t = 1; else is_h = 1;
      baseline += 1.0 / (1.0 + z - tstptr[u]);
    }
    nvt += is_t;
    nvh += is_h;
    for (int32_t r = 0; r < n; r++)
      if (rmarker[rids[r]] == u) {
        nhits[fmarker[rids[r]]]++;
        nhits[2]++;
        larhr += 1.0 / (1.0 + r);
      }
    hr[0] += (nhits[0] > 0 ? 1.0 * nhits[0] / ntrue[0] : 0.0);
    hr[1] += (nhits[1] > 0 ? 1.0 * nhits[1] / ntrue[1] : 0.0);
    hr[2] += 1.0 * nhits[2] / (tstptr[u + 1] - tstptr[u]);
    arhr += larhr / baseline;
  }
  /* the reference keeps these in float (pyapi.c:223-230,360-363) */
  float all_hr = nvalid > 0 ? hr[2] / nvalid : 0;
  float head_hr = nvh > 0 ? hr[0] / nvh : 0;
  float tail_hr = nvt > 0 ? hr[1] / nvt : 0;
  arhr = nvalid > 0 ? arhr / nvalid : 0;
  res[0] = all_hr;
  res[1] = head_hr;
  res[2] = tail_hr;
  res[3] = arhr;
  counts[0] = nvalid;
  counts[1] = nvh;
  counts[2] = nvt;
  free(marker);
  free(cand);
  free(rids);
  free(rsc);
  free(rmarker);
  free(fmarker);
}

/* re-seed libc rand() so ORDER_GLIBC runs are repeatable inside one process
 * (a fresh reference process starts from srand(1), the C default).            */
void oracle_srand(uint32_t s) { srand(s); }

int32_t oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
