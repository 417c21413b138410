// slim_abi.cpp -- every exported symbol of libslim.so (include/slim.h, slim_gpu.h).
//
// Thin C-ABI shim: decodes the option arrays, moves data between the caller's
// buffers and the engine, never computes a CD update on the host.  SLIM_Learn /
// Py_SLIM_Learn / Py_SLIM_Mselect reach the solver only through
// slimamd::learn_cd (HIP, engine.hip); if no gfx950 device is usable they fail
// with SLIM_ERROR* and a message -- there is no CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "engine.hpp"
#include "host_csr.hpp"

using namespace slimamd;

namespace {

slim_csr_t* as_csr(slim_t* h) { return static_cast<slim_csr_t*>(h); }

// SLIM_Learn body shared by the C and Python entry points (the reference
// duplicates it: src/libslim/api.c:36-95 == src/libslim/pyapi.c:136-198).
slim_csr_t* learn_from_host(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                            const float* rowval, int32_t* ioptions, double* doptions,
                            const slim_csr_t* imodel, int32_t* status) {
  set_error("");
  LearnOptions opt = decode_options(ioptions, doptions);
  if (opt.algo == SLIM_ALGO_ADMM)  // estimate.c:38-304 (MKL-only in the reference): admm.hip
    return learn_admm(nrows, rowptr, rowind, rowval, opt, status);
  if (opt.algo != SLIM_ALGO_CD) {  // api.c:81-84 prints "Algorithm not supported" and exits
    set_error("unknown algorithm: algo must be cd (coordinate descent) or admm");
    *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  if (opt.simtype < SLIM_SIMTYPE_COS || opt.simtype > SLIM_SIMTYPE_DOTP) {
    set_error("unknown similarity type (neighbors.c:121-123)");
    *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  // ngpus (option slot 19, or SLIM_GPU_NGPUS in the environment for callers that cannot set
  // it -- the unchanged reference CLIs and Python wrapper): one host thread per device
  if (ioptions == nullptr || ioptions[SLIM_OPTION_GPU_NGPUS] == -1)
    if (const char* e = std::getenv("SLIM_GPU_NGPUS")) opt.ngpus = std::max(1, std::atoi(e));
  slimgpu_matrix_t* mat = multi_from_host(nrows, rowptr, rowind, rowval, opt, status);
  if (!mat) return nullptr;
  slim_csr_t* model = multi_learn(mat, opt, imodel, status);
  if (model && (opt.dbglvl & SLIM_DBG_TIME)) {  // timing.c:27-45
    const slimgpu_stats_t& s = last_stats();
    std::printf("\nTiming Information -------------------------------------------------");
    std::printf("\n Total: \t %7.3lf", (s.setup_ms + s.total_ms) / 1e3);
    std::printf("\n   Setup: \t\t %7.3lf", s.setup_ms / 1e3);
    std::printf("\n   Learn: \t\t %7.3lf", s.total_ms / 1e3);
    std::printf("\n     kernel: \t\t %7.3lf", s.kernel_ms / 1e3);
    std::printf("\n********************************************************************\n");
  }
  matrix_free(mat);
  return model;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- slim.h ------

int32_t SLIM_iSetDefaults(int32_t* options) {
  for (int i = 0; i < SLIM_NOPTIONS; ++i) options[i] = -1;
  return SLIM_OK;
}

int32_t SLIM_dSetDefaults(double* options) {
  for (int i = 0; i < SLIM_NOPTIONS; ++i) options[i] = -1;
  return SLIM_OK;
}

slim_t* SLIM_Learn(int32_t nrows, ssize_t* rowptr, int32_t* rowind, float* rowval,
                   int32_t* ioptions, double* doptions, slim_t* imodel, int32_t* r_status) {
  int32_t status = SLIM_ERROR;
  slim_csr_t* model =
      learn_from_host(nrows, rowptr, rowind, rowval, ioptions, doptions, as_csr(imodel), &status);
  if (r_status) *r_status = status;
  return model;
}

int32_t SLIM_GetTopN(slim_t* model, int32_t nratings, int32_t* itemids, float* ratings,
                     int32_t* /*ioptions*/, int32_t nrcmds, int32_t* rids, float* rscores) {
  const slim_csr_t* W = as_csr(model);
  if (!W || !W->rowptr || nrcmds < 0) return SLIM_ERROR;
  TopNScratch ws(W->ncols > W->nrows ? W->ncols : W->nrows);
  return top_n(W, nratings, itemids, ratings, nrcmds, rids, rscores, ws);
}

int32_t SLIM_WriteModel(slim_t* model, char* filename) {
  const slim_csr_t* W = as_csr(model);
  if (!W || !W->rowptr) return SLIM_ERROR_INPUT;
  return write_binrow(W, filename) ? SLIM_OK : SLIM_ERROR;
}

slim_t* SLIM_ReadModel(char* filename) {
  slim_csr_t* W = read_binrow(filename);
  if (W) csr_build_index(W, 0);  // api.c:191: add the column view
  return W;
}

void SLIM_FreeModel(slim_t** model) {
  if (!model) return;
  csr_free(as_csr(*model));
  *model = nullptr;
}

int32_t* SLIM_DetermineHeadAndTail(int32_t nrows, int32_t ncols, ssize_t* rowptr,
                                   int32_t* rowind) {
  return head_tail_split(nrows, ncols, rowptr, rowind);
}

// ------------------------------------------------------------ Py_* (pyapi.c) --

int32_t Py_csr_wrapper(int32_t nrows, ssize_t* rowptr, int32_t* rowind, float* rowval,
                       slim_t** matrix_out) {
  slim_csr_t* m = csr_from_rows(nrows, rowptr, rowind, rowval);
  if (!m) return SLIM_ERROR_MEMORY;
  *matrix_out = m;
  return SLIM_OK;
}

int32_t Py_csr_save(slim_t* mathandle, char* fname) {
  const slim_csr_t* m = as_csr(mathandle);
  if (!m || !m->rowptr) return SLIM_ERROR_INPUT;
  return write_text_csr(m, fname) ? SLIM_OK : SLIM_ERROR;
}

int32_t Py_csr_load(slim_t** mathandle, char* fname) {
  slim_csr_t* m = read_text_csr(fname);
  if (!m) return SLIM_ERROR;
  *mathandle = m;
  return SLIM_OK;
}

int32_t Py_csr_free(slim_t* mathandle) {
  csr_free(as_csr(mathandle));
  return SLIM_OK;
}

int32_t Py_csr_stat(slim_t* mathandle, int32_t* nnz) {
  const slim_csr_t* m = as_csr(mathandle);
  if (!m || !m->rowptr) return SLIM_ERROR_INPUT;
  *nnz = (int32_t)m->rowptr[m->nrows];
  return SLIM_OK;
}

int32_t Py_csr_export(slim_t* mathandle, int32_t* indptr, int32_t* indices, float* data) {
  const slim_csr_t* m = as_csr(mathandle);
  if (!m || !m->rowptr) return SLIM_ERROR_INPUT;
  const ssize_t nnz = m->rowptr[m->nrows];
  for (int32_t r = 0; r <= m->nrows; ++r) indptr[r] = (int32_t)m->rowptr[r];
  for (ssize_t k = 0; k < nnz; ++k) indices[k] = m->rowind[k];
  if (m->rowval)
    for (ssize_t k = 0; k < nnz; ++k) data[k] = m->rowval[k];
  return SLIM_OK;
}

int32_t Py_SLIM_Learn(slim_t* trnhandle, int32_t* ioptions, double* doptions,
                      slim_t** model_out) {
  const slim_csr_t* trn = as_csr(trnhandle);
  if (!trn || !trn->rowptr) return SLIM_ERROR_INPUT;
  int32_t status = SLIM_ERROR;
  slim_csr_t* model = learn_from_host(trn->nrows, trn->rowptr, trn->rowind, trn->rowval, ioptions,
                                      doptions, nullptr, &status);
  if (!model) return status;
  *model_out = model;
  return SLIM_OK;
}

int32_t Py_SLIM_Mselect(slim_t* trnhandle, slim_t* tsthandle, int32_t* ioptions,
                        double* doptions, double* arrayl1, double* arrayl2, int32_t nl1,
                        int32_t nl2, double* bestl1HR, double* bestl2HR, double* bestHRHR,
                        double* bestARHR, double* bestl1AR, double* bestl2AR, double* bestHRAR,
                        double* bestARAR) {
  const slim_csr_t* trn = as_csr(trnhandle);
  const slim_csr_t* tst = as_csr(tsthandle);
  if (!trn || !tst || !trn->rowptr || !tst->rowptr) return SLIM_ERROR_INPUT;
  set_error("");
  LearnOptions base = decode_options(ioptions, doptions);
  const int32_t nrcmds =
      (!ioptions || ioptions[SLIM_OPTION_NRCMDS] == -1) ? 10 : ioptions[SLIM_OPTION_NRCMDS];
  if (base.algo != SLIM_ALGO_CD && base.algo != SLIM_ALGO_ADMM) {
    set_error("Py_SLIM_Mselect: unknown algorithm");
    return SLIM_ERROR_INPUT;
  }
  const bool admm = base.algo == SLIM_ALGO_ADMM;

  // R goes to HBM once for the whole grid (the reference re-runs
  // CreateTrainingMatrix inside every SLIM_Learn call, pyapi.c:295-297)
  int32_t status = SLIM_ERROR;
  if (ioptions == nullptr || ioptions[SLIM_OPTION_GPU_NGPUS] == -1)
    if (const char* e = std::getenv("SLIM_GPU_NGPUS")) base.ngpus = std::max(1, std::atoi(e));
  slimgpu_matrix_t* mat =
      admm ? nullptr : multi_from_host(trn->nrows, trn->rowptr, trn->rowind, trn->rowval, base, &status);
  if (!mat && !admm) return status;
  // a grid of nl1 x nl2 solves over one R: the engine may pay for G = R^T R once (cd_gram.hpp)
  if (mat) matrix_expect_solves(mat, nl1 * nl2);

  const int32_t trn_ncols = max_index_plus_one(trn->rowptr[trn->nrows], trn->rowind);
  const int32_t tst_ncols = max_index_plus_one(tst->rowptr[tst->nrows], tst->rowind);
  const int32_t ncols = trn_ncols > tst_ncols ? trn_ncols : tst_ncols;  // pyapi.c:255-257
  int32_t* fmarker = head_tail_split(trn->nrows, ncols, trn->rowptr, trn->rowind);

  std::printf("------------------------------------------------------------------\n");
  std::printf("SLIM, version %s (MI355X engine)\n", SLIM_VERSION);
  std::printf("------------------------------------------------------------------\n");
  std::printf("  trn matrix, nrows: %d, ncols: %d, nnz: %zd\n", trn->nrows, ncols,
              trn->rowptr[trn->nrows]);
  std::printf("  tst matrix, nrows: %d, ncols: %d, nnz: %zd\n", tst->nrows, tst_ncols,
              tst->rowptr[tst->nrows]);
  std::printf("  optTol: %.2le, niters: %d\n", base.optTol, base.maxniters);
  std::printf("\nEstimating & evaluating models...\n\n");

  *bestHRHR = *bestARHR = *bestHRAR = *bestARAR = 0.0;
  slim_csr_t* model = nullptr;
  // One GPU, CD: the models of the grid stay in HBM (engine.hpp: learn_resident) -- each pair is
  // warm-started from the previous one without an upload and scored where it lies; only its nnz is
  // printed, so nothing of it ever crosses PCIe.  SLIM_GPU_RESIDENT=0: host models as before.
  const char* res_env = std::getenv("SLIM_GPU_RESIDENT");
  const bool resident = !admm && mat && matrix_replicas(mat).empty() && nrcmds >= 1 && nrcmds <= 128 &&
                        !(res_env && std::atoi(res_env) == 0);
  slimgpu_model* dmodel = nullptr;
  int32_t rc = SLIM_OK;
  for (int32_t a = 0; a < nl1 && rc == SLIM_OK; ++a) {
    for (int32_t b = 0; b < nl2; ++b) {
      doptions[SLIM_OPTION_L1R] = arrayl1[a];  // the reference writes these through
      doptions[SLIM_OPTION_L2R] = arrayl2[b];  // to the caller's array (pyapi.c:288-289)
      LearnOptions opt = base;
      opt.l1r = arrayl1[a];
      opt.l2r = arrayl2[b];
      slim_csr_t* prev = model;  // warm start from the previous cell
      // top-N of every user on the GPU (bit-identical to the host scorer), hits on the host
      std::vector<int32_t> lists((size_t)trn->nrows * nrcmds, -1), lens((size_t)trn->nrows, 0);
      std::vector<float> lsc((size_t)trn->nrows * nrcmds, 0.0f);
      bool on_gpu = false;
      ssize_t model_nnz_now = 0;
      if (resident) {
        slimgpu_model* dprev = dmodel;
        dmodel = learn_resident(mat, opt, dprev, &status);
        model_free(dprev);
        if (!dmodel) {
          rc = status;
          break;
        }
        model_nnz_now = (ssize_t)model_nnz(dmodel);
        DeviceRowView wv;
        on_gpu = model_row_view(dmodel, &wv) == SLIM_OK &&
                 predict_device_view(wv, trn, nrcmds, lists.data(), lsc.data(), lens.data()) == SLIM_OK;
        if (!on_gpu) {  // (the scorer refused: the host loop needs the host model)
          model = model_fetch(dmodel, &status);
          if (!model) {
            rc = status;
            break;
          }
        }
      } else {
      // (ADMM ignores the previous model, estimate.c:38)
      model = admm ? learn_admm(trn->nrows, trn->rowptr, trn->rowind, trn->rowval, opt, &status)
                   : multi_learn(mat, opt, prev, &status);
      csr_free(prev);
      if (!model) {
        rc = status;
        break;
      }
      model_nnz_now = model->rowptr[model->nrows];
      on_gpu = nrcmds <= 128 && predict_device(model, trn, nrcmds, lists.data(),
                                               lsc.data(), lens.data()) == SLIM_OK;
      }
      EvalResult ev;
      if (!on_gpu)
        ev = evaluate(model, trn, tst, nrcmds, fmarker, ncols);
      else if (evaluate_device(std::min(trn->nrows, tst->nrows), nrcmds, lists.data(), lens.data(),
                               tst, fmarker, ncols, &ev) != SLIM_OK) {  // hit counting on the GPU
        if (resident && !model) model = model_fetch(dmodel, &status);  // (the host loop sizes by the model)
        if (!model) {
          rc = status;
          break;
        }
        ev = evaluate(model, trn, tst, nrcmds, fmarker, ncols, lists.data(), lens.data());
      }
      std::printf("l1r: %.2le l2r: %.2le nnz: %7zd hr: %.4f hr_head: %.4f hr_tail: %.4f "
                  "arhr: %.4f time: %.2lf\n",
                  opt.l1r, opt.l2r, model_nnz_now, ev.hr, ev.hr_head, ev.hr_tail,
                  ev.arhr, last_stats().total_ms / 1e3);
      if (resident && model) {  // (a fetched copy served the host scorer only)
        csr_free(model);
        model = nullptr;
      }
      if (ev.nvalid < 1) {  // pyapi.c:377-381
        *bestl1HR = opt.l1r;
        *bestl2HR = opt.l2r;
        rc = SLIM_ERROR;
        break;
      }
      if (ev.hr > *bestHRHR) {
        *bestHRHR = ev.hr;
        *bestARHR = ev.arhr;
        *bestl1HR = opt.l1r;
        *bestl2HR = opt.l2r;
      }
      if (ev.arhr > *bestARAR) {
        *bestHRAR = ev.hr;
        *bestARAR = ev.arhr;
        *bestl1AR = opt.l1r;
        *bestl2AR = opt.l2r;
      }
    }
  }
  std::printf("\nDone.\n------------------------------------------------------------------\n");
  csr_free(model);
  model_free(dmodel);
  std::free(fmarker);
  matrix_free(mat);
  return rc;
}

int32_t Py_SLIM_GetTopN(slim_t* model, int32_t nratings, int32_t* itemids, float* ratings,
                        int32_t nrcmds, int32_t* rids, float* rscores, int32_t /*dbglvl*/) {
  return SLIM_GetTopN(model, nratings, itemids, ratings, nullptr, nrcmds, rids, rscores);
}

int32_t Py_SLIM_GetTopN_1vsk(slim_t* model, int32_t nratings, int32_t* itemids, float* ratings,
                             int32_t nrcmds, int32_t* rids, float* rscores, int32_t nnegs,
                             int32_t* negitems, int32_t /*dbglvl*/) {
  const slim_csr_t* W = as_csr(model);
  if (!W || !W->rowptr || nrcmds < 0 || nnegs < 0) return SLIM_ERROR;
  return top_n_1vsk(W, nratings, itemids, ratings, nrcmds, rids, rscores, nnegs, negitems);
}

// Where Py_SLIM_Predict scores: SLIM_PREDICT=gpu|cpu|auto (default auto: the GPU scorer when
// a device is present and nrcmds <= 128, else the host scorer; both give identical lists).
static int predict_policy() {
  const char* e = std::getenv("SLIM_PREDICT");
  if (e && std::strcmp(e, "cpu") == 0) return 0;
  if (e && std::strcmp(e, "gpu") == 0) return 2;
  return 1;
}

int32_t SLIMGPU_Predict(int32_t nrcmds, slim_t* slimhandle, slim_t* trnhandle, int32_t* output,
                        float* scores) {
  set_error("");
  return predict_device(as_csr(slimhandle), as_csr(trnhandle), nrcmds, output, scores, nullptr);
}

int32_t Py_SLIM_Predict(int32_t nrcmds, slim_t* slimhandle, slim_t* trnhandle, int32_t* output,
                        float* scores) {
  const slim_csr_t* W = as_csr(slimhandle);
  const slim_csr_t* trn = as_csr(trnhandle);
  if (!W || !trn || !W->rowptr || !trn->rowptr || nrcmds < 0) return SLIM_ERROR;
  const int policy = predict_policy();
  if (policy == 2 || (policy == 1 && nrcmds >= 1 && nrcmds <= 128 && trn->nrows > 0 &&
                      device_count() > 0)) {
    const int32_t rc = predict_device(W, trn, nrcmds, output, scores, nullptr);
    if (rc == SLIM_OK || policy == 2) return rc;
  }
  TopNScratch ws(W->ncols > W->nrows ? W->ncols : W->nrows);
  std::vector<int32_t> rids(nrcmds);
  std::vector<float> rsc(nrcmds);
  for (int32_t u = 0; u < trn->nrows; ++u) {
    const ssize_t lo = trn->rowptr[u], hi = trn->rowptr[u + 1];
    const int32_t n = top_n(W, (int32_t)(hi - lo), trn->rowind + lo,
                            trn->rowval ? trn->rowval + lo : nullptr, nrcmds, rids.data(),
                            rsc.data(), ws);
    for (int32_t r = 0; r < n; ++r) {
      output[(ssize_t)u * nrcmds + r] = rids[r];
      scores[(ssize_t)u * nrcmds + r] = rsc[r];
    }
  }
  return trn->nrows > 0 ? SLIM_OK : SLIM_ERROR;  // nvalid < 1 => error (pyapi.c:558-562)
}

int32_t Py_SLIM_Predict_1vsk(int32_t nrcmds, int32_t nnegs, slim_t* slimhandle,
                             slim_t* trnhandle, int32_t* negitems, int32_t* output,
                             float* scores) {
  const slim_csr_t* W = as_csr(slimhandle);
  const slim_csr_t* trn = as_csr(trnhandle);
  if (!W || !trn || !W->rowptr || !trn->rowptr || nrcmds < 0 || nnegs < 0) return SLIM_ERROR;
  const int policy = predict_policy();
  if (policy == 2 || (policy == 1 && nrcmds >= 1 && nnegs >= 1 && nnegs <= 1024 &&
                      trn->nrows > 0 && device_count() > 0)) {
    const int32_t rc = predict_1vsk_device(W, trn, nrcmds, nnegs, negitems, output, scores);
    if (rc == SLIM_OK || policy == 2) return rc;  // (unsorted model rows etc.: host scorer)
  }
  std::vector<int32_t> rids(nrcmds);
  std::vector<float> rsc(nrcmds);
  for (int32_t u = 0; u < trn->nrows; ++u) {
    const ssize_t lo = trn->rowptr[u], hi = trn->rowptr[u + 1];
    const int32_t n = top_n_1vsk(W, (int32_t)(hi - lo), trn->rowind + lo,
                                 trn->rowval ? trn->rowval + lo : nullptr, nrcmds, rids.data(),
                                 rsc.data(), nnegs, negitems + (ssize_t)u * nnegs);
    for (int32_t r = 0; r < n; ++r) {
      output[(ssize_t)u * nrcmds + r] = rids[r];
      scores[(ssize_t)u * nrcmds + r] = rsc[r];
    }
  }
  return trn->nrows > 0 ? SLIM_OK : SLIM_ERROR;
}

// ------------------------------------------------------------- SLIMGPU_* ------

slimgpu_matrix_t* SLIMGPU_MatrixFromHost(int32_t nrows, const ssize_t* rowptr,
                                         const int32_t* rowind, const float* rowval,
                                         int32_t* ioptions, int32_t* r_status) {
  set_error("");
  int32_t status = SLIM_ERROR;
  slimgpu_matrix_t* m =
      multi_from_host(nrows, rowptr, rowind, rowval, decode_options(ioptions, nullptr), &status);
  if (r_status) *r_status = status;
  return m;
}

slimgpu_matrix_t* SLIMGPU_MatrixFromDevice(int32_t nrows, int32_t ncols, const int64_t* d_rowptr,
                                           const int32_t* d_rowind, const float* d_rowval,
                                           int32_t* ioptions, int32_t* r_status) {
  set_error("");
  int32_t status = SLIM_ERROR;
  slimgpu_matrix_t* m = matrix_from_device(nrows, ncols, d_rowptr, d_rowind, d_rowval,
                                           decode_options(ioptions, nullptr), &status);
  if (r_status) *r_status = status;
  return m;
}

void SLIMGPU_MatrixFree(slimgpu_matrix_t** mat) {
  if (!mat) return;
  matrix_free(*mat);
  *mat = nullptr;
}

int32_t SLIMGPU_MatrixInfo(const slimgpu_matrix_t* mat, int32_t* nrows, int32_t* ncols,
                           int64_t* nnz) {
  return matrix_info(mat, nrows, ncols, nnz);
}

int32_t SLIMGPU_MatrixGetColumnView(const slimgpu_matrix_t* mat, int64_t* colptr,
                                    int32_t* colind, float* colval, float* cnorms) {
  return matrix_get_column_view(mat, colptr, colind, colval, cnorms);
}

int32_t SLIMGPU_MatrixColumnCost(const slimgpu_matrix_t* mat, int64_t* cost) {
  return matrix_column_cost(mat, cost);
}

void SLIMGPU_MatrixExpectSolves(slimgpu_matrix_t* mat, int32_t nsolves) {
  matrix_expect_solves(mat, nsolves);
}

int32_t SLIMGPU_MatrixDevice(const slimgpu_matrix_t* mat) { return matrix_device(mat); }

int32_t SLIMGPU_MatrixGramBuildRows(slimgpu_matrix_t* mat, int32_t row_begin, int32_t row_end) {
  set_error("");
  return gram_build_rows(mat, row_begin, row_end);
}

int32_t SLIMGPU_MatrixGramView(slimgpu_matrix_t* mat, void** dptr, int64_t* ld, int32_t* nrows) {
  set_error("");
  return gram_view(mat, dptr, ld, nrows);
}

int32_t SLIMGPU_MatrixGramCommit(slimgpu_matrix_t* mat) {
  set_error("");
  return gram_commit(mat);
}

slim_t* SLIMGPU_Learn(slimgpu_matrix_t* mat, int32_t* ioptions, double* doptions, slim_t* imodel,
                      int32_t* r_status) {
  set_error("");
  int32_t status = SLIM_ERROR;
  LearnOptions opt = decode_options(ioptions, doptions);
  slim_csr_t* model = nullptr;
  if (opt.algo != SLIM_ALGO_CD) {
    set_error("SLIMGPU_Learn: only algo=cd is implemented");
    status = SLIM_ERROR_INPUT;
  } else {
    model = multi_learn(mat, opt, as_csr(imodel), &status);
  }
  if (r_status) *r_status = status;
  return model;
}

slimgpu_model_t* SLIMGPU_LearnResident(slimgpu_matrix_t* mat, int32_t* ioptions, double* doptions,
                                       const slimgpu_model_t* warm, int32_t* r_status) {
  set_error("");
  int32_t status = SLIM_ERROR;
  LearnOptions opt = decode_options(ioptions, doptions);
  slimgpu_model_t* model = nullptr;
  if (opt.algo != SLIM_ALGO_CD) {
    set_error("SLIMGPU_LearnResident: only algo=cd is implemented");
    status = SLIM_ERROR_INPUT;
  } else {
    model = learn_resident(mat, opt, warm, &status);
  }
  if (r_status) *r_status = status;
  return model;
}

int64_t SLIMGPU_ModelNnz(const slimgpu_model_t* model) { return model_nnz(model); }

int32_t SLIMGPU_ModelFetchBegin(slimgpu_model_t* model) {
  set_error("");
  return model_fetch_begin(model);
}

slim_t* SLIMGPU_ModelFetch(slimgpu_model_t* model, int32_t* r_status) {
  set_error("");
  return model_fetch(model, r_status);
}

void SLIMGPU_ModelFree(slimgpu_model_t** model) {
  if (!model || !*model) return;
  model_free(*model);
  *model = nullptr;
}

int32_t SLIMGPU_ModelPredict(int32_t nrcmds, const slimgpu_model_t* model, slim_t* trnhandle,
                             int32_t* output, float* scores) {
  set_error("");
  DeviceRowView v;
  const int32_t rc = model_row_view(model, &v);
  if (rc != SLIM_OK) return rc;
  return predict_device_view(v, as_csr(trnhandle), nrcmds, output, scores, nullptr);
}

slim_t* SLIMGPU_LearnColumns(slimgpu_matrix_t* mat, int32_t ncolumns, const int32_t* columns,
                             int32_t* ioptions, double* doptions, slim_t* imodel,
                             int32_t* r_status) {
  set_error("");
  int32_t status = SLIM_ERROR;
  LearnOptions opt = decode_options(ioptions, doptions);
  slim_csr_t* model = nullptr;
  if (opt.algo != SLIM_ALGO_CD || ncolumns < 0 || (ncolumns > 0 && !columns)) {
    set_error("SLIMGPU_LearnColumns: algo must be cd and the column list non-null");
    status = SLIM_ERROR_INPUT;
  } else {
    static const int32_t none = 0;
    model = multi_learn(mat, opt, as_csr(imodel), &status, columns ? columns : &none, ncolumns);
  }
  if (r_status) *r_status = status;
  return model;
}

int32_t SLIMGPU_Predict1vsK(int32_t nrcmds, int32_t nnegs, slim_t* slimhandle, slim_t* trnhandle,
                            int32_t* negitems, int32_t* output, float* scores) {
  set_error("");
  return predict_1vsk_device(as_csr(slimhandle), as_csr(trnhandle), nrcmds, nnegs, negitems, output,
                             scores);
}

int32_t SLIMGPU_Evaluate(int32_t nusers, int32_t nrcmds, const int32_t* lists,
                         const int32_t* counts, slim_t* tsthandle, const int32_t* fmarker,
                         int32_t fm_ncols, double* metrics, int32_t* nvalid) {
  set_error("");
  if (!metrics || !nvalid) return SLIM_ERROR_INPUT;
  EvalResult ev;
  const int32_t rc =
      evaluate_device(nusers, nrcmds, lists, counts, as_csr(tsthandle), fmarker, fm_ncols, &ev);
  if (rc != SLIM_OK) return rc;
  metrics[0] = ev.hr; metrics[1] = ev.hr_head; metrics[2] = ev.hr_tail; metrics[3] = ev.arhr;
  nvalid[0] = ev.nvalid; nvalid[1] = ev.nvalid_head; nvalid[2] = ev.nvalid_tail;
  return SLIM_OK;
}

int32_t SLIMGPU_LastStats(slimgpu_stats_t* out) {
  if (!out) return SLIM_ERROR_INPUT;
  *out = last_stats();
  return SLIM_OK;
}

int32_t SLIMGPU_LastColumnStats(int32_t ncols, int32_t* nacols, int32_t* sweeps, int32_t* conv,
                                int64_t* G, int64_t* D, int64_t* U) {
  const ColumnStats& cs = last_column_stats();
  if ((size_t)ncols > cs.nacols.size()) return SLIM_ERROR_INPUT;
  const size_t n = (size_t)ncols;
  if (nacols) std::memcpy(nacols, cs.nacols.data(), sizeof(int32_t) * n);
  if (sweeps) std::memcpy(sweeps, cs.sweeps.data(), sizeof(int32_t) * n);
  if (conv) std::memcpy(conv, cs.conv.data(), sizeof(int32_t) * n);
  if (G) std::memcpy(G, cs.G.data(), sizeof(int64_t) * n);
  if (D) std::memcpy(D, cs.D.data(), sizeof(int64_t) * n);
  if (U) std::memcpy(U, cs.U.data(), sizeof(int64_t) * n);
  return SLIM_OK;
}

int32_t SLIMGPU_DeviceCount(void) { return device_count(); }

const char* SLIMGPU_LastError(void) { return last_error(); }

}  // extern "C"
