// slim_mselect -- model selection over the (l1, l2) pairs of an l12 file, warm-starting
// every model from the previous one; R is staged in HBM once for the whole grid.
// Usage and options as /root/reference/src/programs/slim_mselect.c and cmdline_mselect.c
// (slim_mselect [options] train-file test-file l12-file).
#include "cli_common.hpp"
using namespace slimcli;

int main(int argc, char** argv) {
  const std::vector<OptSpec> specs = {
      {"ifmt", true},    {"binarize", false}, {"optTol", true},  {"niters", true},
      {"nnbrs", true},   {"simtype", true},   {"algo", true},    {"nthreads", true},
      {"nrcmds", true},  {"dbglvl", true},    {"nomodels", false}, {"ngpus", true},
      {"help", false}};
  Args a = parse_args(argc, argv, specs);
  if (a.has("help") || a.pos.size() != 3) {
    std::printf("\n Usage: slim_mselect [options] train-file test-file l12-file\n"
                "   -ifmt=csr|csrnv|cluto|ijv  -binarize  -optTol=f  -niters=i  -nnbrs=i  -simtype=s\n"
                "   -nrcmds=i  -nthreads=i  -dbglvl=i  -nomodels (do not write '<l1 l2>.model' files)\n"
                "   -ngpus=i   (engine extension: R replicated on i GPUs, every model sharded over them)\n\n");
    return 0;
  }
  const Fmt fmt = parse_fmt(a.str("ifmt", "csr"));
  for (const auto& p : a.pos)
    if (!file_exists(p)) die("Input file " + p + " does not exist.");
  const int nrcmds = a.integer("nrcmds", 10);
  Csr trn = read_matrix(a.pos[0], fmt), tst = read_matrix(a.pos[1], fmt);
  if (a.has("binarize")) trn.has_val = false;
  const std::string sim = a.str("simtype", "cos");
  banner();
  std::printf("  trnfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[0].c_str(), trn.nrows, trn.ncols, trn.nnz());
  std::printf("  tstfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[1].c_str(), tst.nrows, tst.ncols, tst.nnz());
  std::printf("  l12file: %s\n\nEstimating & evaluating models...\n\n", a.pos[2].c_str());

  int32_t io[SLIM_NOPTIONS];
  double dopt[SLIM_NOPTIONS];
  SLIM_iSetDefaults(io);
  SLIM_dSetDefaults(dopt);
  io[SLIM_OPTION_DBGLVL] = a.integer("dbglvl", 0);
  io[SLIM_OPTION_NNBRS] = a.integer("nnbrs", 0);
  io[SLIM_OPTION_SIMTYPE] = sim == "jac" ? SLIM_SIMTYPE_JAC : sim == "dotp" ? SLIM_SIMTYPE_DOTP : SLIM_SIMTYPE_COS;
  io[SLIM_OPTION_ALGO] = SLIM_ALGO_CD;
  io[SLIM_OPTION_NTHREADS] = a.integer("nthreads", 1);
  io[SLIM_OPTION_MAXNITERS] = a.integer("niters", 10000);
  dopt[SLIM_OPTION_OPTTOL] = a.num("optTol", 1e-7);
  if (a.has("ngpus")) io[SLIM_OPTION_GPU_NGPUS] = a.integer("ngpus", 1);

  int32_t status = SLIM_ERROR;
  slimgpu_matrix_t* R = SLIMGPU_MatrixFromHost(trn.nrows, trn.ptr.data(), trn.ind.data(),
                                               trn.valptr(), io, &status);
  if (!R) die(std::string("cannot stage the training matrix: ") + SLIMGPU_LastError());
  slim_t* hold = to_handle(trn);
  const int32_t ncols = std::max(trn.ncols, tst.ncols);
  int32_t* fmarker = SLIM_DetermineHeadAndTail(trn.nrows, ncols, trn.ptr.data(), trn.ind.data());

  FILE* lf = std::fopen(a.pos[2].c_str(), "r");
  char line[256];
  {  // one R, as many solves as the file has pairs: let the engine plan for that
    int32_t npairs = 0;
    double l1, l2;
    while (std::fgets(line, sizeof line, lf))
      if (std::sscanf(line, "%lf %lf", &l1, &l2) == 2) ++npairs;
    std::rewind(lf);
    SLIMGPU_MatrixExpectSolves(R, npairs);
  }
  slim_t* model = nullptr;
  slimgpu_model_t* dmodel = nullptr;
  const char* ng_env = std::getenv("SLIM_GPU_NGPUS");
  const char* res_env = std::getenv("SLIM_GPU_RESIDENT");
  const bool resident = !(a.has("ngpus") && a.integer("ngpus", 1) > 1) && !(ng_env && std::atoi(ng_env) > 1) &&
                        !(res_env && std::atoi(res_env) == 0);
  double best_hr = 0, best_ar = 0, bh_l1 = 0, bh_l2 = 0, ba_l1 = 0, ba_l2 = 0;
  while (std::fgets(line, sizeof line, lf)) {
    double l1, l2;
    if (std::sscanf(line, "%lf %lf", &l1, &l2) != 2) continue;  // slim_mselect.c:100-101
    dopt[SLIM_OPTION_L1R] = l1;
    dopt[SLIM_OPTION_L2R] = l2;
    // the model stays in HBM (SLIMGPU_LearnResident): warm start without an upload, scored where it
    // lies; with model files wanted it is fetched beside nothing -- the write needs it at once
    // (a model sharded over several GPUs is assembled on the host: SLIMGPU_Learn as before)
    if (!resident) {
      slim_t* next = SLIMGPU_Learn(R, io, dopt, model, &status);  // warm start, :103-113
      SLIM_FreeModel(&model);
      model = next;
    } else {
      slimgpu_model_t* dnext = SLIMGPU_LearnResident(R, io, dopt, dmodel, &status);  // warm start, :103-113
      SLIMGPU_ModelFree(&dmodel);
      dmodel = dnext;
      SLIM_FreeModel(&model);
    }
    if (resident ? !dmodel : !model) {
      std::printf("ERROR: model estimation failed [%.3le %.3le]: rstatus %d\n", l1, l2, status);
      continue;
    }
    slimgpu_stats_t st;
    SLIMGPU_LastStats(&st);
    if (resident && !a.has("nomodels")) {
      model = SLIMGPU_ModelFetch(dmodel, &status);
      if (!model) {
        std::printf("ERROR: model fetch failed [%.3le %.3le]: rstatus %d\n", l1, l2, status);
        continue;
      }
    }
    if (!a.has("nomodels")) {
      std::string name(line);
      while (!name.empty() && (name.back() == '\n' || name.back() == '\r')) name.pop_back();
      write_matrix(static_cast<slim_csr_t*>(model), name + ".model", fmt == Fmt::csrnv ? Fmt::csr : fmt);
    }
    std::vector<int32_t> lists((size_t)trn.nrows * nrcmds, -1), lens(trn.nrows, 0);
    std::vector<float> scores((size_t)trn.nrows * nrcmds, 0.0f);
    if (!resident) {
      if (Py_SLIM_Predict(nrcmds, model, hold, lists.data(), scores.data()) != SLIM_OK) continue;
    } else if (SLIMGPU_ModelPredict(nrcmds, dmodel, hold, lists.data(), scores.data()) != SLIM_OK) {
      if (!model) model = SLIMGPU_ModelFetch(dmodel, &status);  // (lists beyond the GPU scorer's 128: host loop)
      if (!model || Py_SLIM_Predict(nrcmds, model, hold, lists.data(), scores.data()) != SLIM_OK) continue;
    }
    for (int32_t u = 0; u < trn.nrows; ++u)
      while (lens[u] < nrcmds && lists[(size_t)u * nrcmds + lens[u]] >= 0) ++lens[u];
    const Eval e = evaluate_lists(tst, lists, lens, nrcmds, fmarker, ncols);
    std::printf("l1r: %.2le l2r: %.2le nnz: %7zd hr: %.4f hr_head: %.4f hr_tail: %.4f arhr: %.4f time: %.2lf\n",
                l1, l2, resident ? (ssize_t)SLIMGPU_ModelNnz(dmodel) : static_cast<slim_csr_t*>(model)->rowptr[static_cast<slim_csr_t*>(model)->nrows], e.hr, e.hr_head, e.hr_tail, e.arhr, st.total_ms / 1e3);
    if (e.hr > best_hr) { best_hr = e.hr; bh_l1 = l1; bh_l2 = l2; }
    if (e.arhr > best_ar) { best_ar = e.arhr; ba_l1 = l1; ba_l2 = l2; }
  }
  std::fclose(lf);
  std::printf("\nbest hr: %.4f at l1 %.4g l2 %.4g; best arhr: %.4f at l1 %.4g l2 %.4g\n", best_hr, bh_l1,
              bh_l2, best_ar, ba_l1, ba_l2);
  std::printf("\nDone.\n------------------------------------------------------------------\n");
  SLIM_FreeModel(&model);
  SLIMGPU_ModelFree(&dmodel);
  std::free(fmarker);
  Py_csr_free(hold);
  SLIMGPU_MatrixFree(&R);
  return 0;
}
