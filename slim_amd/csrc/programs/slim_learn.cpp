// slim_learn -- estimate a SLIM / fSLIM model from a rating matrix file.
// Options, defaults and output as /root/reference/src/programs/slim_learn.c and
// cmdline_learn.c:19-33,144-161 (usage: slim_learn [options] train-file [model-file]).
#include "cli_common.hpp"
using namespace slimcli;

int main(int argc, char** argv) {
  const std::vector<OptSpec> specs = {
      {"ifmt", true},    {"binarize", false}, {"l1r", true},     {"l2r", true},
      {"optTol", true},  {"niters", true},    {"nnbrs", true},   {"simtype", true},
      {"algo", true},    {"ordered", false},  {"nthreads", true}, {"ipmdlfile", true},
      {"dbglvl", true},  {"ngpus", true},   {"help", false}};
  Args a = parse_args(argc, argv, specs);
  if (a.has("help") || a.pos.empty() || a.pos.size() > 2) {
    std::printf("\n Usage: slim_learn [options] train-file [model-file]\n"
                "   -ifmt=csr|csrnv|cluto|ijv  -binarize  -l1r=f  -l2r=f  -optTol=f  -niters=i\n"
                "   -nnbrs=i  -simtype=cos|jac|dotp  -algo=cd  -nthreads=i  -ipmdlfile=file  -dbglvl=i\n"
                "   -ngpus=i   (engine extension: shard the item columns over i GPUs of this node)\n\n");
    return 0;
  }
  const Fmt fmt = parse_fmt(a.str("ifmt", "csr"));
  const std::string trnfile = a.pos[0], mdlfile = a.pos.size() > 1 ? a.pos[1] : "slim.model";
  if (!file_exists(trnfile)) die("Input training file " + trnfile + " does not exist.");
  const double l1r = a.num("l1r", 1.0), l2r = a.num("l2r", 1.0), optTol = a.num("optTol", 1e-7);
  const int niters = a.integer("niters", 10000), nnbrs = a.integer("nnbrs", 0);
  const int dbglvl = a.integer("dbglvl", SLIM_DBG_INFO | SLIM_DBG_TIME);
  if (l1r < 0 || l2r < 0 || optTol < 0 || niters < 0 || nnbrs < 0 || dbglvl < 0)
    die("The -l1r, -l2r, -optTol, -niters, -nnbrs and -dbglvl parameters should be non-negative.");
  const std::string sim = a.str("simtype", "cos"), algo = a.str("algo", "cd");
  const int simtype = sim == "cos" ? SLIM_SIMTYPE_COS : sim == "jac" ? SLIM_SIMTYPE_JAC
                      : sim == "dotp" ? SLIM_SIMTYPE_DOTP : -1;
  if (simtype < 0) die("Invalid -simtype of " + sim + ".");
  if (algo != "cd" && algo != "admm") die("Invalid -algo of " + algo + ".");

  Csr trn = read_matrix(trnfile, fmt);
  banner();
  std::printf("  trnfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", trnfile.c_str(), trn.nrows,
              trn.ncols, trn.nnz());
  std::printf("  l1r: %.2le, l2r: %.2le, binarize: %s\n", l1r, l2r, a.has("binarize") ? "Yes" : "No");
  std::printf("  solver: %s, optTol: %.2le, niters: %d\n", algo.c_str(), optTol, niters);
  std::printf("  mdlfile: %s, nthreads: %d, dbglvl: %d\n", mdlfile.c_str(), a.integer("nthreads", 1), dbglvl);
  std::printf("  simtype: %s, nnbrs: %d\n", sim.c_str(), nnbrs);
  std::printf("\nEstimating model...\n");
  if (a.has("binarize")) trn.has_val = false;  // slim_learn.c:47-48

  slim_t* imodel = nullptr;
  if (a.has("ipmdlfile")) imodel = read_model(a.str("ipmdlfile", ""), Fmt::csr);  // :51-57

  int32_t io[SLIM_NOPTIONS];
  double dopt[SLIM_NOPTIONS];
  SLIM_iSetDefaults(io);
  SLIM_dSetDefaults(dopt);
  io[SLIM_OPTION_DBGLVL] = dbglvl;
  io[SLIM_OPTION_NNBRS] = nnbrs;
  io[SLIM_OPTION_SIMTYPE] = simtype;
  io[SLIM_OPTION_ALGO] = algo == "cd" ? SLIM_ALGO_CD : SLIM_ALGO_ADMM;
  io[SLIM_OPTION_NTHREADS] = a.integer("nthreads", 1);
  io[SLIM_OPTION_MAXNITERS] = niters;
  if (a.has("ngpus")) io[SLIM_OPTION_GPU_NGPUS] = a.integer("ngpus", 1);
  dopt[SLIM_OPTION_L1R] = l1r;
  dopt[SLIM_OPTION_L2R] = l2r;
  dopt[SLIM_OPTION_OPTTOL] = optTol;

  int32_t status = SLIM_ERROR;
  slim_t* model = SLIM_Learn(trn.nrows, trn.ptr.data(), trn.ind.data(), trn.valptr(), io, dopt,
                             imodel, &status);
  int rc = 0;
  if (status != SLIM_OK || !model) {
    std::printf("ERROR: Something went wrong with model estimation: rstatus: %d [%s]\n", status,
                SLIMGPU_LastError());
    rc = 1;
  } else {
    write_matrix(static_cast<slim_csr_t*>(model), mdlfile, fmt);
  }
  std::printf("\nDone.\n------------------------------------------------------------------\n");
  SLIM_FreeModel(&model);
  SLIM_FreeModel(&imodel);
  return rc;
}
