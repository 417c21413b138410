// slim_predict -- top-N recommendations and, given hidden items, HR / ARHR.
// Usage and options as /root/reference/src/programs/slim_predict.c and
// cmdline_predict.c:15-20 (slim_predict [options] model-file old-file [test-file] [neg-file]).
#include <random>

#include "cli_common.hpp"
using namespace slimcli;

int main(int argc, char** argv) {
  const std::vector<OptSpec> specs = {{"ifmt", true},   {"binarize", false}, {"outfile", true},
                                      {"nrcmds", true}, {"dbglvl", true},    {"help", false}};
  Args a = parse_args(argc, argv, specs);
  if (a.has("help") || a.pos.size() < 2 || a.pos.size() > 4) {
    std::printf("\n Usage: slim_predict [options] model-file old-file [test-file] [neg-file]\n"
                "   -ifmt=csr|csrnv|cluto|ijv  -binarize  -outfile=file  -nrcmds=i  -dbglvl=i\n\n");
    return 0;
  }
  const Fmt fmt = parse_fmt(a.str("ifmt", "csr"));
  const int nrcmds = a.integer("nrcmds", 10);
  if (nrcmds < 1) die("The -nrcmds parameter should be positive.");
  for (const auto& p : a.pos)
    if (!file_exists(p)) die("Input file " + p + " does not exist.");

  slim_t* model = read_model(a.pos[0], fmt == Fmt::csrnv ? Fmt::csr : fmt);
  const slim_csr_t* W = static_cast<slim_csr_t*>(model);
  Csr old = read_matrix(a.pos[1], fmt);
  Csr tst, neg;
  const bool has_tst = a.pos.size() > 2, has_neg = a.pos.size() > 3;
  if (has_tst) tst = read_matrix(a.pos[2], fmt);
  if (has_neg) neg = read_matrix(a.pos[3], fmt);
  banner();
  std::printf("  mdlfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[0].c_str(), W->nrows, W->ncols,
              W->rowptr[W->nrows]);
  std::printf("  oldfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[1].c_str(), old.nrows, old.ncols, old.nnz());
  if (has_tst) std::printf("  tstfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[2].c_str(), tst.nrows, tst.ncols, tst.nnz());
  if (has_neg) std::printf("  negfile: %s, nrows: %d, ncols: %d, nnz: %zd\n", a.pos[3].c_str(), neg.nrows, neg.ncols, neg.nnz());
  std::printf("  binarize: %d, nrcmds: %d\n\nMaking predictions...\n", (int)a.has("binarize"), nrcmds);
  if (has_tst && old.nrows != tst.nrows) die("The number of rows in the old and test files do not match.");
  if (a.has("binarize")) old.has_val = false;

  const int32_t nusers = old.nrows;
  const int32_t ncols = std::max({W->ncols, W->nrows, old.ncols, tst.ncols, neg.ncols});
  std::vector<int32_t> lists((size_t)nusers * nrcmds, -1), lens(nusers, 0);
  std::vector<float> scores((size_t)nusers * nrcmds, 0.0f);
  if (!has_neg) {
    // all users at once (GPU scorer when a device is present, identical lists otherwise)
    slim_t* hold = to_handle(old);
    if (Py_SLIM_Predict(nrcmds, model, hold, lists.data(), scores.data()) != SLIM_OK)
      die(std::string("prediction failed: ") + SLIMGPU_LastError());
    Py_csr_free(hold);
    for (int32_t u = 0; u < nusers; ++u)
      while (lens[u] < nrcmds && lists[(size_t)u * nrcmds + lens[u]] >= 0) ++lens[u];
  } else {
    // 1-vs-k protocol (slim_predict.c:110-165): score everything, keep the hidden and the
    // negative items, shuffle (ties), sort, keep nrcmds
    std::mt19937 rng(1);
    const int32_t ask = std::max(W->nrows, 1);
    std::vector<int32_t> rids(ask);
    std::vector<float> rsc(ask);
    for (int32_t u = 0; u < nusers; ++u) {
      const ssize_t h0 = old.ptr[u], h1 = old.ptr[u + 1];
      const int32_t n = SLIM_GetTopN(model, (int32_t)(h1 - h0), old.ind.data() + h0,
                                     old.has_val ? old.val.data() + h0 : nullptr, nullptr, ask,
                                     rids.data(), rsc.data());
      if (n < 0) continue;
      std::map<int32_t, float> cand;
      for (ssize_t z = tst.ptr[u]; z < tst.ptr[u + 1]; ++z) cand[tst.ind[z]] = 0.0f;
      if (u < neg.nrows)
        for (ssize_t z = neg.ptr[u]; z < neg.ptr[u + 1]; ++z) cand[neg.ind[z]] = 0.0f;
      for (int32_t r = 0; r < n; ++r) {
        auto it = cand.find(rids[r]);
        if (it != cand.end()) it->second = rsc[r];
      }
      std::vector<std::pair<float, int32_t>> v;
      for (auto& c : cand) v.emplace_back(c.second, c.first);
      std::shuffle(v.begin(), v.end(), rng);
      std::stable_sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
      lens[u] = (int32_t)std::min<size_t>(v.size(), nrcmds);
      for (int32_t r = 0; r < lens[u]; ++r) {
        lists[(size_t)u * nrcmds + r] = v[r].second;
        scores[(size_t)u * nrcmds + r] = v[r].first;
      }
    }
  }
  if (a.has("outfile")) {
    FILE* f = std::fopen(a.str("outfile", "").c_str(), "w");
    if (!f) die("cannot open the output file");
    for (int32_t u = 0; u < nusers; ++u) {
      for (int32_t r = 0; r < lens[u]; ++r)
        std::fprintf(f, " %d %f", lists[(size_t)u * nrcmds + r], scores[(size_t)u * nrcmds + r]);
      std::fputc('\n', f);
    }
    std::fclose(f);
  }
  if (has_tst) {
    int32_t* fmarker = SLIM_DetermineHeadAndTail(old.nrows, ncols, old.ptr.data(), old.ind.data());
    const Eval e = evaluate_lists(tst, lists, lens, nrcmds, fmarker, ncols);
    std::free(fmarker);
    std::printf("\nnvalid: %d nvalid_head: %d nvalid_tail: %d", e.nvalid, e.nvalid_head, e.nvalid_tail);
    std::printf("\nhr: %.4f hr_head: %.4f hr_tail: %.4f arhr: %.4f\n", e.hr, e.hr_head, e.hr_tail, e.arhr);
  }
  std::printf("------------------------------------------------------------------\n");
  SLIM_FreeModel(&model);
  return 0;
}
