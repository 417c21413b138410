// host_csr.cpp -- see host_csr.hpp.
#include "host_csr.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

namespace slimamd {

namespace {
thread_local std::string g_err;

template <class T>
T* xmalloc(size_t n) {
  return static_cast<T*>(std::malloc(sizeof(T) * (n ? n : 1)));
}
}  // namespace

void set_error(const std::string& msg) {
  g_err = msg;
  if (!msg.empty()) std::fprintf(stderr, "[libslim/gfx950] %s\n", msg.c_str());
}
const char* last_error() { return g_err.c_str(); }

slim_csr_t* csr_new() {
  auto* m = static_cast<slim_csr_t*>(std::calloc(1, sizeof(slim_csr_t)));
  if (m) m->nrows = m->ncols = -1;
  return m;
}

void csr_free(slim_csr_t* m) {
  if (!m) return;
  void* owned[] = {m->rowptr, m->colptr,  m->rowind,  m->colind, m->rowids, m->colids,
                   m->rlabels, m->clabels, m->rmap,   m->cmap,   m->rowval, m->colval,
                   m->rnorms,  m->cnorms,  m->rsums,  m->csums,  m->rsizes, m->csizes,
                   m->rvols,   m->cvols,   m->rwgts,  m->cwgts};
  for (void* p : owned) std::free(p);
  std::free(m);
}

int32_t max_index_plus_one(int64_t nnz, const int32_t* ind) {
  int32_t hi = -1;
  for (int64_t k = 0; k < nnz; ++k) hi = std::max(hi, ind[k]);
  return hi + 1;
}

slim_csr_t* csr_from_rows(int32_t nrows, const ssize_t* ptr, const int32_t* ind,
                          const float* val) {
  slim_csr_t* m = csr_new();
  if (!m) return nullptr;
  const int64_t nnz = ptr[nrows];
  m->nrows = nrows;
  m->ncols = max_index_plus_one(nnz, ind);
  m->rowptr = xmalloc<ssize_t>(nrows + 1);
  m->rowind = xmalloc<int32_t>(nnz);
  m->rowval = val ? xmalloc<float>(nnz) : nullptr;
  if (!m->rowptr || !m->rowind || (val && !m->rowval)) {
    csr_free(m);
    return nullptr;
  }
  std::memcpy(m->rowptr, ptr, sizeof(ssize_t) * (nrows + 1));
  std::memcpy(m->rowind, ind, sizeof(int32_t) * nnz);
  if (val) std::memcpy(m->rowval, val, sizeof(float) * nnz);
  return m;
}

void csr_build_index(slim_csr_t* m, int what) {
  // source view -> destination view
  const int32_t nsrc = what == 0 ? m->nrows : m->ncols;
  const int32_t ndst = what == 0 ? m->ncols : m->nrows;
  const ssize_t* sp = what == 0 ? m->rowptr : m->colptr;
  const int32_t* si = what == 0 ? m->rowind : m->colind;
  const float* sv = what == 0 ? m->rowval : m->colval;
  const int64_t nnz = sp[nsrc];

  ssize_t* dp = xmalloc<ssize_t>(ndst + 1);
  int32_t* di = xmalloc<int32_t>(nnz);
  float* dv = sv ? xmalloc<float>(nnz) : nullptr;
  std::fill(dp, dp + ndst + 1, 0);
  // Large models (a C5 grid step returns 77M entries 45 times over): the counting-sort transpose
  // on several host threads -- sources cut into T ranges of equal nnz, one histogram per range,
  // every range scatters from its own start offsets.  Within a destination row the sources stay
  // ascending (ranges ascending, each walked in order): the same arrays as the serial form.
  const unsigned hw = std::thread::hardware_concurrency();
  const int T = nnz >= (int64_t(1) << 22) ? (int)std::min<unsigned>(16u, std::max(1u, hw)) : 1;
  if (T > 1) {
    std::vector<int32_t> cut((size_t)T + 1, nsrc);
    cut[0] = 0;
    for (int t = 1; t < T; ++t)
      cut[(size_t)t] = (int32_t)(std::lower_bound(sp, sp + nsrc + 1, (ssize_t)(nnz / T * t)) - sp);
    std::vector<std::vector<ssize_t>> hist((size_t)T, std::vector<ssize_t>((size_t)ndst, 0));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        ssize_t* h = hist[(size_t)t].data();
        for (ssize_t k = sp[cut[(size_t)t]]; k < sp[cut[(size_t)t + 1]]; ++k) ++h[si[k]];
      });
    for (auto& x : th) x.join();
    th.clear();
    ssize_t run = 0;
    for (int32_t d = 0; d < ndst; ++d) {  // hist[t][d] becomes range t's first slot in row d
      dp[d] = run;
      for (int t = 0; t < T; ++t) {
        const ssize_t c = hist[(size_t)t][(size_t)d];
        hist[(size_t)t][(size_t)d] = run;
        run += c;
      }
    }
    dp[ndst] = run;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        ssize_t* fill = hist[(size_t)t].data();
        for (int32_t s = cut[(size_t)t]; s < cut[(size_t)t + 1]; ++s)
          for (ssize_t k = sp[s]; k < sp[s + 1]; ++k) {
            const ssize_t slot = fill[si[k]]++;
            di[slot] = s;
            if (dv) dv[slot] = sv[k];
          }
      });
    for (auto& x : th) x.join();
  } else {
    for (int64_t k = 0; k < nnz; ++k) ++dp[si[k] + 1];
    std::partial_sum(dp, dp + ndst + 1, dp);
    std::vector<ssize_t> fill(dp, dp + ndst);
    for (int32_t s = 0; s < nsrc; ++s)
      for (ssize_t k = sp[s]; k < sp[s + 1]; ++k) {
        const ssize_t slot = fill[si[k]]++;
        di[slot] = s;
        if (dv) dv[slot] = sv[k];
      }
  }
  if (what == 0) {
    std::free(m->colptr); std::free(m->colind); std::free(m->colval);
    m->colptr = dp; m->colind = di; m->colval = dv;
  } else {
    std::free(m->rowptr); std::free(m->rowind); std::free(m->rowval);
    m->rowptr = dp; m->rowind = di; m->rowval = dv;
  }
}

slim_csr_t* model_from_columns(int32_t n, ssize_t* colptr, int32_t* colind,
                               float* colval, bool row_view) {
  slim_csr_t* m = csr_new();
  if (!m) return nullptr;
  m->nrows = m->ncols = n;
  m->colptr = colptr;
  m->colind = colind;
  m->colval = colval;
  if (row_view) csr_build_index(m, 1);
  return m;
}

// ---------------------------------------------------------------------------
// top-N (reference predict.c)
// ---------------------------------------------------------------------------
namespace {
// descending by score; equal scores keep their discovery order (the
// reference's gk_fkvsortd leaves tie order undefined)
int32_t emit_best(int32_t ncand, int32_t nrcmds, const std::vector<float>& key,
                  const std::vector<int32_t>& val, int32_t* rids, float* rscores) {
  std::vector<int32_t> order(ncand);
  std::iota(order.begin(), order.end(), 0);
  const int32_t n = std::min(ncand, nrcmds);
  auto better = [&](int32_t a, int32_t b) {
    return key[a] > key[b] || (key[a] == key[b] && a < b);
  };
  std::partial_sort(order.begin(), order.begin() + n, order.end(), better);
  for (int32_t r = 0; r < n; ++r) {
    rids[r] = val[order[r]];
    rscores[r] = key[order[r]];
  }
  return n;
}
}  // namespace

int32_t top_n(const slim_csr_t* W, int32_t nratings, const int32_t* itemids,
              const float* ratings, int32_t nrcmds, int32_t* rids, float* rscores,
              TopNScratch& ws) {
  const int32_t ncols = W->ncols, nrows = W->nrows;
  auto in_cols = [&](int32_t i) { return i >= 0 && i < ncols; };
  for (int32_t r = 0; r < nratings; ++r)
    if (in_cols(itemids[r])) ws.marker[itemids[r]] = -2;  // history: never recommended

  int32_t ncand = 0;
  for (int32_t r = 0; r < nratings; ++r) {
    const int32_t i = itemids[r];
    if (i < 0 || i >= nrows) continue;  // (the reference's guard cannot fire; ids
                                        //  outside the model are undefined there)
    const float rating = ratings ? ratings[r] : 1.0f;
    for (ssize_t j = W->rowptr[i]; j < W->rowptr[i + 1]; ++j) {
      const int32_t k = W->rowind[j];
      int32_t& slot = ws.marker[k];
      if (slot == -2) continue;
      if (slot == -1) {
        ws.val[ncand] = k;
        ws.key[ncand] = 0.0f;
        slot = ncand++;
      }
      ws.key[slot] += rating * W->rowval[j];
    }
  }
  const int32_t n = emit_best(ncand, nrcmds, ws.key, ws.val, rids, rscores);
  for (int32_t c = 0; c < ncand; ++c) ws.marker[ws.val[c]] = -1;
  for (int32_t r = 0; r < nratings; ++r)
    if (in_cols(itemids[r])) ws.marker[itemids[r]] = -1;
  return n;
}

int32_t top_n_1vsk(const slim_csr_t* W, int32_t nratings, const int32_t* itemids,
                   const float* ratings, int32_t nrcmds, int32_t* rids,
                   float* rscores, int32_t nnegs, const int32_t* negitems) {
  const int32_t ncols = W->ncols, nrows = W->nrows;
  std::vector<int32_t> slot_of(ncols, -2);  // -2: not a candidate
  std::vector<float> key(nnegs, 0.0f);
  std::vector<int32_t> val(negitems, negitems + nnegs);
  for (int32_t c = 0; c < nnegs; ++c)
    if (negitems[c] >= 0 && negitems[c] < ncols) slot_of[negitems[c]] = c;
  for (int32_t r = 0; r < nratings; ++r) {
    const int32_t i = itemids[r];
    if (i < 0 || i >= nrows) continue;
    const float rating = ratings ? ratings[r] : 1.0f;
    for (ssize_t j = W->rowptr[i]; j < W->rowptr[i + 1]; ++j) {
      const int32_t s = slot_of[W->rowind[j]];
      if (s >= 0) key[s] += rating * W->rowval[j];
    }
  }
  return emit_best(nnegs, nrcmds, key, val, rids, rscores);
}

int32_t* head_tail_split(int32_t nrows, int32_t ncols, const ssize_t* rowptr,
                         const int32_t* rowind) {
  int32_t* mark = xmalloc<int32_t>(ncols);
  std::vector<int64_t> pop(ncols, 0);
  for (ssize_t k = 0; k < rowptr[nrows]; ++k)
    if (rowind[k] >= 0 && rowind[k] < ncols) ++pop[rowind[k]];
  std::vector<int32_t> by_pop(ncols);
  std::iota(by_pop.begin(), by_pop.end(), 0);
  std::stable_sort(by_pop.begin(), by_pop.end(),
                   [&](int32_t a, int32_t b) { return pop[a] > pop[b]; });
  std::fill(mark, mark + ncols, 1);
  int64_t budget = rowptr[nrows] / 2;
  for (int32_t c = 0; c < ncols && budget > 0; ++c) {
    mark[by_pop[c]] = 0;
    budget -= pop[by_pop[c]];
  }
  return mark;
}

EvalResult evaluate(const slim_csr_t* model, const slim_csr_t* trn,
                    const slim_csr_t* tst, int32_t nrcmds, const int32_t* fmarker,
                    int32_t fm_ncols, const int32_t* lists, const int32_t* counts) {
  EvalResult out;
  TopNScratch ws(std::max(model->ncols, model->nrows));
  std::vector<int32_t> rids(nrcmds);
  std::vector<float> rsc(nrcmds);
  std::vector<int32_t> wanted(std::max(fm_ncols, model->ncols), -1);
  // accumulators are float in the reference (pyapi.c:223-230)
  float hr_all = 0, hr_head = 0, hr_tail = 0, arhr = 0;
  const int32_t nusers = std::min(trn->nrows, tst->nrows);
  for (int32_t u = 0; u < nusers; ++u) {
    const ssize_t t0 = tst->rowptr[u], t1 = tst->rowptr[u + 1];
    if (t1 - t0 < 1) continue;
    const ssize_t h0 = trn->rowptr[u], h1 = trn->rowptr[u + 1];
    int32_t n;
    if (lists) {
      n = counts[u];
      std::copy(lists + (size_t)u * nrcmds, lists + (size_t)u * nrcmds + n, rids.begin());
    } else {
      n = top_n(model, int32_t(h1 - h0), trn->rowind + h0,
                trn->rowval ? trn->rowval + h0 : nullptr, nrcmds, rids.data(), rsc.data(), ws);
    }
    ++out.nvalid;
    int32_t ntrue[2] = {0, 0}, nhits[3] = {0, 0, 0};
    bool has_head = false, has_tail = false;
    float gain = 0, ideal = 0;
    for (ssize_t z = t0; z < t1; ++z) {
      const int32_t it = tst->rowind[z];
      wanted[it] = u;
      ++ntrue[fmarker[it]];
      (fmarker[it] ? has_tail : has_head) = true;
      ideal += 1.0 / (1.0 + double(z - t0));
    }
    out.nvalid_head += has_head;
    out.nvalid_tail += has_tail;
    for (int32_t r = 0; r < n; ++r)
      if (wanted[rids[r]] == u) {
        ++nhits[fmarker[rids[r]]];
        ++nhits[2];
        gain += 1.0 / (1.0 + r);
      }
    hr_head += nhits[0] > 0 ? 1.0 * nhits[0] / ntrue[0] : 0.0;
    hr_tail += nhits[1] > 0 ? 1.0 * nhits[1] / ntrue[1] : 0.0;
    hr_all += 1.0 * nhits[2] / double(t1 - t0);
    arhr += gain / ideal;
  }
  out.hr = out.nvalid > 0 ? hr_all / out.nvalid : 0;
  out.hr_head = out.nvalid_head > 0 ? hr_head / out.nvalid_head : 0;
  out.hr_tail = out.nvalid_tail > 0 ? hr_tail / out.nvalid_tail : 0;
  out.arhr = out.nvalid > 0 ? arhr / out.nvalid : 0;
  return out;
}

// ---------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------
bool write_binrow(const slim_csr_t* m, const char* path) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return false;
  const int64_t nnz = m->rowptr[m->nrows];
  bool ok = std::fwrite(&m->nrows, sizeof(int32_t), 1, f) == 1 &&
            std::fwrite(&m->ncols, sizeof(int32_t), 1, f) == 1 &&
            std::fwrite(m->rowptr, sizeof(ssize_t), m->nrows + 1, f) == size_t(m->nrows + 1) &&
            std::fwrite(m->rowind, sizeof(int32_t), nnz, f) == size_t(nnz) &&
            (!m->rowval || std::fwrite(m->rowval, sizeof(float), nnz, f) == size_t(nnz));
  return std::fclose(f) == 0 && ok;
}

slim_csr_t* read_binrow(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return nullptr;
  slim_csr_t* m = csr_new();
  bool ok = std::fread(&m->nrows, sizeof(int32_t), 1, f) == 1 &&
            std::fread(&m->ncols, sizeof(int32_t), 1, f) == 1 && m->nrows >= 0;
  if (ok) {
    m->rowptr = xmalloc<ssize_t>(m->nrows + 1);
    ok = std::fread(m->rowptr, sizeof(ssize_t), m->nrows + 1, f) == size_t(m->nrows + 1);
  }
  if (ok) {
    const int64_t nnz = m->rowptr[m->nrows];
    m->rowind = xmalloc<int32_t>(nnz);
    m->rowval = xmalloc<float>(nnz);
    ok = std::fread(m->rowind, sizeof(int32_t), nnz, f) == size_t(nnz) &&
         std::fread(m->rowval, sizeof(float), nnz, f) == size_t(nnz);
  }
  std::fclose(f);
  if (!ok) {
    csr_free(m);
    return nullptr;
  }
  return m;
}

bool write_text_csr(const slim_csr_t* m, const char* path) {
  FILE* f = std::fopen(path, "w");
  if (!f) return false;
  for (int32_t r = 0; r < m->nrows; ++r) {
    for (ssize_t k = m->rowptr[r]; k < m->rowptr[r + 1]; ++k) {
      // %.9g round-trips a float exactly; GKlib's writer prints fewer digits,
      // any float syntax is accepted by both readers
      if (m->rowval) std::fprintf(f, " %d %.9g", m->rowind[k], double(m->rowval[k]));
      else std::fprintf(f, " %d", m->rowind[k]);
    }
    std::fputc('\n', f);
  }
  return std::fclose(f) == 0;
}

slim_csr_t* read_text_csr(const char* path) {
  FILE* f = std::fopen(path, "r");
  if (!f) return nullptr;
  std::vector<ssize_t> ptr{0};
  std::vector<int32_t> ind;
  std::vector<float> val;
  char* line = nullptr;
  size_t cap = 0;
  bool ok = true;
  while (getline(&line, &cap, f) >= 0) {
    char* p = line;
    for (;;) {
      char* e;
      const long id = std::strtol(p, &e, 10);
      if (e == p) break;
      p = e;
      const float v = std::strtof(p, &e);
      if (e == p) { ok = false; break; }  // id without a value
      p = e;
      ind.push_back(int32_t(id));
      val.push_back(v);
    }
    ptr.push_back(ssize_t(ind.size()));
  }
  std::free(line);
  std::fclose(f);
  if (!ok) return nullptr;
  slim_csr_t* m = csr_from_rows(int32_t(ptr.size() - 1), ptr.data(), ind.data(), val.data());
  return m;
}

}  // namespace slimamd
