// cli_common.hpp -- shared pieces of the command-line programs: option parsing in the
// reference's long-only style (-name=value or -name value; /root/reference/src/programs/
// cmdline_*.c use gk_getopt_long_only), the matrix file formats GKlib's gk_csr_Read/Write
// handle for them (csr, csrnv, cluto, ijv; layouts in SURVEY.md Appendix C) and the HR/ARHR
// evaluation of src/programs/slim_predict.c:181-236.  The programs use only the public C
// ABI of libslim.so (include/slim.h, include/slim_gpu.h).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../../include/slim_gpu.h"

namespace slimcli {

[[noreturn]] inline void die(const std::string& msg) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(1);
}

// ---- options -------------------------------------------------------------------
struct OptSpec {
  const char* name;
  bool takes_value;
};

struct Args {
  std::map<std::string, std::string> opt;  // flags map to "1"
  std::vector<std::string> pos;
  bool has(const char* k) const { return opt.count(k) != 0; }
  std::string str(const char* k, const std::string& d) const {
    auto it = opt.find(k);
    return it == opt.end() ? d : it->second;
  }
  double num(const char* k, double d) const { return has(k) ? std::atof(opt.at(k).c_str()) : d; }
  int integer(const char* k, int d) const { return has(k) ? std::atoi(opt.at(k).c_str()) : d; }
};

inline Args parse_args(int argc, char** argv, const std::vector<OptSpec>& specs) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string t = argv[i];
    if (t.size() > 1 && t[0] == '-' && !(t[1] >= '0' && t[1] <= '9') && t[1] != '.') {
      std::string name = t.substr(t[1] == '-' ? 2 : 1), value;
      bool inline_value = false;
      const size_t eq = name.find('=');
      if (eq != std::string::npos) {
        value = name.substr(eq + 1);
        name = name.substr(0, eq);
        inline_value = true;
      }
      const OptSpec* spec = nullptr;
      for (const auto& s : specs)
        if (name == s.name) spec = &s;
      if (!spec) die("Illegal command-line option(s) -" + name);
      if (spec->takes_value) {
        if (!inline_value) {
          if (i + 1 >= argc) die("Option -" + name + " needs a value.");
          value = argv[++i];
        }
        a.opt[name] = value;
      } else {
        a.opt[name] = "1";
      }
    } else {
      a.pos.push_back(t);
    }
  }
  return a;
}

// ---- matrices ------------------------------------------------------------------
struct Csr {
  int32_t nrows = 0, ncols = 0;
  std::vector<ssize_t> ptr{0};
  std::vector<int32_t> ind;
  std::vector<float> val;
  bool has_val = true;
  float* valptr() { return has_val ? val.data() : nullptr; }
  ssize_t nnz() const { return ptr.back(); }
};

enum class Fmt { csr, csrnv, cluto, ijv };

inline Fmt parse_fmt(const std::string& s) {
  if (s == "csr") return Fmt::csr;
  if (s == "csrnv") return Fmt::csrnv;
  if (s == "cluto") return Fmt::cluto;
  if (s == "ijv") return Fmt::ijv;
  die("Invalid -ifmt of " + s + ".");
}

inline bool file_exists(const std::string& p) {
  FILE* f = std::fopen(p.c_str(), "r");
  if (f) std::fclose(f);
  return f != nullptr;
}

// csr / csrnv: one line per row, "id value" pairs (or bare ids), ids as written.
// cluto: header "nrows ncols nnz", then rows of 1-based "col value" pairs.
// ijv: "row col value" triplets, 0-based.
inline Csr read_matrix(const std::string& path, Fmt fmt) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) die("Failed to open " + path);
  Csr m;
  char* line = nullptr;
  size_t cap = 0;
  int32_t maxcol = -1;
  if (fmt == Fmt::ijv) {
    std::vector<std::vector<std::pair<int32_t, float>>> rows;
    long r, c;
    double v;
    while (std::fscanf(f, "%ld %ld %lf", &r, &c, &v) == 3) {
      if (r < 0 || c < 0) die("negative id in " + path);
      if ((size_t)r >= rows.size()) rows.resize(r + 1);
      rows[r].emplace_back((int32_t)c, (float)v);
    }
    for (auto& row : rows) {
      std::stable_sort(row.begin(), row.end(),
                       [](const auto& a, const auto& b) { return a.first < b.first; });
      for (auto& e : row) {
        m.ind.push_back(e.first);
        m.val.push_back(e.second);
        maxcol = std::max(maxcol, e.first);
      }
      m.ptr.push_back((ssize_t)m.ind.size());
    }
  } else {
    bool header = fmt == Fmt::cluto;
    long hdr_cols = 0;
    while (getline(&line, &cap, f) >= 0) {
      char* p = line;
      if (header) {
        long hr = 0, hn = 0;
        if (std::sscanf(p, "%ld %ld %ld", &hr, &hdr_cols, &hn) != 3) die("bad cluto header in " + path);
        header = false;
        continue;
      }
      for (;;) {
        char* e;
        const long id = std::strtol(p, &e, 10);
        if (e == p) break;
        p = e;
        float v = 1.0f;
        if (fmt != Fmt::csrnv) {
          v = std::strtof(p, &e);
          if (e == p) die("id without a value in " + path);
          p = e;
        }
        const int32_t col = (int32_t)(fmt == Fmt::cluto ? id - 1 : id);
        m.ind.push_back(col);
        m.val.push_back(v);
        maxcol = std::max(maxcol, col);
      }
      m.ptr.push_back((ssize_t)m.ind.size());
    }
    if (fmt == Fmt::cluto) maxcol = std::max<int32_t>(maxcol, (int32_t)hdr_cols - 1);
  }
  std::free(line);
  std::fclose(f);
  m.nrows = (int32_t)m.ptr.size() - 1;
  m.ncols = maxcol + 1;
  return m;
}

// row view of a model handle, in the same format as the input (slim_learn.c:83)
inline void write_matrix(const slim_csr_t* m, const std::string& path, Fmt fmt) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) die("Failed to open " + path + " for writing");
  if (fmt == Fmt::cluto) std::fprintf(f, "%d %d %zd\n", m->nrows, m->ncols, m->rowptr[m->nrows]);
  for (int32_t r = 0; r < m->nrows; ++r) {
    for (ssize_t k = m->rowptr[r]; k < m->rowptr[r + 1]; ++k) {
      if (fmt == Fmt::ijv)
        std::fprintf(f, "%d %d %.9g\n", r, m->rowind[k], (double)m->rowval[k]);
      else if (fmt == Fmt::csrnv)
        std::fprintf(f, " %d", m->rowind[k]);
      else
        std::fprintf(f, " %d %.9g", m->rowind[k] + (fmt == Fmt::cluto ? 1 : 0), (double)m->rowval[k]);
    }
    if (fmt != Fmt::ijv) std::fputc('\n', f);
  }
  std::fclose(f);
}

// library-owned handle of a matrix read from a file
inline slim_t* to_handle(Csr& m) {
  slim_t* h = nullptr;
  if (Py_csr_wrapper(m.nrows, m.ptr.data(), m.ind.data(), m.valptr(), &h) != SLIM_OK)
    die("out of memory");
  return h;
}

// model handle (row + column views) from a matrix file: through the binary row format
inline slim_t* read_model(const std::string& path, Fmt fmt) {
  Csr m = read_matrix(path, fmt);
  slim_t* h = to_handle(m);
  slim_csr_t* c = static_cast<slim_csr_t*>(h);
  c->ncols = std::max(c->ncols, c->nrows);
  std::string tmp = path + ".tmp.slimbin";
  if (SLIM_WriteModel(h, const_cast<char*>(tmp.c_str())) != SLIM_OK) die("cannot write " + tmp);
  slim_t* model = SLIM_ReadModel(const_cast<char*>(tmp.c_str()));
  std::remove(tmp.c_str());
  Py_csr_free(h);
  if (!model) die("cannot read model " + path);
  return model;
}

// ---- evaluation (slim_predict.c:181-236; empty test rows are skipped like
// slim_mselect.c:129 does -- slim_predict.c divides by zero on them) ----------------
struct Eval {
  double hr = 0, hr_head = 0, hr_tail = 0, arhr = 0;
  int nvalid = 0, nvalid_head = 0, nvalid_tail = 0;
};

inline Eval evaluate_lists_host(const Csr& tst, const std::vector<int32_t>& lists,
                           const std::vector<int32_t>& lens, int nrcmds, const int32_t* fmarker,
                           int32_t ncols) {
  Eval e;
  std::vector<int32_t> wanted(ncols, -1);
  float hr[3] = {0, 0, 0}, arhr = 0;
  for (int32_t u = 0; u < tst.nrows && u < (int32_t)lens.size(); ++u) {
    const ssize_t t0 = tst.ptr[u], t1 = tst.ptr[u + 1];
    if (t1 - t0 < 1) continue;
    ++e.nvalid;
    int ntrue[2] = {0, 0}, nhits[3] = {0, 0, 0};
    bool head = false, tail = false;
    float gain = 0, ideal = 0;
    for (ssize_t z = t0; z < t1; ++z) {
      const int32_t it = tst.ind[z];
      if (it < 0 || it >= ncols) continue;
      wanted[it] = u;
      ++ntrue[fmarker[it]];
      (fmarker[it] ? tail : head) = true;
      ideal += 1.0 / (1.0 + double(z - t0));
    }
    e.nvalid_head += head;
    e.nvalid_tail += tail;
    for (int r = 0; r < lens[u]; ++r) {
      const int32_t id = lists[(size_t)u * nrcmds + r];
      if (id >= 0 && id < ncols && wanted[id] == u) {
        ++nhits[fmarker[id]];
        ++nhits[2];
        gain += 1.0 / (1.0 + r);
      }
    }
    hr[0] += nhits[0] > 0 ? 1.0 * nhits[0] / ntrue[0] : 0.0;
    hr[1] += nhits[1] > 0 ? 1.0 * nhits[1] / ntrue[1] : 0.0;
    hr[2] += 1.0 * nhits[2] / double(t1 - t0);
    arhr += ideal > 0 ? gain / ideal : 0.0f;
  }
  e.hr = e.nvalid ? hr[2] / e.nvalid : 0;
  e.hr_head = e.nvalid_head ? hr[0] / e.nvalid_head : 0;
  e.hr_tail = e.nvalid_tail ? hr[1] / e.nvalid_tail : 0;
  e.arhr = e.nvalid ? arhr / e.nvalid : 0;
  return e;
}

// HR / ARHR on the GPU (SLIMGPU_Evaluate: the same figures as the host loop above); the host
// loop serves when no device is usable -- evaluation, unlike training, may run anywhere.
inline Eval evaluate_lists(Csr& tst, const std::vector<int32_t>& lists,
                           const std::vector<int32_t>& lens, int nrcmds, const int32_t* fmarker,
                           int32_t ncols) {
  const int32_t nusers = std::min<int32_t>(tst.nrows, (int32_t)lens.size());
  if (SLIMGPU_DeviceCount() > 0 && nusers > 0) {
    slim_t* th = to_handle(tst);
    double m[4];
    int32_t nv[3];
    const int32_t rc =
        SLIMGPU_Evaluate(nusers, nrcmds, lists.data(), lens.data(), th, fmarker, ncols, m, nv);
    Py_csr_free(th);
    if (rc == SLIM_OK) {
      Eval e;
      e.hr = m[0]; e.hr_head = m[1]; e.hr_tail = m[2]; e.arhr = m[3];
      e.nvalid = nv[0]; e.nvalid_head = nv[1]; e.nvalid_tail = nv[2];
      return e;
    }
  }
  return evaluate_lists_host(tst, lists, lens, nrcmds, fmarker, ncols);
}

inline void banner() {
  std::printf("------------------------------------------------------------------\n");
  std::printf("SLIM, version %s (MI355X engine)\n", SLIM_VERSION);
  std::printf("------------------------------------------------------------------\n");
}

}  // namespace slimcli
