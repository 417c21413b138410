// multi_gpu.cpp -- SLIM_Learn sharded over the GPUs of one node, inside one process.
//
// The reference parallelises inside SLIM_Learn: an OpenMP team over the item columns with
// the read-only training matrix shared (src/libslim/api.c:69-85 -> estimate.c:371-373,402).
// The device form of that team: one host thread + one HIP stream per GPU, R replicated in
// every GPU's HBM, the cost-ordered work list dealt to the devices in granules of 32 columns
// (engine.hip: shard_count / shard_index), no exchange during the solve, and the learned
// columns concatenated on the host into one model (SaveModel, estimate.c:570-593).
//
// Replication of R: every device thread copies the caller's CSR over its own PCIe link
// (default, "h2d"), or device 0 receives it once and RCCL broadcasts it over xGMI
// (SLIM_GPU_STAGE=rccl; librccl is loaded on demand, the library has no link-time
// dependency on it); both leave every device to build its own column view.  SLIM_GPU_STAGE=view
// stages on the first device only and copies the finished views device to device.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "engine.hpp"
#include "host_csr.hpp"

namespace slimamd {

namespace {

// SLIM_GPU_DEVICES="0,1,2" names the devices explicitly (ordinals may repeat: tests run the
// two-shard path on a one-GPU box with "0,0"); otherwise devices 0 .. ngpus-1.
bool resolve_devices(const LearnOptions& opt, std::vector<int>* out, std::string* err) {
  out->clear();
  const int count = device_count();
  if (count <= 0) {
    *err = "no usable gfx950 device -- the SLIM CD path has no CPU fallback";
    return false;
  }
  // An explicit device option (SLIM_OPTION_GPU_DEVICE with ngpus <= 1: what a
  // one-process-per-GPU launcher passes) wins over the environment: a rank that inherits
  // SLIM_GPU_DEVICES must not replicate R on every listed GPU.
  if (opt.ngpus <= 1 && opt.device >= 0) {
    if (opt.device >= count) {
      *err = "device " + std::to_string(opt.device) + " requested but the node has " +
             std::to_string(count);
      return false;
    }
    out->push_back(opt.device);
    return true;
  }
  if (const char* e = std::getenv("SLIM_GPU_DEVICES")) {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p) break;
      if (d < 0 || d >= count) {
        *err = "SLIM_GPU_DEVICES names device " + std::to_string(d) + " but the node has " +
               std::to_string(count);
        return false;
      }
      out->push_back((int)d);
      p = *end == ',' ? end + 1 : end;
    }
    // the list names the devices to use; ngpus (unset = 1) says how many of them
    if ((int)out->size() > std::max(1, opt.ngpus)) out->resize((size_t)std::max(1, opt.ngpus));
    if (!out->empty()) return true;
  }
  if (opt.ngpus <= 1) {
    int dev = opt.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;  // -1: the current device
    out->push_back(dev);
    return true;
  }
  if (opt.ngpus > count) {
    *err = "ngpus=" + std::to_string(opt.ngpus) + " requested but the node has " +
           std::to_string(count) + " GPU(s)";
    return false;
  }
  for (int d = 0; d < opt.ngpus; ++d) out->push_back(d);
  return true;
}

// ---- RCCL broadcast of the CSR (opt-in) ------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  bool load() {
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
  }
};

struct DevCsr {
  int64_t* rowptr = nullptr;
  int32_t* rowind = nullptr;
  float* rowval = nullptr;
};

// Device 0 of `devs` gets the CSR over PCIe, the others over xGMI (ncclBroadcast, root 0,
// one communicator per device, all in this process).  Returns false (with *err) on failure;
// buffers already allocated are released by the caller through free_dev().
bool broadcast_csr_rccl(const std::vector<int>& devs, int32_t nrows, const ssize_t* rowptr,
                        const int32_t* rowind, const float* rowval, std::vector<DevCsr>* out,
                        std::string* err) {
  Rccl nc;
  if (!nc.load()) {
    *err = "SLIM_GPU_STAGE=rccl: librccl.so could not be loaded";
    return false;
  }
  const size_t n = devs.size();
  const int64_t nnz = rowptr[nrows];
  out->assign(n, DevCsr());
  std::vector<hipStream_t> streams(n, nullptr);
  std::vector<void*> comms(n, nullptr);
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    *err = std::string("SLIM_GPU_STAGE=rccl: ") + what + ": " + hipGetErrorString(e);
    return false;
  };
  bool ok = true;
  for (size_t d = 0; d < n && ok; ++d) {
    ok = hip_ok(hipSetDevice(devs[d]), "hipSetDevice") &&
         hip_ok(hipStreamCreateWithFlags(&streams[d], hipStreamNonBlocking), "stream") &&
         hip_ok(hipMalloc((void**)&(*out)[d].rowptr, sizeof(int64_t) * ((size_t)nrows + 1)), "hipMalloc") &&
         hip_ok(hipMalloc((void**)&(*out)[d].rowind, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)), "hipMalloc") &&
         (!rowval || hip_ok(hipMalloc((void**)&(*out)[d].rowval, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1)), "hipMalloc"));
  }
  if (ok) {
    ok = hip_ok(hipSetDevice(devs[0]), "hipSetDevice") &&
         hip_ok(hipMemcpyAsync((*out)[0].rowptr, rowptr, sizeof(int64_t) * ((size_t)nrows + 1), hipMemcpyHostToDevice, streams[0]), "H2D") &&
         hip_ok(hipMemcpyAsync((*out)[0].rowind, rowind, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, streams[0]), "H2D") &&
         (!rowval || hip_ok(hipMemcpyAsync((*out)[0].rowval, rowval, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, streams[0]), "H2D"));
  }
  if (ok && nc.CommInitAll(comms.data(), (int)n, devs.data()) != 0) {
    *err = "SLIM_GPU_STAGE=rccl: ncclCommInitAll failed";
    ok = false;
  }
  if (ok) {
    // three grouped broadcasts (one per array); ncclInt8 = 0: plain bytes
    struct Part { size_t bytes; int which; };
    const Part parts[3] = {{sizeof(int64_t) * ((size_t)nrows + 1), 0},
                           {sizeof(int32_t) * (size_t)nnz, 1},
                           {rowval ? sizeof(float) * (size_t)nnz : 0, 2}};
    for (const Part& pt : parts) {
      if (pt.bytes == 0) continue;
      nc.GroupStart();
      for (size_t d = 0; d < n; ++d) {
        (void)hipSetDevice(devs[d]);  // (older RCCL wants the current device to match the comm)
        void* buf = pt.which == 0 ? (void*)(*out)[d].rowptr
                                  : pt.which == 1 ? (void*)(*out)[d].rowind : (void*)(*out)[d].rowval;
        if (nc.Broadcast(buf, buf, pt.bytes, /*ncclInt8*/ 0, /*root*/ 0, comms[d], streams[d]) != 0) ok = false;
      }
      if (nc.GroupEnd() != 0) ok = false;
    }
    if (!ok) *err = "SLIM_GPU_STAGE=rccl: ncclBroadcast failed";
  }
  for (size_t d = 0; d < n; ++d) {
    if (streams[d]) {
      (void)hipSetDevice(devs[d]);
      if (hipStreamSynchronize(streams[d]) != hipSuccess && ok) {
        ok = false;
        *err = "SLIM_GPU_STAGE=rccl: stream synchronisation failed";
      }
    }
  }
  for (size_t d = 0; d < n; ++d) {
    if (comms[d]) nc.CommDestroy(comms[d]);
    if (streams[d]) {
      (void)hipSetDevice(devs[d]);
      (void)hipStreamDestroy(streams[d]);
    }
  }
  return ok;
}

void free_dev(const std::vector<int>& devs, std::vector<DevCsr>& bufs) {
  for (size_t d = 0; d < bufs.size(); ++d) {
    (void)hipSetDevice(devs[d]);
    (void)hipFree(bufs[d].rowptr);
    (void)hipFree(bufs[d].rowind);
    (void)hipFree(bufs[d].rowval);
    bufs[d] = DevCsr();
  }
}

}  // namespace

slimgpu_matrix_t* multi_from_host(int32_t nrows, const ssize_t* rowptr, const int32_t* rowind,
                                  const float* rowval, const LearnOptions& opt, int32_t* status) {
  std::string err;
  std::vector<int> devs;
  if (!resolve_devices(opt, &devs, &err)) {
    set_error("SLIM_Learn: " + err);
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  const size_t n = devs.size();
  const char* stage = std::getenv("SLIM_GPU_STAGE");
  const bool use_rccl = stage && std::strcmp(stage, "rccl") == 0 && nrows >= 0 && rowptr &&
                        (rowptr[nrows] == 0 || rowind);
  if (n == 1 && !use_rccl) {
    LearnOptions o = opt;
    o.device = devs[0];
    return matrix_from_host(nrows, rowptr, rowind, rowval, o, status);
  }
  std::vector<slimgpu_matrix_t*> mats(n, nullptr);
  std::vector<int32_t> st(n, SLIM_ERROR);
  std::vector<std::string> msg(n);
  // SLIM_GPU_STAGE=view: the first device stages R (one H2D copy, one sort), every other
  // device receives the finished CSR + CSC + column scalars device to device -- N - 1 peer
  // copies on N - 1 xGMI links at once, no sort and no sort temporaries on the targets.
  // Which form wins on a real node (this, N parallel H2D copies + N sorts, or the RCCL
  // broadcast of the CSR) is unmeasured: no multi-GPU box is available to the builder; all three
  // are ~1 s against a solve of a minute.
  if (stage && std::strcmp(stage, "view") == 0 && n > 1) {
    LearnOptions o0 = opt;
    o0.device = devs[0];
    o0.ngpus = 1;
    mats[0] = matrix_from_host(nrows, rowptr, rowind, rowval, o0, status);
    if (!mats[0]) return nullptr;
    auto clone = [&](size_t d) {
      set_error("");
      mats[d] = matrix_clone_to_device(mats[0], devs[d], &st[d]);
      if (!mats[d]) msg[d] = last_error();
    };
    {
      std::vector<std::thread> team;
      for (size_t d = 1; d < n; ++d) team.emplace_back(clone, d);
      for (auto& t : team) t.join();
    }
    double setup_ms = matrix_setup_ms(mats[0]), copy_ms = 0;
    for (size_t d = 1; d < n; ++d) {
      if (!mats[d]) {
        set_error(msg[d]);
        if (status) *status = st[d];
        for (slimgpu_matrix_t* m : mats) matrix_free(m);
        return nullptr;
      }
      copy_ms = std::max(copy_ms, matrix_setup_ms(mats[d]));
    }
    if (std::getenv("SLIM_GPU_TRACE"))
      std::fprintf(stderr, "[trace] staging 'view': device %d staged in %.1f ms, %zu device-to-device "
                           "copies in %.1f ms\n", devs[0], setup_ms, n - 1, copy_ms);
    for (size_t d = 1; d < n; ++d) matrix_add_replica(mats[0], mats[d]);
    matrix_set_setup_ms(mats[0], setup_ms + copy_ms);
    if (status) *status = SLIM_OK;
    return mats[0];
  }
  std::vector<DevCsr> bufs;
  if (use_rccl && !broadcast_csr_rccl(devs, nrows, rowptr, rowind, rowval, &bufs, &err)) {
    set_error("SLIM_Learn: " + err);
    if (status) *status = SLIM_ERROR;
    free_dev(devs, bufs);
    return nullptr;
  }
  auto work = [&](size_t d) {
    set_error("");
    LearnOptions o = opt;
    o.device = devs[d];
    if (use_rccl) {
      mats[d] = matrix_from_device(nrows, 0, bufs[d].rowptr, bufs[d].rowind, bufs[d].rowval, o, &st[d]);
      if (mats[d]) {
        matrix_adopt_csr(mats[d]);  // the handle frees the broadcast buffers
        bufs[d] = DevCsr();
      }
    } else {
      mats[d] = matrix_from_host(nrows, rowptr, rowind, rowval, o, &st[d]);
    }
    if (!mats[d]) msg[d] = last_error();
  };
  {
    std::vector<std::thread> team;
    for (size_t d = 0; d < n; ++d) team.emplace_back(work, d);
    for (auto& t : team) t.join();
  }
  double setup_ms = 0;
  for (size_t d = 0; d < n; ++d) {
    if (!mats[d]) {
      set_error(msg[d]);
      if (status) *status = st[d];
      for (slimgpu_matrix_t* m : mats) matrix_free(m);
      if (use_rccl) free_dev(devs, bufs);
      return nullptr;
    }
    setup_ms = std::max(setup_ms, matrix_setup_ms(mats[d]));
  }
  for (size_t d = 1; d < n; ++d) matrix_add_replica(mats[0], mats[d]);
  matrix_set_setup_ms(mats[0], setup_ms);
  if (status) *status = SLIM_OK;
  return mats[0];
}

slim_csr_t* multi_learn(slimgpu_matrix_t* m0, const LearnOptions& opt, const slim_csr_t* imodel,
                        int32_t* status, const int32_t* columns, int32_t ncolumns) {
  if (!m0) {
    set_error("SLIMGPU_Learn: null matrix");
    if (status) *status = SLIM_ERROR_INPUT;
    return nullptr;
  }
  std::vector<slimgpu_matrix_t*> mats(1, m0);
  for (slimgpu_matrix_t* r : matrix_replicas(m0)) mats.push_back(r);
  const size_t n = mats.size();
  if (n == 1) return learn_cd(m0, opt, imodel, status, columns, ncolumns);
  // one host thread per device; the caller's own shard (index i of c) splits into shards
  // i*n + d of c*n: granules d, d + n, ... of the cost-ordered list for a whole-matrix call
  std::vector<slim_csr_t*> part(n, nullptr);
  std::vector<int32_t> st(n, SLIM_ERROR);
  std::vector<std::string> msg(n);
  std::vector<slimgpu_stats_t> stats(n);
  std::vector<ColumnStats> cstats(n);
  auto work = [&](size_t d) {
    set_error("");
    LearnOptions o = opt;
    o.dbglvl = 0;  // the summary line is printed once, below
    o.shard_count = opt.shard_count * (int32_t)n;
    o.shard_index = opt.shard_index * (int32_t)n + (int32_t)d;
    part[d] = learn_cd(mats[d], o, imodel, &st[d], columns, ncolumns, /*row_view=*/false);
    if (!part[d]) {
      msg[d] = last_error();
    } else {
      stats[d] = last_stats();
      cstats[d] = last_column_stats();
    }
  };
  {
    std::vector<std::thread> team;
    for (size_t d = 0; d < n; ++d) team.emplace_back(work, d);
    for (auto& t : team) t.join();
  }
  bool ok = true;
  for (size_t d = 0; d < n && ok; ++d) {
    if (!part[d]) {
      set_error(msg[d]);
      if (status) *status = st[d];
      ok = false;
    }
  }
  if (!ok) {
    for (slim_csr_t* p : part) csr_free(p);
    return nullptr;
  }
  // SaveModel: column c comes from the one shard that solved it (the others left it empty)
  const int32_t ncols = part[0]->ncols;
  int64_t tnnz = 0;
  for (size_t d = 0; d < n; ++d) tnnz += part[d]->colptr[ncols];
  auto* colptr = static_cast<ssize_t*>(std::malloc(sizeof(ssize_t) * ((size_t)ncols + 1)));
  auto* colind = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(tnnz, 1)));
  auto* colval = static_cast<float*>(std::malloc(sizeof(float) * (size_t)std::max<int64_t>(tnnz, 1)));
  if (!colptr || !colind || !colval) {
    std::free(colptr); std::free(colind); std::free(colval);
    for (slim_csr_t* p : part) csr_free(p);
    set_error("SLIM_Learn: out of host memory for the model");
    if (status) *status = SLIM_ERROR_MEMORY;
    return nullptr;
  }
  colptr[0] = 0;
  for (int32_t c = 0; c < ncols; ++c) {
    ssize_t at = colptr[c];
    for (size_t d = 0; d < n; ++d) {
      const ssize_t lo = part[d]->colptr[c], cnt = part[d]->colptr[c + 1] - lo;
      if (cnt > 0) {
        std::memcpy(colind + at, part[d]->colind + lo, sizeof(int32_t) * (size_t)cnt);
        std::memcpy(colval + at, part[d]->colval + lo, sizeof(float) * (size_t)cnt);
        at += cnt;
      }
    }
    colptr[c + 1] = at;
  }
  for (slim_csr_t* p : part) csr_free(p);
  slim_csr_t* model = model_from_columns(ncols, colptr, colind, colval);

  // the reductions of EstimateModelCD (estimate.c:371-373) and the counters, over the team
  slimgpu_stats_t tot = stats[0];
  ColumnStats& cs = last_column_stats();
  cs = cstats[0];
  for (size_t d = 1; d < n; ++d) {
    const slimgpu_stats_t& s = stats[d];
    tot.ncols_solved += s.ncols_solved;
    tot.nwaves += s.nwaves;
    tot.kernel_ms = std::max(tot.kernel_ms, s.kernel_ms);
    tot.gather_ms = std::max(tot.gather_ms, s.gather_ms);
    tot.total_ms = std::max(tot.total_ms, s.total_ms);
    tot.G += s.G; tot.D += s.D; tot.U += s.U; tot.nnzW += s.nnzW;
    tot.sweeps += s.sweeps; tot.visits += s.visits;
    tot.alg_bytes += s.alg_bytes;
    tot.error += s.error; tot.objval += s.objval;
    for (size_t c = 0; c < cs.nacols.size() && c < cstats[d].nacols.size(); ++c) {
      if (cstats[d].sweeps[c] == 0 && cstats[d].G[c] == 0) continue;  // not this shard's column
      cs.nacols[c] = cstats[d].nacols[c];
      cs.sweeps[c] = cstats[d].sweeps[c];
      cs.conv[c] = cstats[d].conv[c];
      cs.G[c] = cstats[d].G[c];
      cs.D[c] = cstats[d].D[c];
      cs.U[c] = cstats[d].U[c];
    }
  }
  tot.setup_ms = matrix_setup_ms(m0);
  last_stats() = tot;
  if (opt.dbglvl & SLIM_DBG_INFO)  // estimate.c:552-555
    std::printf("Done estimation: loss: %.5le, fit: %.5le, ffrac: %.3lf,  #nzs: %zd\n", tot.objval,
                tot.error, tot.objval != 0 ? tot.error / tot.objval : 0.0, (ssize_t)tot.nnzW);
  if (status) *status = SLIM_OK;
  return model;
}

}  // namespace slimamd
