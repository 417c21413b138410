// cd_gramr_kernel<KR, KL> for up to 49 152 items, and the packing kernels; see gramr_inst.hpp
#define SLIM_GRAM_PACK_KERNELS
#include <cstdlib>
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
namespace slimamd {
GramrFn gramr_kernel(int nchunks, bool dma, int* kr, int* kl, size_t* lds_bytes) {
  const int k = (nchunks + kGramrNT - 1) / kGramrNT;
  *kl = 0;
  int ring_ah = 2;
  GramrFn fn = nullptr;
  if (k <= 1) { *kr = 1; fn = dma ? cd_gramr_kernel<1, 0, true> : cd_gramr_kernel<1, 0, false>; }
  else if (k <= 3) {
    *kr = 3;
    // 128 VGPRs, two workgroups per CU (a few spills outside the row loop): a one-sweep C5 pair
    // 1.32 -> 1.19 s, the cold pair 3.30 -> 3.43 s -- 40 of the grid's 45 pairs are one-sweep
    // pairs.  SLIM_GPU_GRAMR_WPS=2: 256 VGPRs, one workgroup per CU.
    const char* e = std::getenv("SLIM_GPU_GRAMR_WPS");
    if (e && std::atoi(e) == 2) fn = dma ? cd_gramr_kernel<3, 0, true> : cd_gramr_kernel<3, 0, false>;
    else fn = dma ? cd_gramr_kernel<3, 0, true, 4> : cd_gramr_kernel<3, 0, false, 4>;
  }
  else if (k <= 6) { *kr = 6; fn = dma ? cd_gramr_kernel<6, 0, true> : cd_gramr_kernel<6, 0, false>; }
  else if (k <= kGramrMaxGroups) {
    static_assert(kGramrMaxGroups == 10 + 3, "the largest instantiation below");
    *kr = 10;
    *kl = 3;
    // <10, 3>: one form, chosen at build time (gramr_k13.hip: the LDS ring, 4 slots)
    fn = gramr_kernel_k13(&dma, &ring_ah);
  }
  *lds_bytes = sizeof(float) * (size_t)*kl * kPackGroup + (dma ? (size_t)gramr_ring_bytes(ring_ah) : 0) + (size_t)gramr_hdr_bytes();
  return fn;
}
GramrFn gramr_union_fn() { return gramr_union_kernel; }
int gramr_union_threads() { return kGramrUnionNT; }
PackScanFn gram_pack_scan_fn() { return gram_pack_scan; }
PackWriteFn gram_pack_write_fn() { return gram_pack_write; }
PackMetaFn gram_pack_meta_fn() { return gram_pack_meta; }
}  // namespace slimamd
