// cd_gramr_kernel<KR, KL> for up to 49 152 items, and the packing kernels; see gramr_inst.hpp
#define SLIM_GRAM_PACK_KERNELS
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
namespace slimamd {
GramrFn gramr_kernel(int nchunks, bool dma, int* kr, int* kl, size_t* lds_bytes) {
  const int k = (nchunks + kGramrNT - 1) / kGramrNT;
  *kl = 0;
  GramrFn fn = nullptr;
  if (k <= 1) { *kr = 1; fn = dma ? cd_gramr_kernel<1, 0, true> : cd_gramr_kernel<1, 0, false>; }
  else if (k <= 3) { *kr = 3; fn = dma ? cd_gramr_kernel<3, 0, true> : cd_gramr_kernel<3, 0, false>; }
  else if (k <= 6) { *kr = 6; fn = dma ? cd_gramr_kernel<6, 0, true> : cd_gramr_kernel<6, 0, false>; }
  else if (k <= 13) { *kr = 10; *kl = 3; fn = gramr_kernel_k13(dma); }
  *lds_bytes = sizeof(float) * (size_t)*kl * kPackGroup + (dma ? (size_t)kGramrRingBytes : 0);
  return fn;
}
PackScanFn gram_pack_scan_fn() { return gram_pack_scan; }
PackWriteFn gram_pack_write_fn() { return gram_pack_write; }
}  // namespace slimamd
