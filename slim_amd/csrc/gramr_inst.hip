// cd_gramr_kernel<KR, KL> for up to 49 152 items, and the packing kernels; see gramr_inst.hpp
#define SLIM_GRAM_PACK_KERNELS
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
namespace slimamd {
GramrFn gramr_kernel(int nchunks, int* kr, int* kl) {
  const int k = (nchunks + kGramrNT - 1) / kGramrNT;
  *kl = 0;
  if (k <= 1) { *kr = 1; return cd_gramr_kernel<1, 0>; }
  if (k <= 3) { *kr = 3; return cd_gramr_kernel<3, 0>; }
  if (k <= 6) { *kr = 6; return cd_gramr_kernel<6, 0>; }
  if (k <= 13) { *kr = 10; *kl = 3; return gramr_kernel_k13(); }
  return nullptr;
}
PackScanFn gram_pack_scan_fn() { return gram_pack_scan; }
PackWriteFn gram_pack_write_fn() { return gram_pack_write; }
}  // namespace slimamd
