// cd_gramr_kernel<10, 3>: up to 106 496 items (the 1M x 100K configuration); see gramr_inst.hpp
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
namespace slimamd {
GramrFn gramr_kernel_k13(bool dma, bool alt) {
  if (alt) return cd_gramr_kernel<10, 3, true, 2, 3>;
  return dma ? cd_gramr_kernel<10, 3, true> : cd_gramr_kernel<10, 3, false>;
}
}  // namespace slimamd
