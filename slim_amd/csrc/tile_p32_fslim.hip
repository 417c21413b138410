// cd_tile_kernel<32, *, false, *, FSLIM>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
KernelFn tile_kernel_p32_fslim(bool has_val, bool nw16) {
  return has_val ? (nw16 ? cd_tile_kernel<32, true, false, 16, true> : cd_tile_kernel<32, true, false, 8, true>)
                 : (nw16 ? cd_tile_kernel<32, false, false, 16, true> : cd_tile_kernel<32, false, false, 8, true>);
}
}  // namespace slimamd
