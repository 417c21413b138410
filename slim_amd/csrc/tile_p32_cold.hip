// cd_tile_kernel<32, *, false, *, false, 0>: cold starts only; see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
KernelFn tile_kernel_p32_cold(bool has_val, bool nw16) {
  return has_val ? (nw16 ? cd_tile_kernel<32, true, false, 16, false, 0> : cd_tile_kernel<32, true, false, 8, false, 0>)
                 : (nw16 ? cd_tile_kernel<32, false, false, 16, false, 0> : cd_tile_kernel<32, false, false, 8, false, 0>);
}
}  // namespace slimamd
