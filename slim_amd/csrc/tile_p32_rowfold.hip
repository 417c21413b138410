// cd_tile_kernel<32, *, false, *, false, 2>: warm starts folded row by row; see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
KernelFn tile_kernel_p32_rowfold(bool has_val, bool nw16) {
  return has_val ? (nw16 ? cd_tile_kernel<32, true, false, 16, false, 2> : cd_tile_kernel<32, true, false, 8, false, 2>)
                 : (nw16 ? cd_tile_kernel<32, false, false, 16, false, 2> : cd_tile_kernel<32, false, false, 8, false, 2>);
}
}  // namespace slimamd
