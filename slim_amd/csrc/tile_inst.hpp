// tile_inst.hpp -- one translation unit per (tile width, workgroup size) of cd_tile_kernel,
// so that the four geometries compile side by side (`make -j`); engine.hip picks the
// instantiation through tile_kernel().
#pragma once
#include "cd_tile.hpp"

namespace slimamd {

using KernelFn = void (*)(const DevMatrix, const SolveArgs);

KernelFn tile_kernel_p32_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p32_nw8(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw8(bool has_val, bool profile);

#define SLIM_TILE_INSTANTIATE(NAME, PP, NWW)                                        \
  KernelFn NAME(bool has_val, bool profile) {                                        \
    return has_val ? (profile ? cd_tile_kernel<PP, true, true, NWW>                  \
                              : cd_tile_kernel<PP, true, false, NWW>)                \
                   : (profile ? cd_tile_kernel<PP, false, true, NWW>                 \
                              : cd_tile_kernel<PP, false, false, NWW>);              \
  }

}  // namespace slimamd
