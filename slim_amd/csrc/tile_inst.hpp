// tile_inst.hpp -- the instantiations of cd_tile_kernel, a few per translation unit so that
// they compile side by side (`make -j`); engine.hip picks one through tile_kernel().
//   base : <P, HAS_VAL, PROFILE, NW>                       P in {32, 16}, NW in {16, 8}
//   extra: <32, HAS_VAL, false, NW, FSLIM> and <32, HAS_VAL, false, NW, false, PARK>
#pragma once
#include "cd_tile.hpp"

namespace slimamd {

using KernelFn = void (*)(const DevMatrix, const SolveArgs);

KernelFn tile_kernel_p32_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p32_nw8(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw8(bool has_val, bool profile);
// P = 32 only: neighbour selection (FSLIM) / LDS parking of the second-to-last chunk
KernelFn tile_kernel_p32_nw16_extra(bool has_val, bool fslim);
KernelFn tile_kernel_p32_nw8_extra(bool has_val, bool fslim);
// P = 32, 8 wavefronts, one workgroup per CU, two blocks per wavefront (WIDE)
KernelFn tile_kernel_p32_wide(bool has_val, bool profile);

#define SLIM_TILE_INSTANTIATE(NAME, PP, NWW)                                        \
  KernelFn NAME(bool has_val, bool profile) {                                        \
    return has_val ? (profile ? cd_tile_kernel<PP, true, true, NWW>                  \
                              : cd_tile_kernel<PP, true, false, NWW>)                \
                   : (profile ? cd_tile_kernel<PP, false, true, NWW>                 \
                              : cd_tile_kernel<PP, false, false, NWW>);              \
  }

#define SLIM_TILE_INSTANTIATE_EXTRA(NAME, NWW)                                             \
  KernelFn NAME(bool has_val, bool fslim) {                                                \
    return has_val ? (fslim ? cd_tile_kernel<32, true, false, NWW, true, false>            \
                            : cd_tile_kernel<32, true, false, NWW, false, true>)           \
                   : (fslim ? cd_tile_kernel<32, false, false, NWW, true, false>           \
                            : cd_tile_kernel<32, false, false, NWW, false, true>);         \
  }

#define SLIM_TILE_INSTANTIATE_WIDE(NAME)                                                      \
  KernelFn NAME(bool has_val, bool profile) {                                                 \
    return has_val ? (profile ? cd_tile_kernel<32, true, true, 8, false, false, true>         \
                              : cd_tile_kernel<32, true, false, 8, false, false, true>)       \
                   : (profile ? cd_tile_kernel<32, false, true, 8, false, false, true>        \
                              : cd_tile_kernel<32, false, false, 8, false, false, true>);     \
  }

}  // namespace slimamd
