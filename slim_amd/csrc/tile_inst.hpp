// tile_inst.hpp -- the instantiations of cd_tile_kernel, a few per translation unit so that
// they compile side by side (`make -j`); engine.hip picks one through tile_kernel().
//   base : <P, HAS_VAL, PROFILE, NW>                       P in {32, 16}, NW in {16, 8}
//   FSLIM: <32, HAS_VAL, false, NW, true>
//   cold / row fold: <32, HAS_VAL, false, NW, false, 0 / 2>
#pragma once
#include "cd_tile.hpp"

namespace slimamd {

using KernelFn = void (*)(const DevMatrix, const SolveArgs);

KernelFn tile_kernel_p32_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p32_nw8(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw16(bool has_val, bool profile);
KernelFn tile_kernel_p16_nw8(bool has_val, bool profile);
// P = 32 only: neighbour selection instead of the l1 screen (FSLIM)
KernelFn tile_kernel_p32_fslim(bool has_val, bool nw16);
// P = 32 only: a kernel without any warm-start code (cold starts), and one that folds the
// warm-start coefficients row by row (FOLD = 2); the base instantiations fold column by column
KernelFn tile_kernel_p32_cold(bool has_val, bool nw16);
KernelFn tile_kernel_p32_rowfold(bool has_val, bool nw16);

#define SLIM_TILE_INSTANTIATE(NAME, PP, NWW)                                        \
  KernelFn NAME(bool has_val, bool profile) {                                        \
    return has_val ? (profile ? cd_tile_kernel<PP, true, true, NWW>                  \
                              : cd_tile_kernel<PP, true, false, NWW>)                \
                   : (profile ? cd_tile_kernel<PP, false, true, NWW>                 \
                              : cd_tile_kernel<PP, false, false, NWW>);              \
  }

}  // namespace slimamd
