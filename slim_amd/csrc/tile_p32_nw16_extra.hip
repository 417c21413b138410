// cd_tile_kernel<32, *, false, 16, FSLIM | PARK>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE_EXTRA(tile_kernel_p32_nw16_extra, 16)
}
