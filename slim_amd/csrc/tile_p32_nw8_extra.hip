// cd_tile_kernel<32, *, false, 8, FSLIM | PARK>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE_EXTRA(tile_kernel_p32_nw8_extra, 8)
}
