// cd_tile_kernel<16, *, *, 16>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE(tile_kernel_p16_nw16, 16, 16)
}
