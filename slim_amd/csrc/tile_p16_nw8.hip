// cd_tile_kernel<16, *, *, 8>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE(tile_kernel_p16_nw8, 16, 8)
}
