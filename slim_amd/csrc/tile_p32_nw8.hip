// cd_tile_kernel<32, *, *, 8>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE(tile_kernel_p32_nw8, 32, 8)
}
