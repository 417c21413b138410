// cd_tile_kernel<32, *, *, 16>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE(tile_kernel_p32_nw16, 32, 16)
}
