// cd_tile_kernel<32, *, *, 8, false, false, WIDE>: see tile_inst.hpp
#include "tile_inst.hpp"
namespace slimamd {
SLIM_TILE_INSTANTIATE_WIDE(tile_kernel_p32_wide)
}
