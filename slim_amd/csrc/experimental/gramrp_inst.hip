// cd_gramrp_kernel (cd_gramrp.hpp): the pipelined form of the item-space kernel; see gramr_inst.hpp
#include "cd_gramrp.hpp"  // (compile with -Iexperimental -I.)
#include "gramr_inst.hpp"
namespace slimamd {
// 12 full groups + a tail group of 128 threads: up to 100 352 items (the 1M x 100K configuration)
GramrFn gramrp_kernel_t128(size_t* lds_bytes) {
  *lds_bytes = (size_t)gramrp_lds_bytes(2, 128, 3);
  return cd_gramrp_kernel<10, 2, 128, 3>;
}
}  // namespace slimamd
