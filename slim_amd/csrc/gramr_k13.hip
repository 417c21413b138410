// cd_gramr_kernel<10, 3>: up to 106 496 items (the 1M x 100K configuration); see gramr_inst.hpp
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
// The shipped form: rows through the LDS ring, three groups ahead.  -DSLIM_K13_DMA=0 (register loads)
// and -DSLIM_K13_AH=2 (ring two ahead) build the sibling forms for scripts/gramr_k13_sweep.py (A/B
// libraries via scripts/build_variant.sh); see DESIGN 4.2e for what round 6 found about them.
#ifndef SLIM_K13_DMA
#define SLIM_K13_DMA 1
#endif
#ifndef SLIM_K13_AH
#define SLIM_K13_AH 3
#endif
namespace slimamd {
GramrFn gramr_kernel_k13(bool* dma, int* ring_ah) {
  *dma = SLIM_K13_DMA != 0;
  *ring_ah = SLIM_K13_AH;
  return cd_gramr_kernel<10, 3, (SLIM_K13_DMA != 0), 2, SLIM_K13_AH>;
}
}  // namespace slimamd
