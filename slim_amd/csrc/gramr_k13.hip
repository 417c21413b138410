// cd_gramr_kernel<10, 3>: up to 106 496 items (the 1M x 100K configuration); see gramr_inst.hpp
#include "cd_gramr.hpp"
#include "gramr_inst.hpp"
namespace slimamd {
// ONE form: rows through the LDS ring, three groups ahead.  The register-load form and the
// two-ahead ring of this instantiation gave wrong models on 50 000 - 100 000 items
// (scripts/gramr_k13_check.py against the tile kernel; the same source is right for <1,0>, <3,0>, <6,0> and
// for this form) and were never the default: they are not built.
GramrFn gramr_kernel_k13() { return cd_gramr_kernel<10, 3, true, 2, 3>; }
}  // namespace slimamd
