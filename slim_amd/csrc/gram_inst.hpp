// gram_inst.hpp -- the instantiations of cd_gram_kernel (cd_gram.hpp); engine.hip picks one
// through gram_kernel().
#pragma once
#include "tile_inst.hpp"

namespace slimamd {

// workgroup geometry for a matrix of ncols_pad item columns: nw wavefronts, v float4 of a row of
// G per thread (4 * v * 64 * nw >= ncols_pad), g in LDS; v = 0 when g does not fit the LDS of a
// CU and lives in HBM instead
bool gram_geometry(int ncols_pad, int* nw, int* v);
KernelFn gram_kernel(int nw, int v);
KernelFn gram_union_fn();
// static LDS of the kernel on top of the 4 * ncols_pad bytes of g
constexpr int kGramStaticLds = 512;
constexpr int kGramMaxColsPad = (160 * 1024 - kGramStaticLds) / 4 / 64 * 64;

}  // namespace slimamd
