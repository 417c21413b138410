// gramr_inst.hpp -- the instantiations of cd_gramr_kernel (cd_gramr.hpp) and the packing kernels
// (gram_pack.hpp); engine.hip picks one through gramr_kernel().
#pragma once
#include "gram_pack.hpp"
#include "tile_inst.hpp"

namespace slimamd {

using GramrFn = void (*)(const DevMatrix, const SolveArgs, const GramPacked);

// the smallest instantiation whose K = KR + KL groups of 8192 ranks cover nchunks 16-rank chunks
// (nullptr: more items than the largest one holds on chip -- 106 496)
// dma: the variant that streams rows through an LDS ring (global_load_lds); lds_bytes = the dynamic
// LDS the chosen instantiation needs
GramrFn gramr_kernel(int nchunks, bool dma, int* kr, int* kl, size_t* lds_bytes);
GramrFn gramr_kernel_k13(bool* dma, int* ring_ah);  // <10, 3>: its own translation unit (compiles beside the others)

// the tiles' union lists read off the byte planes (cd_gramr.hpp: gramr_union_kernel; kGramrUnionNT threads per tile)
GramrFn gramr_union_fn();
int gramr_union_threads();

using PackScanFn = void (*)(const float*, int64_t, int, const int32_t*, int, int, int32_t*, int32_t*, int32_t*);
using PackWriteFn = void (*)(const float*, int64_t, int, const int32_t*, int, uint8_t*, int64_t, uint8_t*,
                             const int64_t*, const int32_t*, uint8_t*, const int64_t*, const int32_t*, uint8_t*, float*);
PackScanFn gram_pack_scan_fn();
PackWriteFn gram_pack_write_fn();
using PackMetaFn = void (*)(int, const int32_t*, const int32_t*, const int32_t*, const int64_t*, const float*,
                            const int64_t*, const float*, const float*, uint4*, int32_t*);
PackMetaFn gram_pack_meta_fn();

}  // namespace slimamd
