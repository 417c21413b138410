// cd_gram_kernel<NW, V> and gram_union_kernel; see gram_inst.hpp
#include "cd_gram.hpp"
#include "gram_inst.hpp"
namespace slimamd {
bool gram_geometry(int ncols_pad, int* nw, int* v) {
  const int n4 = ncols_pad / 4;
  if (ncols_pad > kGramMaxColsPad) {  // g in HBM (cd_gram.hpp, V = 0)
    *nw = 8;  // (two 8-wavefront workgroups per CU, 8 visit slots per lane: the whole C4 matrix in
              // 125 s of kernel against 137-146 s with one workgroup of 16 slots and 144 s with four
              // 4-wavefront workgroups)
    *v = 0;
    return true;
  }
  if (n4 <= 2 * 256) { *nw = 4; *v = 2; }
  else if (n4 <= 2 * 512) { *nw = 8; *v = 2; }
  else if (n4 <= 2 * 1024) { *nw = 16; *v = 2; }
  else if (n4 <= 5 * 1024) { *nw = 16; *v = 5; }
  else { *nw = 16; *v = 10; }
  return true;
}
KernelFn gram_kernel(int nw, int v) {
  if (nw == 4 && v == 2) return cd_gram_kernel<4, 2>;
  if (nw == 8 && v == 2) return cd_gram_kernel<8, 2>;
  if (nw == 16 && v == 2) return cd_gram_kernel<16, 2>;
  if (nw == 16 && v == 5) return cd_gram_kernel<16, 5>;
  if (nw == 16 && v == 10) return cd_gram_kernel<16, 10>;
  if (nw == 16 && v == 0) return cd_gram_kernel<16, 0>;
  if (nw == 8 && v == 0) return cd_gram_kernel<8, 0>;
  return nullptr;
}
KernelFn gram_union_fn() { return gram_union_kernel; }
}  // namespace slimamd
