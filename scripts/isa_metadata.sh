#!/bin/bash
# Register / spill figures of every tile-kernel instantiation and the per-loop spill remarks of the
# dominant one (cross-compiled here, no GPU needed):  bash scripts/isa_metadata.sh > profiles/rNN/isa_metadata.txt
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
echo "# hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage, $(/opt/rocm/bin/hipcc --version | grep -m1 -i 'hip version')"
echo "# cd_tile_kernel<P, HAS_VAL, PROFILE, NW, FSLIM, FOLD>, cd_gram_kernel<NW, V>, cd_gramr_kernel<KR, KL, DMA> : VGPRs, SGPRs, VGPR spills, SGPR spills (to VGPR lanes), scratch bytes per lane, LDS bytes"
for f in tile_p32_nw16 tile_p32_nw8 tile_p32_cold tile_p32_rowfold tile_p32_fslim tile_p16_nw16 tile_p16_nw8 gram_inst gramr_inst gramr_k13; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage -c -o $T/x.o $R/slim_amd/csrc/$f.hip 2>&1 |
  grep -E "Function Name|VGPRs:|TotalSGPRs|VGPRs Spill|SGPRs Spill|ScratchSize|LDS Size" | sed 's/.*remark: *//;s/ \[-Rpass.*//' |
  awk '/Function Name/{if (n) print n, v; n=$3; v=""; next} {v=v" | "$0} END{print n, v}' |
  sed 's/_ZN7slimamd14cd_tile_kernelI/cd_tile_kernel</;s/_ZN7slimamd14cd_gram_kernelI/cd_gram_kernel</;s/_ZN7slimamd15cd_gramr_kernelI/cd_gramr_kernel</;s/EEvNS_9DevMatrixENS_9SolveArgsENS_10GramPackedE/>/;s/EEvNS_9DevMatrixENS_9SolveArgsE/>/;s/Li\([0-9]*\)E/\1,/g;s/Lb0E/false,/g;s/Lb1E/true,/g;s/,>/>/'
done
cat > $T/one.hip <<EOT
#include "tile_inst.hpp"
namespace slimamd { KernelFn one() { return cd_tile_kernel<32, false, false, 16, false, 0>; } }
EOT
echo
echo "# cd_tile_kernel<32,false,false,16,false,0> (the C4 / C5 cold-start kernel): spills and reloads by loop (-Rpass-missed=regalloc;"
echo "# two lines per loop = the SGPR and the VGPR allocation pass; the visit loop is the 'for (int p ...' loop inside the sweep loop)"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$R/slim_amd/csrc --cuda-device-only \
  -Rpass-missed=regalloc -c -o $T/one.o $T/one.hip 2>&1 | grep remark | sed 's/.*csrc\///;s/ \[-Rpass.*//' | grep -E "spills|reloads" |
  sort -t: -k2 -n | awk -F: '{print}' | cut -c1-200
L1=$(grep -n "for (int t = 0;; ++t) {" $R/slim_amd/csrc/cd_tile.hpp | cut -d: -f1)
L2=$(grep -n "for (int p = 0; p < nunion; ++p) {" $R/slim_amd/csrc/cd_tile.hpp | tail -1 | cut -d: -f1)
echo "# (sweep loop: cd_tile.hpp:$L1, visit loop: cd_tile.hpp:$L2)"
cat > $T/two.hip <<EOT
#include "gramr_inst.hpp"
#include "cd_gramr.hpp"
namespace slimamd { GramrFn two() { return cd_gramr_kernel<10, 3, true>; } }
EOT
echo
echo "# cd_gramr_kernel<10,3,true> (C4 in item space): spills and reloads by loop"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$R/slim_amd/csrc --cuda-device-only \
  -Rpass-missed=regalloc -c -o $T/two.o $T/two.hip 2>&1 | grep remark | grep "cd_gramr" | sed 's/.*csrc\///;s/ \[-Rpass.*//' | grep -E "spills|reloads" |
  sort -t: -k2 -n | cut -c1-200
rm -rf $T
