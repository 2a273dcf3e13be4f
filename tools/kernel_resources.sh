#!/bin/bash
# Register / spill / LDS usage of every kernel of one .hip file (device ELF notes), e.g.
#   EXTRA="-mllvm -disable-machine-licm" tools/kernel_resources.sh atsc_kernels.hip k_compress   (EXTRA: the per-file flags of atsc_amd/build.py)
set -e
cd "$(dirname "$0")/../atsc_amd/csrc"
f=${1:-atsc_kernels.hip}; pat=${2:-.}
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 --offload-device-only -O3 -ffp-contract=off -std=c++17 $EXTRA -c "$f" -o $tmp/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/k.co --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.elf | grep -E "\.name:|\.sgpr_count|sgpr_spill|\.vgpr_count|vgpr_spill|private_segment_fixed" \
  | paste - - - - - - | sed 's/ \+/ /g; s/\.private_segment_fixed_size/scratch/; s/_ZN4atsc//' | grep -E "$pat" | sed -E "s/EEEvPKd.*UniArgsE/>/; s/(\.name: [0-9]*k_[a-z0-9_]*(I[^E]*E)?)[^\t]*/\1/" | cut -c1-200
[ -n "$KEEP" ] && cp $tmp/k.elf "$KEEP"
rm -rf $tmp
