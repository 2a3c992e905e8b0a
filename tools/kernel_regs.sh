#!/bin/bash
# tools/kernel_regs.sh <host object (.o) or libmmwgpu.so> [name filter]: VGPRs / scratch bytes / static LDS of every kernel in
# the gfx950 code object (the persistent kernels must show scratch=0)
f=$1; pat=${2:-.}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$f" "$tmp/fb" || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fb" --output="$tmp/co" --unbundle || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/co" | awk '
  /\.group_segment_fixed_size:/ {lds=$2}
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {scr=$2}
  /\.vgpr_count:/ {vg=$2}
  /\.wavefront_size:/ {print vg, scr, lds, name}' | while read vg scr lds name; do
    echo "vgpr=$vg scratch=$scr lds=$lds $(echo $name | c++filt | cut -c1-160)"; done | grep -E "$pat" | sort -k4
rm -rf "$tmp"
