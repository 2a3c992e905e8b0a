#!/bin/bash
# tools/kernel_asm.sh <host object> <mangled-name regex> [grep pattern]: disassembly of one gfx950 kernel (default: the memory
# instructions, waits and barriers -- where a load is followed by its own s_waitcnt, prefetching is not happening)
f=$1; name=$2; pat=${3:-"s_barrier|buffer_load|global_load|s_waitcnt vmcnt|global_atomic|global_store|buffer_store"}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$f" "$tmp/fb" || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fb" --output="$tmp/co" --unbundle || exit 1
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$tmp/co" | awk -v n="$name" '/^[0-9a-f]+ <.*>:/{on = ($0 ~ n)} on' | sed 's#//.*##' | grep -nE "$pat"
rm -rf "$tmp"
