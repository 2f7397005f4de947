#!/bin/bash
# usage: tools/kernel_isa.sh <object stem, e.g. conv> -> .scratch/<stem>.s (gfx950 disassembly of the product object)
cd "$(dirname "$0")/.." && mkdir -p .scratch && B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section=.hip_fatbin=.scratch/$1.fat build/obj/$1.o && \
$B/clang-offload-bundler --unbundle --type=o --input=.scratch/$1.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=.scratch/$1.co && \
$B/llvm-objdump -d --no-show-raw-insn .scratch/$1.co > .scratch/$1.s
