#!/bin/bash
# build_rb.sh [name] [-D...]: standalone check + timing binary of csrc/rowblock.hip -> scripts/ubench/rb_<name>.bin
set -e
cd "$(dirname "$0")"
name=${1:-base}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" rowblock_bench.hip -o rb_${name}.bin 2>&1 | grep -E " error|undefined" || true
ls -la rb_${name}.bin
