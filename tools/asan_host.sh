#!/bin/bash
# Host-side AddressSanitizer run of libsmx (panel compiler, window packer, streaming I/O) through the Python binding.
# CPU only (device code is compiled normally; GPU ASan is not available on this pool).  ~1 minute.
#   tools/asan_host.sh            -> builds build_exp/libsmx_asan.so and runs the CPU host tests against it
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_exp"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -shared --offload-arch=gfx950 -fsanitize=address -fno-omit-frame-pointer \
    -Wno-option-ignored -I"$ROOT/include" -I"$ROOT/specimux_amd/csrc" -x hip \
    "$ROOT/specimux_amd/csrc/smx_kernels.hip" "$ROOT/specimux_amd/csrc/smx_api.cpp" "$ROOT/specimux_amd/csrc/smx_io.cpp" \
    -o "$ROOT/build_exp/libsmx_asan.so" -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 SMX_LIB="$ROOT/build_exp/libsmx_asan.so" \
    python -m pytest tests/test_host_cpu.py tests/test_native_io_cpu.py -x -q
