#!/bin/bash
# AddressSanitizer pass over the CPU-side code (no GPU needed): the oracle (gcc, -fsanitize=address,undefined) against its
# golden vectors, and the host half of the C ABI (hipcc, host code only) against the argument-validation tests.
set -eu
cd "$(dirname "$0")/.."
make -C oracle -s asan
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 CS_ORACLE_LIB=$PWD/oracle/libstereo_oracle_asan.so \
    python -m pytest tests/test_oracle_goldens.py tests/test_oracle_math.py tests/test_dialect_oracle.py -q -x
make -C comfystereo_amd/csrc -s asan
LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
    CS_LIB_PATH=$PWD/comfystereo_amd/libcomfystereo_hip_asan.so python -m pytest tests/test_abi_exports.py -q -x
echo "asan_check OK"
