#!/bin/bash
# Experiment build: the plain-GEMM entry point with a 256x128 workgroup tile (wave tile 128x64, one
# wave per SIMD: 220 VGPRs + 128 AGPRs, 75-80 KB of LDS) selectable by DD_FORCE_TILE=256x128, as a
# small library under tools/exp_big_libs for tools/exp_big.py on the GPU box.
set -e
cd "$(dirname "$0")/../daydreamer_amd/csrc"
mkdir -p ../../tools/exp_big_libs
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
hipcc $FL -DDD_EXP_BIG -shared capi.hip gemm.hip -o ../../tools/exp_big_libs/lib_big.so
ls -la ../../tools/exp_big_libs
