#!/bin/bash
# Experiment build: the plain-GEMM entry point with the 64x64 loop variants (DD_V64 = 0 current,
# 1 interleaved split, 2 two accumulator chains, 3 both) and the ablations (no split / no MFMA /
# one MFMA) as separate small libraries under /tmp/exp64, for tools/exp64.py on the GPU box.
set -e
cd "$(dirname "$0")/../daydreamer_amd/csrc"
mkdir -p ../../tools/exp64_libs
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
build() { hipcc $FL $2 -shared capi.hip gemm.hip -o ../../tools/exp64_libs/lib_$1.so; }
build var "-DDD_EXP64" &
build nosplit "-DDD_ABL_NOSPLIT" &
build nomfma "-DDD_ABL_NOMFMA" &
build onemfma "-DDD_ABL_ONEMFMA" &
wait
ls -la ../../tools/exp64_libs
