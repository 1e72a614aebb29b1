#!/bin/bash
# Experiment build: the wave-specialised contraction loop (tools/exp_ws/gemm_ws.hip) next to the
# product loop, as a small library under tools/exp_ws/libs for tools/exp_ws.py on the GPU box.
set -e
cd "$(dirname "$0")/../daydreamer_amd/csrc"
mkdir -p ../../tools/exp_ws/libs
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I. -I../../include"
hipcc $FL -shared capi.hip ../../tools/exp_ws/gemm_ws.hip -o ../../tools/exp_ws/libs/lib_ws.so
ls -la ../../tools/exp_ws/libs
