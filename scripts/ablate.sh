#!/bin/bash
# Timing-only ablation builds of the critic kernel (outputs are wrong by construction; only the time matters).
set -e
cd "$(dirname "$0")/../relearn_amd/csrc"
mkdir -p _build/abl
for a in 1 2 4 3 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DRL_ABLATE=$a -c kernels_mfma.hip -o _build/abl/kernels_mfma_$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _build/abi.o _build/abi_update.o _build/abi_cbor.o _build/abi_dqn.o _build/host_abi.o _build/kernels_rollout.o _build/kernels_update.o _build/kernels_dqn.o _build/kernels_seq.o _build/kernels_seq_bwd.o _build/kernels_seq_fvp.o _build/abl/kernels_mfma_$a.o -o _build/abl/librelearn_abl_$a.so -ldl -lpthread
done
