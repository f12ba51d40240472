#!/bin/bash
# Builds libmslam_ablate<N>.so (N = 1, 2, 3): the product library with the GEMM translation units compiled under
# -DMSLAM_GEMM_ABLATE=N (see csrc/gemm_kernel.h).  Results of these libraries are WRONG by construction; they exist to
# time the parts of the K loop (tools/gemm_cfg_ab.py with MSLAM_LIB=...).
set -e
cd "$(dirname "$0")/../../mast3r-slam-quality-dualtsdf_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
for n in ${ABLATE_SET:-1 2 3 4}; do
  ( for f in gemm_t64 gemm_t128 gemm_t256; do /opt/rocm/bin/hipcc $FLAGS -DMSLAM_GEMM_ABLATE=$n -c $f.hip -o /tmp/${f}_abl$n.o; done
    OBJS=$(ls *.o | grep -v "gemm_t" | tr '\n' ' ')
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemm_t64_abl$n.o /tmp/gemm_t128_abl$n.o /tmp/gemm_t256_abl$n.o -o ../../tools/probes/libmslam_ablate$n.so ) &
done
wait
ls -la ../../tools/probes/*.so
