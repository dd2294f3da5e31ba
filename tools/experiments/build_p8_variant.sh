#!/bin/bash
# Developer tool: scratch build of the library with gemm_p8.hip compiled with extra flags (the timing ablations of profiles/r04_gemm_p8_ablation.txt:
# -DP8_ABL=<bits>, -DDC_GELU_VARIANT=1), every other object taken from tools/build_dev.sh (run that first).
# usage: tools/experiments/build_p8_variant.sh <name> <flags...>  -> tools/ab/libdc_p8_<name>.so ; then DC_LIB_PATH=... DC_GEMM_P8=2 python tools/bench_gemm.py 32
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
PKG=diffcodec-controlling-latent-diffusion-for-perceptual-video-compression_amd
mkdir -p tools/ab/p8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DDC_DEV_KNOBS "$@" -c $PKG/csrc/gemm_p8.hip -o tools/ab/p8/gemm_p8_$name.o
objs=$(ls tools/ab/obj/*.o | grep -v gemm_p8.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdc_p8_$name.so $objs tools/ab/p8/gemm_p8_$name.o
echo tools/ab/libdc_p8_$name.so
