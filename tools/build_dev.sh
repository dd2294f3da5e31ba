#!/bin/bash
# Developer tool: scratch build of the library with the launcher A/B knobs compiled in (-DDC_DEV_KNOBS: DC_KNOB reads the
# environment) -> tools/ab/libdc_dev.so (git-ignored; travels to the GPU box).  tools/*.py pick it up with DC_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
PKG=diffcodec-controlling-latent-diffusion-for-perceptual-video-compression_amd
mkdir -p tools/ab/obj
pids=()
for f in igemm conv3x3_tile gemm_dma gemm_wide gemm_p8 gemm_rowpanel attention norm splat conv_direct conv_f32_mfma elementwise text; do
  if [ ! -f tools/ab/obj/$f.o ] || [ $PKG/csrc/$f.hip -nt tools/ab/obj/$f.o ] || [ $PKG/csrc/dc_common.h -nt tools/ab/obj/$f.o ]; then
    extra=""; [ $f = attention ] && extra="-fno-slp-vectorize"      # as diffcodec_amd/build.py EXTRA_FLAGS
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DDC_DEV_KNOBS $extra "$@" -c $PKG/csrc/$f.hip -o tools/ab/obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdc_dev.so tools/ab/obj/*.o
echo tools/ab/libdc_dev.so
