"""Ahead-of-time build of the gfx950 shared object (hipcc cross-compiles without a GPU).

    python -m diffcodec_amd.build        -> <package>/libdiffcodec_hip.so

The .so is built in-tree so that it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffcodec_hip.so")
SOURCES = ["igemm.hip", "conv3x3_tile.hip", "gemm_dma.hip", "gemm_wide.hip", "gemm_p8.hip", "gemm_rowpanel.hip", "attention.hip", "norm.hip", "splat.hip", "conv_direct.hip", "conv_f32_mfma.hip", "elementwise.hip", "text.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
# per-file flags: attention.hip keeps its softmax arithmetic as single v_fma_f32 / v_mul_f32 (no compiler-formed v_pk_*_f32 with
# cross-half op_sel operands: the cause of the round-3 d = 16 wrong rows)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "dc_common.h"), os.path.join(os.path.dirname(HERE), "include", "diffcodec_hip.h")]
    jobs = []
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and os.path.exists(LIB) and not _stale(LIB, srcs + hdrs):
        return LIB                                   # prebuilt and newer than every source (e.g. on the GPU box: objects do not travel)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
