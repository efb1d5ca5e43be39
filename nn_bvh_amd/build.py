"""Builds nn_bvh_amd/libnnbvh_hip.so (HIP kernels + C ABI + host BVH builder) for gfx950.

In-tree, explicit hipcc invocation: the .so travels to the GPU box with the repo
snapshot.  -ffp-contract=off is part of the numerical contract.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnnbvh_hip.so")
SOURCES = ["bvh_trace.hip", "wavefront.hip", "wavefront2.hip", "bvh_build_gpu.hip", "bvh_bake.hip", "interaction.hip", "film.hip", "kd_trace.hip", "bvh_capi.cpp", "bvh_build.cpp", "kd_build.cpp"]
HEADERS = ["bvh_trace.h", "trace_math.h", "spawn_math.h", "anim_math.h", "wavefront2.h", "wavefront.h", "bvh_build_gpu.h", "interaction.h", "nnbvh_internal.h", os.path.join("..", "..", "include", "nnbvh.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_stats(verbose=False):
    """Diagnostic build with scheduling statistics compiled in (libnnbvh_hip_stats.so)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(HERE, "libnnbvh_hip_stats.so")
    cmd = [hipcc] + FLAGS + ["-DNNBVH_STATS", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


def build_variant(name, defines, verbose=False, flags=()):
    """Experimental build nn_bvh_amd/libnnbvh_hip_<name>.so with extra -D flags (and compiler flags; select
    it with NNBVH_LIB=libnnbvh_hip_<name>.so); never the product."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(HERE, f"libnnbvh_hip_{name}.so")
    cmd = [hipcc] + FLAGS + list(flags) + [f"-D{d}" for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


def build(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libnnbvh_hip.so")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
