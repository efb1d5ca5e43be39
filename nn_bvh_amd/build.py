"""Builds nn_bvh_amd/libnnbvh_hip.so (HIP kernels + C ABI + host BVH builder) for gfx950.

In-tree, explicit hipcc invocations: every source is compiled to an object under
nn_bvh_amd/_obj/<variant>/ (in parallel, re-used while it is newer than the source and the
headers) and the objects are linked into the .so, which travels to the GPU box with the repo
snapshot.  -ffp-contract=off is part of the numerical contract.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnnbvh_hip.so")
SOURCES = ["bvh_trace.hip", "wavefront.hip", "wavefront2.hip", "bvh_build_gpu.hip", "bvh_bake.hip", "interaction.hip", "film.hip", "kd_trace.hip", "kd_build_gpu.hip", "bvh_capi.cpp", "bvh_layout.cpp", "bvh_build.cpp", "kd_build.cpp"]
HEADERS = ["bvh_trace.h", "trace_math.h", "spawn_math.h", "anim_math.h", "wavefront2.h", "wavefront.h", "bvh_build_gpu.h", "interaction.h", "nnbvh_internal.h", os.path.join("..", "..", "include", "nnbvh.h")]
CFLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
          "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
FLAGS = CFLAGS + ["-shared"]  # kept for the tools that print the command line


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libnnbvh_hip.so")
    return hipcc


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps_mtime():
    deps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _deps_mtime() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


def _build(out, extra=(), force=False, verbose=False, jobs=None):
    """Compile every source with CFLAGS + extra into objects and link them into `out`."""
    hipcc = _hipcc()
    tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10] if extra else "product"
    objdir = os.path.join(HERE, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _deps_mtime()
    todo, objs = [], []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            todo.append([hipcc] + CFLAGS + list(extra) + ["-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1), 6)
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_stats(verbose=False):
    """Diagnostic build with scheduling statistics compiled in (libnnbvh_hip_stats.so)."""
    return _build(os.path.join(HERE, "libnnbvh_hip_stats.so"), ["-DNNBVH_STATS"], verbose=verbose)


def build_variant(name, defines, verbose=False, flags=()):
    """Experimental build nn_bvh_amd/libnnbvh_hip_<name>.so with extra -D flags (and compiler flags; select
    it with NNBVH_LIB=libnnbvh_hip_<name>.so); never the product."""
    return _build(os.path.join(HERE, f"libnnbvh_hip_{name}.so"), list(flags) + [f"-D{d}" for d in defines],
                  verbose=verbose)


def build(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not _stale():
        return LIB
    return _build(LIB, force=force, verbose=verbose)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:  # python -m nn_bvh_amd.build NAME DEFINE...
        print(build_variant(sys.argv[1], sys.argv[2:], verbose=True))
    else:
        print(build(force=True, verbose=True))
