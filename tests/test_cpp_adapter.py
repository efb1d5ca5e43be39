"""The C++ adapter (include/nnbvh_aggregate.hpp): compiles with a plain host compiler against
the C-ABI library (CPU check), and on a GPU its per-ray and batched forms agree (gpu check)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_check")


def _build(nnbvh_lib):
    src = os.path.join(ROOT, "tests", "cpp", "adapter_check.cpp")
    libdir = os.path.join(ROOT, "nn_bvh_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    src, "-o", EXE, "-pthread", "-L", libdir, "-l:libnnbvh_hip.so", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath-link,/opt/rocm/lib"], check=True)


def test_adapter_compiles_with_host_compiler_only(nnbvh_lib):
    _build(nnbvh_lib)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_adapter_per_ray_and_batched_agree(nnbvh_lib):
    _build(nnbvh_lib)
    out = subprocess.run([EXE], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter ok" in out.stdout


C_DEMO = os.path.join(ROOT, "examples", "trace_demo")


def _build_c_demo():
    libdir = os.path.join(ROOT, "nn_bvh_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "trace_demo.c"), "-o", C_DEMO, "-L", libdir,
                    "-l:libnnbvh_hip.so", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-lm"],
                   check=True)


def test_header_is_plain_c_and_demo_builds(nnbvh_lib):
    """include/nnbvh.h must compile as C11 (the ABI claim), and the pure-C example must link."""
    _build_c_demo()
    out = subprocess.run([C_DEMO], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2048 triangles" in out.stdout


@pytest.mark.gpu
def test_c_demo_traces_on_gpu(nnbvh_lib):
    _build_c_demo()
    out = subprocess.run([C_DEMO], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ray 0: prim" in out.stdout and "t 5" in out.stdout
