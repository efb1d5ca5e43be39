"""AddressSanitizer + UBSan on the host-side builder (CPU build only; GPU sanitizers are not
available on the pool): build bvh_build.cpp with g++ -fsanitize=address,undefined into a small
driver and run every split method over degenerate and ordinary inputs."""
import os
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r'''
    #include <cstdio>
    #include <random>
    #include <string>
    #include <vector>
    #include "nnbvh.h"
    #include "bvh_build_gpu.h"
    namespace nnbvh { void set_error(const std::string &m) { std::fprintf(stderr, "err: %s\n", m.c_str()); }
    // the device pipeline lives in bvh_build_gpu.hip, which a CPU-only sanitizer build cannot link
    bool gpu_hlbvh(const nnbvh_prim *, int, const float *, int, const float *, int, int, GpuBuildResult *,
                   std::string *e) { *e = "no device code in this build"; return false; }
    bool gpu_sah(const nnbvh_prim *, int, const float *, int, const float *, int, int, GpuBuildResult *,
                 std::string *e) { *e = "no device code in this build"; return false; } }
    int main() {
        std::mt19937 rng(3);
        std::uniform_real_distribution<float> U(-1.f, 1.f);
        for (int n : {1, 2, 3, 7, 500, 5000}) {
            std::vector<float> verts;
            std::vector<nnbvh_prim> prims;
            for (int i = 0; i < n; ++i) {
                float c[3] = {10 * U(rng), 10 * U(rng), 10 * U(rng)};
                if (i % 50 == 7) c[0] = c[1] = c[2] = 0;  // coincident clusters
                bool patch = (i % 11 == 3);
                int nv = patch ? 4 : 3, base = (int)verts.size() / 3;
                for (int k = 0; k < nv; ++k)
                    for (int a = 0; a < 3; ++a) verts.push_back(c[a] + (i % 50 == 7 ? 0.f : 0.3f * U(rng)));
                prims.push_back(nnbvh_prim{patch ? 1 : 0, i, {base, base + 1, base + 2, patch ? base + 3 : 0}});
            }
            for (int method : {NNBVH_SPLIT_SAH, NNBVH_SPLIT_HLBVH, NNBVH_SPLIT_MIDDLE, NNBVH_SPLIT_EQUAL_COUNTS})
                for (int maxp : {1, 4, 255}) {
                    nnbvh_build *b = nnbvh_build_create(prims.data(), n, verts.data(), (int)verts.size() / 3, maxp, method);
                    if (!b) continue;  // HLBVH may refuse degenerate inputs the reference aborts on
                    int nn = 0, np = 0;
                    const nnbvh_linear_node *nodes = nnbvh_build_nodes(b, &nn);
                    const nnbvh_prim *op = nnbvh_build_ordered_prims(b, &np);
                    long sum = 0;
                    for (int i = 0; i < nn; ++i) sum += nodes[i].offset + nodes[i].nprims;
                    for (int i = 0; i < np; ++i) sum += op[i].id;
                    if (np != n || nn < 1 || sum < 0) return 1;
                    nnbvh_build_destroy(b);
                }
        }
        // non-finite coordinates are rejected, never indexed with (round 1: SIGSEGV in the bucket loop)
        for (float bad : {1.0f / 0.0f, -1.0f / 0.0f, 0.0f / 0.0f}) {
            std::vector<float> verts;
            std::vector<nnbvh_prim> prims;
            for (int i = 0; i < 64; ++i) {
                int base = (int)verts.size() / 3;
                for (int k = 0; k < 9; ++k) verts.push_back(5 * U(rng));
                prims.push_back(nnbvh_prim{0, i, {base, base + 1, base + 2, 0}});
            }
            verts[3 * 40 + 1] = bad;
            for (int method : {NNBVH_SPLIT_SAH, NNBVH_SPLIT_HLBVH, NNBVH_SPLIT_MIDDLE, NNBVH_SPLIT_EQUAL_COUNTS})
                if (nnbvh_build_create(prims.data(), 64, verts.data(), (int)verts.size() / 3, 4, method)) return 6;
        }
        // the host half of the GPU HLBVH build: upper SAH tree + DFS layout over treelet roots
        // (3392 = the shape of the round-1 bring-up abort: thousands of treelets of ONE leaf each)
        for (int nt : {1, 2, 37, 4096, 3392}) {
            std::vector<float> tb(6 * (size_t)nt);
            std::vector<int> ts(nt);
            for (int t = 0; t < nt; ++t) {
                for (int a = 0; a < 3; ++a) {
                    tb[6 * t + a] = 50 * U(rng);
                    tb[6 * t + 3 + a] = tb[6 * t + a] + 1 + U(rng) * 0.5f;
                }
                ts[t] = nt == 3392 ? 1 : 1 + 2 * (int)(rng() % 9);
            }
            nnbvh::UpperLayout up;
            std::string err;
            if (!nnbvh::hlbvh_upper_layout(tb.data(), ts.data(), nt, &up, &err)) return 2;
            long want = nt - 1;
            for (int t = 0; t < nt; ++t) want += ts[t];
            if (up.total_nodes != want || (int)up.upper_index.size() != nt - 1) return 3;
            std::vector<char> used((size_t)up.total_nodes, 0);
            for (int t = 0; t < nt; ++t)
                for (int k = 0; k < ts[t]; ++k) {
                    if (used[(size_t)up.base[t] + k]) return 4;  // treelet regions must not overlap
                    used[(size_t)up.base[t] + k] = 1;
                }
            for (int i : up.upper_index) {
                if (used[(size_t)i]) return 5;
                used[(size_t)i] = 1;
            }
        }
        std::puts("sanitized builder ok");
        return 0;
    }
''')


def test_builder_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "nn_bvh_amd", "csrc"),
                    str(src), os.path.join(ROOT, "nn_bvh_amd", "csrc", "bvh_build.cpp"), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitized builder ok" in out.stdout


TSAN_DRIVER = textwrap.dedent(r'''
    #include <cstdio>
    #include <cstring>
    #include <random>
    #include <string>
    #include <vector>
    #include "nnbvh.h"
    #include "bvh_build_gpu.h"
    namespace nnbvh { void set_error(const std::string &m) { std::fprintf(stderr, "err: %s\n", m.c_str()); }
    // the device pipeline lives in bvh_build_gpu.hip, which a CPU-only sanitizer build cannot link
    bool gpu_hlbvh(const nnbvh_prim *, int, const float *, int, const float *, int, int, GpuBuildResult *,
                   std::string *e) { *e = "no device code in this build"; return false; }
    bool gpu_sah(const nnbvh_prim *, int, const float *, int, const float *, int, int, GpuBuildResult *,
                 std::string *e) { *e = "no device code in this build"; return false; } }
    int main() {
        // 300 000 primitives: above the 128 K threshold, so sub-trees are built by separate threads
        const int n = 300000;
        std::mt19937 rng(5);
        std::uniform_real_distribution<float> U(-1.f, 1.f);
        std::vector<float> verts;
        std::vector<nnbvh_prim> prims;
        for (int i = 0; i < n; ++i) {
            float c[3] = {20 * U(rng), 20 * U(rng), 20 * U(rng)};
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) verts.push_back(c[a] + 0.1f * U(rng));
            prims.push_back(nnbvh_prim{0, i, {3 * i, 3 * i + 1, 3 * i + 2, 0}});
        }
        nnbvh_build *a = nnbvh_build_create(prims.data(), n, verts.data(), 3 * n, 4, NNBVH_SPLIT_SAH);
        nnbvh_build *b = nnbvh_build_create(prims.data(), n, verts.data(), 3 * n, 4, NNBVH_SPLIT_SAH);
        int na = 0, nb = 0;
        const nnbvh_linear_node *A = nnbvh_build_nodes(a, &na), *B = nnbvh_build_nodes(b, &nb);
        bool same = na == nb && !std::memcmp(A, B, sizeof(nnbvh_linear_node) * na);
        nnbvh_build_destroy(a);
        nnbvh_build_destroy(b);
        std::puts(same ? "threaded builder deterministic" : "MISMATCH");
        return same ? 0 : 1;
    }
''')


def test_threaded_builder_under_tsan(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(TSAN_DRIVER)
    exe = tmp_path / "driver"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "nn_bvh_amd", "csrc"), str(src),
                    os.path.join(ROOT, "nn_bvh_amd", "csrc", "bvh_build.cpp"), "-o", str(exe), "-lpthread"],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "threaded builder deterministic" in out.stdout and "ThreadSanitizer" not in out.stderr
