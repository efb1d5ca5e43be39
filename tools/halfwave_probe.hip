// halfwave_probe.hip — does a gfx950 SIMD skip the half of a wave64 VALU instruction whose 32
// lanes are all masked off?  (It decides whether packing a wavefront's active lanes into one
// half can buy anything in the traversal kernel.)  Build: hipcc --offload-arch=gfx950 -O3
// tools/halfwave_probe.hip -o gpurun_out/halfwave_probe ; run on the GPU box.
//
// Every wave runs the same dependent-FMA streams; variants differ only in the exec mask:
//   full   all 64 lanes          low32  lanes 0..31       even  every second lane
//   q16    lanes 0..15           hi32   lanes 32..63
// Times are per launch at `waves` waves per SIMD on every CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_probe(float *out, int iters, unsigned long long mask) {
    const int lane = threadIdx.x & 63;
    float a0 = lane * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
          a7 = a0 + 7;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_fmaf(a0, 1.0001f, 0.5f);
            a1 = __builtin_fmaf(a1, 1.0001f, 0.5f);
            a2 = __builtin_fmaf(a2, 1.0001f, 0.5f);
            a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
            a4 = __builtin_fmaf(a4, 1.0001f, 0.5f);
            a5 = __builtin_fmaf(a5, 1.0001f, 0.5f);
            a6 = __builtin_fmaf(a6, 1.0001f, 0.5f);
            a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    const int iters = 20000;
    struct V {
        const char *name;
        unsigned long long mask;
    } variants[] = {{"full", ~0ull},
                    {"low32", 0xffffffffull},
                    {"hi32", 0xffffffff00000000ull},
                    {"even", 0x5555555555555555ull},
                    {"q16", 0xffffull},
                    {"low32+1", 0x1ffffffffull}};
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;  // 256 CUs x (4 waves per block = one per SIMD)
        float *out;
        if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return 1;
        for (auto &v : variants) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, out, iters, v.mask);  // warm-up
            hipEventRecord(a);
            hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, out, iters, v.mask);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            // cycles per wave-instruction per SIMD at 2.4 GHz
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wavesPerSimd);
            std::printf("waves/SIMD %d  %-8s %8.3f ms  %.2f cycles per wave-instruction per SIMD\n",
                        wavesPerSimd, v.name, ms, cyc);
        }
        hipFree(out);
    }
    return 0;
}
