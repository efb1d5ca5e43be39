// record_load_probe.hip — how should a lane fetch a 64-B interior record?
//
// The traversal kernel's interior step loads one 64-B record per lane with four
// global_load_dwordx4 (each instruction touches 64 different cache lines, and the four hit the
// same line per lane).  Variant `quad` lets the four lanes of a quad fetch ONE record per
// instruction (16 B each, one 64-B segment per quad), so a record costs one line lookup
// instead of four, and hands the pieces round with DPP quad permutes.  This probe measures a
// dependent chase over `nRec` records (next index = a word of the record just loaded), 256
// threads x `blocksPerCu` blocks per CU, for footprints from L2-resident to HBM-resident.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/record_load_probe.hip -o tools/record_load_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_lane(const float4 *__restrict__ rec, unsigned nRec, int steps,
                                              unsigned *out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nRec;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        const float4 *r = rec + 4ul * idx;
        const float4 q3 = r[3], q0 = r[0], q1 = r[1], q2 = r[2];
        acc += __float_as_uint(q0.x) ^ __float_as_uint(q1.y) ^ __float_as_uint(q2.z);
        idx = __float_as_uint(q3.w);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + idx;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
// quad_perm:[a,b,c,d] control word = a | b<<2 | c<<4 | d<<6
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

// 4x4 transpose inside each quad: in: v[j] of lane c = piece c of ray j's record (one dword of
// it); out: v[c] of lane j = the same dword of piece c of ITS record.  Two exchange stages.
__device__ __forceinline__ void quad_transpose(unsigned v[4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    // stage 1: lanes differing in bit 0 exchange (v0 <-> v1), (v2 <-> v3)
    unsigned a0 = dpp<QP(1, 0, 3, 2)>(b0 ? v[0] : v[1]);
    unsigned a2 = dpp<QP(1, 0, 3, 2)>(b0 ? v[2] : v[3]);
    if (b0) {
        v[0] = a0;
        v[2] = a2;
    } else {
        v[1] = a0;
        v[3] = a2;
    }
    // stage 2: lanes differing in bit 1 exchange (v0 <-> v2), (v1 <-> v3)
    unsigned c0 = dpp<QP(2, 3, 0, 1)>(b1 ? v[0] : v[2]);
    unsigned c1 = dpp<QP(2, 3, 0, 1)>(b1 ? v[1] : v[3]);
    if (b1) {
        v[0] = c0;
        v[1] = c1;
    } else {
        v[2] = c0;
        v[3] = c1;
    }
}

__global__ __launch_bounds__(256) void k_quad(const float4 *__restrict__ rec, unsigned nRec, int steps,
                                              unsigned *out, int transposeAll) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 3;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nRec;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        // instruction j: the quad fetches the record of its lane j, 16 B per lane
        const unsigned i0 = dpp<QP(0, 0, 0, 0)>(idx), i1 = dpp<QP(1, 1, 1, 1)>(idx),
                       i2 = dpp<QP(2, 2, 2, 2)>(idx), i3 = dpp<QP(3, 3, 3, 3)>(idx);
        const float4 p0 = rec[4ul * i0 + c], p1 = rec[4ul * i1 + c], p2 = rec[4ul * i2 + c],
                     p3 = rec[4ul * i3 + c];
        unsigned x[4] = {__float_as_uint(p0.x), __float_as_uint(p1.x), __float_as_uint(p2.x), __float_as_uint(p3.x)};
        unsigned y[4] = {__float_as_uint(p0.y), __float_as_uint(p1.y), __float_as_uint(p2.y), __float_as_uint(p3.y)};
        unsigned z[4] = {__float_as_uint(p0.z), __float_as_uint(p1.z), __float_as_uint(p2.z), __float_as_uint(p3.z)};
        unsigned w[4] = {__float_as_uint(p0.w), __float_as_uint(p1.w), __float_as_uint(p2.w), __float_as_uint(p3.w)};
        quad_transpose(w, lane);
        if (transposeAll) {
            quad_transpose(x, lane);
            quad_transpose(y, lane);
            quad_transpose(z, lane);
            acc += x[0] ^ y[1] ^ z[2];  // q0.x ^ q1.y ^ q2.z of this lane's record
        } else {
            acc += x[0] ^ y[1] ^ z[2];
        }
        idx = w[3];  // q3.w of this lane's record
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + idx;
}

int main() {
    const int steps = 2000;
    for (size_t mb : {2, 32, 200}) {
        const unsigned nRec = (unsigned)(mb * 1024 * 1024 / 64);
        std::vector<unsigned> h((size_t)nRec * 16);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < h.size(); ++i) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            h[i] = (unsigned)(s >> 20) % nRec;
        }
        float4 *d;
        unsigned *out;
        if (hipMalloc(&d, h.size() * 4) != hipSuccess) return 1;
        (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int blocksPerCu : {6, 4}) {
            const int blocks = 256 * blocksPerCu;
            (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
            std::vector<unsigned> ref, got((size_t)blocks * 256);
            for (int variant = 0; variant < 3; ++variant) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEvent_t a, b;
                    (void)hipEventCreate(&a);
                    (void)hipEventCreate(&b);
                    (void)hipEventRecord(a);
                    if (variant == 0)
                        hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(256), 0, 0, d, nRec, steps, out);
                    else
                        hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(256), 0, 0, d, nRec, steps, out, variant == 2);
                    (void)hipEventRecord(b);
                    (void)hipEventSynchronize(b);
                    float ms = 0;
                    (void)hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                (void)hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost);
                if (variant == 0) ref = got;
                const bool same = variant != 2 || got == ref;
                const double recs = (double)blocks * 256 * steps;
                std::printf("%4zu MB  %d blocks/CU  %-22s %8.3f ms  %7.2f G records/s  %s\n", mb, blocksPerCu,
                            variant == 0 ? "per-lane 4 x dwordx4" : (variant == 1 ? "quad, w transposed" : "quad, all transposed"),
                            best, recs / best / 1e6, same ? "" : "MISMATCH");
            }
            (void)hipFree(out);
        }
        (void)hipFree(d);
    }
    return 0;
}
