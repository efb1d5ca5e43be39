// wavefront2.hip — the bookkeeping kernels of WavefrontAggregate::IntersectShadowTr and
// ::IntersectOneRandom (/root/reference/src/pbrt/wavefront/aggregate.cpp:70-116) for scenes
// without participating media.
//
// Both reference functions are loops of closest-hit traces per work item:
//   IntersectShadowTr -> TraceTransmittance (wavefront/intersect.h:164-274): trace; an opaque hit
//       ends the ray with T = 0; a hit on an interface surface (no material) continues from the hit
//       point towards the light point with SpawnRayTo; without media T_ray, r_u, r_l stay 1, and a
//       ray that arrives adds Ld * (1 / (sr.r_u * 1 + sr.r_l * 1).Average()) to its pixel sample;
//   IntersectOneRandom (aggregate.cpp:90-116): walk the segment p0 -> p1 surface by surface and
//       reservoir-sample one hit whose material matches, WeightedReservoirSampler seeded with
//       Hash(p0, p1).
// On the device one loop iteration is one pass over the still-active items: the trace kernel
// (bvh_trace.hip), the hit -> interaction post-pass (interaction.hip) and the kernels here, which
// classify / update per-item state and write the next segment's rays compacted (one atomic per
// wavefront).  Every kernel is a streaming pass, HBM-bound.
#include <hip/hip_runtime.h>

#include "spawn_math.h"
#include "wavefront2.h"

namespace nnbvh {

static constexpr int kW2Block = 256;

__device__ __forceinline__ int w2_count(WavefrontCount c) {
    int n = c.n;
    if (c.nDev) {
        const int nd = *c.nDev;
        n = nd < 0 ? 0 : (nd < n ? nd : n);
    }
    return n;
}

// wave-aggregated append: returns this lane's slot in the list `counter` counts, or -1
__device__ __forceinline__ int w2_append(bool want, int32_t *counter) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    return want ? base + rank : -1;
}

static int w2_grid(int n, int maxBlocks) {
    int blocks = (n + kW2Block - 1) / kW2Block;
    if (blocks < 1) blocks = 1;
    return blocks < maxBlocks ? blocks : maxBlocks;
}

struct Intr192 {  // nnbvh_interaction as float4s: [0] pi_lo.xyz pi_hi.x [1] pi_hi.yz uv [2] wo time [3] n face
    float4 q[12];
};

__device__ __forceinline__ void w2_read_pi_n(const float4 *rec, V3 &lo, V3 &hi, V3 &n, int &status) {
    const float4 a = rec[0], b = rec[1], c = rec[3], z = rec[11];
    lo = {a.x, a.y, a.z};
    hi = {a.w, b.x, b.y};
    n = {c.x, c.y, c.z};
    status = __float_as_int(z.y);  // {pad0, prim, status, pad1[0..]}: floats 44..47 = q[11]
}

// ---- IntersectShadowTr ------------------------------------------------------------------------
// state: 0 = arrives (T_ray = 1), 1 = blocked by an opaque surface, 2 = a host-only primitive or an
// interaction the device cannot finish lies on the way: the caller's to finish
__global__ __launch_bounds__(kW2Block) void str_init(nnbvh_ray_soa q, WavefrontCount cnt, float4 *rays,
                                                     int32_t *orig, float4 *pLight, uint8_t *state) {
    const int n = w2_count(cnt);
    for (int i = blockIdx.x * kW2Block + threadIdx.x; i < n; i += gridDim.x * kW2Block) {
        float4 a, b;
        a.x = q.ox[i];
        a.y = q.oy[i];
        a.z = q.oz[i];
        a.w = q.tmax ? q.tmax[i] : __builtin_inff();
        b.x = q.dx[i];
        b.y = q.dy[i];
        b.z = q.dz[i];
        b.w = q.time ? q.time[i] : 0.0f;
        rays[2 * (long)i] = a;
        rays[2 * (long)i + 1] = b;
        orig[i] = i;
        // Point3f pLight = ray(tMax) = o + d * t (ray.h:33, intersect.h:177)
        pLight[i] = make_float4(a.x + b.x * a.w, a.y + b.y * a.w, a.z + b.z * a.w, 0.0f);
        state[i] = 0;
    }
}

__global__ __launch_bounds__(kW2Block) void str_classify(const float4 *raysCur, const float4 *hitsCur,
                                                         const int32_t *origCur, const int32_t *nCur,
                                                         const uint8_t *primClass, long nPrimClass, uint8_t *state,
                                                         float4 *raysNext, float4 *hitsNext, int32_t *origNext,
                                                         int32_t *counter, int maxItems) {
    int n = *nCur;
    n = n < 0 ? 0 : (n < maxItems ? n : maxItems);
    const int nPad = (n + 63) & ~63;  // whole wavefronts take part in the ballots
    for (int j = blockIdx.x * kW2Block + threadIdx.x; j < nPad; j += gridDim.x * kW2Block) {
        bool goOn = false;
        float4 r0, r1, h0, h1;
        int item = 0;
        if (j < n) {
            r0 = raysCur[2 * (long)j];
            r1 = raysCur[2 * (long)j + 1];
            h0 = hitsCur[2 * (long)j];
            h1 = hitsCur[2 * (long)j + 1];
            item = origCur[j];
            // while (ray.d != Vector3f(0, 0, 0)), intersect.h:183: a zero direction ends the walk, T stays 1
            const bool zeroDir = r1.x == 0.0f && r1.y == 0.0f && r1.z == 0.0f;
            const int prim = __float_as_int(h0.x);
            if (zeroDir) {
            } else if (__float_as_int(h1.w) == -1) {
                state[item] = 2;
            } else if (prim >= 0) {
                unsigned cls = NNBVH_CLASS_BASIC;
                if (primClass && (long)prim < nPrimClass) cls = primClass[prim];
                if (cls & NNBVH_CLASS_INTERFACE) goOn = true;  // result.hit && !result.material
                else state[item] = 1;                          // :190-195 hit opaque surface
            }
        }
        const int at = w2_append(goOn, counter);
        if (at >= 0) {
            raysNext[2 * (long)at] = r0;
            raysNext[2 * (long)at + 1] = r1;
            hitsNext[2 * (long)at] = h0;
            hitsNext[2 * (long)at + 1] = h1;
            origNext[at] = item;
        }
    }
}

__global__ __launch_bounds__(kW2Block) void str_spawn(const float4 *raysIn, const float4 *intr, const int32_t *origIn,
                                                      const int32_t *nIn, const float4 *pLight, uint8_t *state,
                                                      float4 *raysOut, int32_t *origOut, int32_t *counter,
                                                      int maxItems) {
    int n = *nIn;
    n = n < 0 ? 0 : (n < maxItems ? n : maxItems);
    const int nPad = (n + 63) & ~63;
    for (int j = blockIdx.x * kW2Block + threadIdx.x; j < nPad; j += gridDim.x * kW2Block) {
        bool goOn = false;
        float4 a, b;
        int item = 0;
        if (j < n) {
            item = origIn[j];
            V3 lo, hi, nn;
            int status;
            w2_read_pi_n(intr + 12 * (long)j, lo, hi, nn, status);
            if (status != NNBVH_INTERACTION_TRIANGLE && status != NNBVH_INTERACTION_PATCH) {
                state[item] = 2;
            } else {
                const float4 pl = pLight[item];
                V3 o, d;
                spawn_ray_to(lo, hi, nn, {pl.x, pl.y, pl.z}, o, d);  // ray = spawnTo(pLight), intersect.h:255
                const float4 r0 = raysIn[2 * (long)j], r1 = raysIn[2 * (long)j + 1];
                a = make_float4(o.x, o.y, o.z, r0.w);  // tMax is the work item's, unchanged (:175, :187)
                b = make_float4(d.x, d.y, d.z, r1.w);
                goOn = true;  // a zero direction is caught by the next classify pass
            }
        }
        const int at = w2_append(goOn, counter);
        if (at >= 0) {
            raysOut[2 * (long)at] = a;
            raysOut[2 * (long)at + 1] = b;
            origOut[at] = item;
        }
    }
}

// intersect.h:258-273 with T_ray = r_u = r_l = SampledSpectrum(1.f)
__global__ __launch_bounds__(kW2Block) void str_record(const uint8_t *state, WavefrontCount cnt, const float4 *Ld,
                                                       const float4 *ru, const float4 *rl,
                                                       const int32_t *pixelIndex, float *L, long nPixels,
                                                       uint8_t *visibleOut) {
    const int n = w2_count(cnt);
    for (int i = blockIdx.x * kW2Block + threadIdx.x; i < n; i += gridDim.x * kW2Block) {
        const uint8_t st = state[i];
        if (visibleOut) visibleOut[i] = st;
        if (st != 0) continue;
        const float4 ld = Ld[i], u = ru[i], l = rl[i];
        // (sr.r_u * r_u + sr.r_l * r_l).Average(), r_u = r_l = 1
        const float s0 = u.x * 1.0f + l.x * 1.0f, s1 = u.y * 1.0f + l.y * 1.0f, s2 = u.z * 1.0f + l.z * 1.0f,
                    s3 = u.w * 1.0f + l.w * 1.0f;
        float sum = s0;
        sum += s1;
        sum += s2;
        sum += s3;
        const float avg = sum / 4.0f;
        const float k = 1.0f / avg;  // T_ray / Average(): SampledSpectrum / Float, component by component
        const long px = pixelIndex[i];
        if (px < 0 || px >= nPixels) continue;
        float4 *dst = reinterpret_cast<float4 *>(L) + px;
        float4 v = *dst;
        v.x = v.x + ld.x * k;  // Ld *= T_ray / ...; L = Lpixel + Ld
        v.y = v.y + ld.y * k;
        v.z = v.z + ld.z * k;
        v.w = v.w + ld.w * k;
        *dst = v;
    }
}

// ---- IntersectOneRandom -----------------------------------------------------------------------
__global__ __launch_bounds__(kW2Block) void or_init(const float *p0, const float *p1, WavefrontCount cnt,
                                                    OneRandomState st, float4 *raysOut, int32_t *origOut,
                                                    int32_t *counter, float4 *selHits, float4 *selRays) {
    const int n = w2_count(cnt);
    const int nPad = (n + 63) & ~63;
    for (int i = blockIdx.x * kW2Block + threadIdx.x; i < nPad; i += gridDim.x * kW2Block) {
        bool goOn = false;
        float4 a, b;
        if (i < n) {
            const V3 a0 = {p0[3 * (long)i], p0[3 * (long)i + 1], p0[3 * (long)i + 2]};
            const V3 a1 = {p1[3 * (long)i], p1[3 * (long)i + 1], p1[3 * (long)i + 2]};
            Pcg32 g;  // WeightedReservoirSampler wrs(Hash(w.p0, w.p1)), aggregate.cpp:94-96
            pcg32_set_sequence(g, hash_6f(a0, a1));
            st.rng[2 * (long)i] = g.state;
            st.rng[2 * (long)i + 1] = g.inc;
            st.weights[2 * (long)i] = 0.0f;
            st.weights[2 * (long)i + 1] = 0.0f;
            // Interaction base(w.p0, 0, Medium()): exact point, n = (0, 0, 0)  (:97)
            float *pi = st.pi + 9 * (long)i;
            pi[0] = a0.x, pi[1] = a0.y, pi[2] = a0.z, pi[3] = a0.x, pi[4] = a0.y, pi[5] = a0.z;
            pi[6] = pi[7] = pi[8] = 0.0f;
            selHits[2 * (long)i] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
            selHits[2 * (long)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            selRays[2 * (long)i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            selRays[2 * (long)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            V3 o, d;
            spawn_ray_to(a0, a0, {0.0f, 0.0f, 0.0f}, a1, o, d);  // r = base.SpawnRayTo(w.p1), :99
            if (!(d.x == 0.0f && d.y == 0.0f && d.z == 0.0f)) {  // :100-101
                goOn = true;
                a = make_float4(o.x, o.y, o.z, 1.0f);  // aggregate.Intersect(r, 1), :102
                b = make_float4(d.x, d.y, d.z, 0.0f);
            }
        }
        const int at = w2_append(goOn, counter);
        if (at >= 0) {
            raysOut[2 * (long)at] = a;
            raysOut[2 * (long)at + 1] = b;
            origOut[at] = i;
        }
    }
}

__global__ __launch_bounds__(kW2Block) void or_step(const float4 *raysCur, const float4 *hitsCur, const float4 *intrCur,
                                                    const int32_t *origCur, const int32_t *nCur, const float *p1,
                                                    const int32_t *material, const int32_t *primMaterial,
                                                    long nPrimMaterial, OneRandomState st, float4 *raysNext,
                                                    int32_t *origNext, int32_t *counter, float4 *selHits,
                                                    float4 *selRays, int maxItems) {
    int n = *nCur;
    n = n < 0 ? 0 : (n < maxItems ? n : maxItems);
    const int nPad = (n + 63) & ~63;
    for (int j = blockIdx.x * kW2Block + threadIdx.x; j < nPad; j += gridDim.x * kW2Block) {
        bool goOn = false;
        float4 a, b;
        int item = 0;
        if (j < n) {
            item = origCur[j];
            const float4 h0 = hitsCur[2 * (long)j], h1 = hitsCur[2 * (long)j + 1];
            const int prim = __float_as_int(h0.x);
            V3 lo, hi, nn;
            int status;
            w2_read_pi_n(intrCur + 12 * (long)j, lo, hi, nn, status);
            if (prim < 0 && __float_as_int(h1.w) != -1) {
                // :103-104 no further surface on the segment: the walk ends
            } else if (__float_as_int(h1.w) == -1 ||
                       (status != NNBVH_INTERACTION_TRIANGLE && status != NNBVH_INTERACTION_PATCH)) {
                // a host-only primitive / an interaction the device cannot finish: the item is the caller's
                selHits[2 * (long)item + 1].w = __int_as_float(-1);
            } else {
                float *pi = st.pi + 9 * (long)item;  // base = si->intr, :105
                pi[0] = lo.x, pi[1] = lo.y, pi[2] = lo.z, pi[3] = hi.x, pi[4] = hi.y, pi[5] = hi.z;
                pi[6] = nn.x, pi[7] = nn.y, pi[8] = nn.z;
                const int mat = (primMaterial && (long)prim < nPrimMaterial) ? primMaterial[prim] : 0;
                if (mat == material[item]) {  // :106-107 wrs.Add(SubsurfaceInteraction(si->intr), 1.f)
                    Pcg32 g = {st.rng[2 * (long)item], st.rng[2 * (long)item + 1]};
                    float weightSum = st.weights[2 * (long)item];
                    const float weight = 1.0f;
                    weightSum += weight;
                    const float p = weight / weightSum;
                    if (pcg32_float(g) < p) {  // util/sampling.h:535-546
                        selHits[2 * (long)item] = h0;
                        selHits[2 * (long)item + 1] = h1;
                        selRays[2 * (long)item] = raysCur[2 * (long)j];
                        selRays[2 * (long)item + 1] = raysCur[2 * (long)j + 1];
                        st.weights[2 * (long)item + 1] = weight;
                    }
                    st.weights[2 * (long)item] = weightSum;
                    st.rng[2 * (long)item] = g.state;
                }
                const V3 a1 = {p1[3 * (long)item], p1[3 * (long)item + 1], p1[3 * (long)item + 2]};
                V3 o, d;
                spawn_ray_to(lo, hi, nn, a1, o, d);
                if (!(d.x == 0.0f && d.y == 0.0f && d.z == 0.0f)) {
                    goOn = true;
                    a = make_float4(o.x, o.y, o.z, 1.0f);
                    b = make_float4(d.x, d.y, d.z, 0.0f);
                }
            }
        }
        const int at = w2_append(goOn, counter);
        if (at >= 0) {
            raysNext[2 * (long)at] = a;
            raysNext[2 * (long)at + 1] = b;
            origNext[at] = item;
        }
    }
}

__global__ __launch_bounds__(kW2Block) void or_finish(WavefrontCount cnt, OneRandomState st, float *pdf,
                                                      float *weightSumOut) {
    const int n = w2_count(cnt);
    for (int i = blockIdx.x * kW2Block + threadIdx.x; i < n; i += gridDim.x * kW2Block) {
        const float weightSum = st.weights[2 * (long)i], reservoirWeight = st.weights[2 * (long)i + 1];
        // :110-114: HasSample() ? SampleProbability() = reservoirWeight / weightSum : 0
        pdf[i] = weightSum > 0 ? reservoirWeight / weightSum : 0.0f;
        if (weightSumOut) weightSumOut[i] = weightSum;
    }
}

// ---- launchers ----------------------------------------------------------------------------------
hipError_t launch_str_init(const nnbvh_ray_soa &q, WavefrontCount cnt, void *rays, int32_t *orig, float4 *pLight,
                           uint8_t *state, int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(str_init, dim3(w2_grid(cnt.n, maxBlocks)), dim3(kW2Block), 0, stream, q, cnt, (float4 *)rays,
                       orig, pLight, state);
    return hipGetLastError();
}
hipError_t launch_str_classify(const void *raysCur, const void *hitsCur, const int32_t *origCur,
                               const int32_t *nCur, const uint8_t *primClass, long nPrimClass, uint8_t *state,
                               void *raysNext, void *hitsNext, int32_t *origNext, int32_t *counter, int maxItems,
                               int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(str_classify, dim3(w2_grid(maxItems, maxBlocks)), dim3(kW2Block), 0, stream,
                       (const float4 *)raysCur, (const float4 *)hitsCur, origCur, nCur, primClass, nPrimClass, state,
                       (float4 *)raysNext, (float4 *)hitsNext, origNext, counter, maxItems);
    return hipGetLastError();
}
hipError_t launch_str_spawn(const void *raysIn, const void *intr, const int32_t *origIn, const int32_t *nIn,
                            const float4 *pLight, uint8_t *state, void *raysOut, int32_t *origOut,
                            int32_t *counter, int maxItems, int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(str_spawn, dim3(w2_grid(maxItems, maxBlocks)), dim3(kW2Block), 0, stream,
                       (const float4 *)raysIn, (const float4 *)intr, origIn, nIn, pLight, state, (float4 *)raysOut,
                       origOut, counter, maxItems);
    return hipGetLastError();
}
hipError_t launch_str_record(const uint8_t *state, WavefrontCount cnt, const float *Ld, const float *ru,
                             const float *rl, const int32_t *pixelIndex, float *L, long nPixels,
                             uint8_t *visibleOut, int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(str_record, dim3(w2_grid(cnt.n, maxBlocks)), dim3(kW2Block), 0, stream, state, cnt,
                       (const float4 *)Ld, (const float4 *)ru, (const float4 *)rl, pixelIndex, L, nPixels,
                       visibleOut);
    return hipGetLastError();
}
hipError_t launch_or_init(const float *p0, const float *p1, WavefrontCount cnt, OneRandomState st, void *raysOut,
                          int32_t *origOut, int32_t *counter, void *selHits, void *selRays, int maxBlocks,
                          hipStream_t stream) {
    hipLaunchKernelGGL(or_init, dim3(w2_grid(cnt.n, maxBlocks)), dim3(kW2Block), 0, stream, p0, p1, cnt, st,
                       (float4 *)raysOut, origOut, counter, (float4 *)selHits, (float4 *)selRays);
    return hipGetLastError();
}
hipError_t launch_or_step(const void *raysCur, const void *hitsCur, const void *intrCur, const int32_t *origCur,
                          const int32_t *nCur, const float *p1, const int32_t *material,
                          const int32_t *primMaterial, long nPrimMaterial, OneRandomState st, void *raysNext,
                          int32_t *origNext, int32_t *counter, void *selHits, void *selRays, int maxItems,
                          int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(or_step, dim3(w2_grid(maxItems, maxBlocks)), dim3(kW2Block), 0, stream,
                       (const float4 *)raysCur, (const float4 *)hitsCur, (const float4 *)intrCur, origCur, nCur, p1,
                       material, primMaterial, nPrimMaterial, st, (float4 *)raysNext, origNext, counter,
                       (float4 *)selHits, (float4 *)selRays, maxItems);
    return hipGetLastError();
}
hipError_t launch_or_finish(WavefrontCount cnt, OneRandomState st, float *pdf, float *weightSum, int maxBlocks,
                            hipStream_t stream) {
    hipLaunchKernelGGL(or_finish, dim3(w2_grid(cnt.n, maxBlocks)), dim3(kW2Block), 0, stream, cnt, st, pdf,
                       weightSum);
    return hipGetLastError();
}

}  // namespace nnbvh
