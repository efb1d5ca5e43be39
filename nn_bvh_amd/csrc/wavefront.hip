// wavefront.hip — the queue side of the batched caller (WavefrontAggregate,
// /root/reference/src/pbrt/wavefront/integrator.h:32-54): gather a SOA ray queue into the trace
// kernel's 32-B ray records, and after the trace apply the reference's enqueue rules
// (wavefront/intersect.h:16-30, 49-156) and shadow-ray bookkeeping (intersect.h:32-47) on the
// device, so a wavefront stage never round-trips through the host.
//
// All three kernels are streaming passes over n work items (64-100 B each): HBM-bound, a few
// percent of the traversal they bracket.
#include <hip/hip_runtime.h>

#include "wavefront.h"

namespace nnbvh {

static constexpr int kWfBlock = 256;

__device__ __forceinline__ int wf_count(const WavefrontCount &c) {
    int n = c.n;
    if (c.nDev) {
        const int nd = *c.nDev;
        n = nd < 0 ? 0 : (nd < n ? nd : n);
    }
    return n;
}

// SOA<Ray> (o.x o.y o.z d.x d.y d.z time; wavefront/workitems.soa:40-50) -> nnbvh_ray records.
__global__ __launch_bounds__(kWfBlock) void wf_gather_rays(nnbvh_ray_soa q, WavefrontCount cnt,
                                                           float4 *__restrict__ rays) {
    const int n = wf_count(cnt);
    for (int i = blockIdx.x * kWfBlock + threadIdx.x; i < n; i += gridDim.x * kWfBlock) {
        float4 a, b;
        a.x = q.ox[i];
        a.y = q.oy[i];
        a.z = q.oz[i];
        a.w = q.tmax ? q.tmax[i] : __builtin_inff();  // RayQueue rays: tMax = Infinity
        b.x = q.dx[i];
        b.y = q.dy[i];
        b.z = q.dz[i];
        b.w = q.time ? q.time[i] : 0.0f;
        rays[2 * (long)i] = a;
        rays[2 * (long)i + 1] = b;
    }
}

// ---- EnqueueWorkAfterMiss / EnqueueWorkAfterIntersection (wavefront/intersect.h:16-30, 49-156)
// on work-item indices.  prim_class[prim id] carries what the reference reads off the hit's
// SurfaceInteraction: material present?, evaluable with the basic texture evaluator?, area light?
//
// WorkQueue::Push (wavefront/workqueue.h:78-99) is one atomic per item in the reference.  Here a
// block classifies a chunk of kWfChunk items first, reserves each destination queue's range with
// ONE atomicAdd per queue and chunk, and then stores the indices: device-scope atomics on one
// address are served at the memory side of the 8 XCD-private L2s (~175 K per queue and launch
// serialised when issued per wavefront: 4 ms; per chunk they vanish).
static constexpr int kWfItems = 16;                       // items per thread and chunk
static constexpr int kWfChunk = kWfBlock * kWfItems;      // 4096 items per block iteration
static constexpr int kWfQueues = 6;
enum : unsigned { kQEscaped = 1, kQAreaLight = 2, kQBasic = 4, kQUniversal = 8, kQMedium = 16, kQNext = 32 };

__device__ __forceinline__ unsigned wf_destinations(int prim, bool medium, unsigned cls) {
    // intersect.h:19-29 (miss) and :56-90 (hit): a ray inside a medium goes to the medium sampler
    // whatever it hit
    if (medium) return kQMedium;
    if (prim < 0) return kQEscaped;
    if (cls & NNBVH_CLASS_INTERFACE) return kQNext;  // :103-111 no material: the ray continues
    // :113-120 emissive surface; :122-128 which material-evaluation queue
    return ((cls & NNBVH_CLASS_AREA_LIGHT) ? kQAreaLight : 0u) |
           ((cls & NNBVH_CLASS_UNIVERSAL) ? kQUniversal : kQBasic);
}

__global__ __launch_bounds__(kWfBlock) void wf_enqueue_closest(
    const float4 *__restrict__ hits, WavefrontCount cnt, const uint8_t *__restrict__ hasMedium,
    const uint8_t *__restrict__ primClass, long nPrimClass, nnbvh_closest_queues out) {
    __shared__ int waveCount[kWfBlock / 64][kWfQueues];
    __shared__ int waveBase[kWfBlock / 64][kWfQueues];
    const nnbvh_work_queue *queues = &out.escaped;  // six consecutive members, in bit order
    const int n = wf_count(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long chunk = (long)blockIdx.x * kWfChunk; chunk < n; chunk += (long)gridDim.x * kWfChunk) {
        unsigned dest[kWfItems];
        int count[kWfQueues] = {0, 0, 0, 0, 0, 0};  // wave-uniform
#pragma unroll
        for (int k = 0; k < kWfItems; ++k) {
            const long i = chunk + k * kWfBlock + threadIdx.x;
            unsigned d = 0;
            if (i < n) {
                const int prim = __float_as_int(hits[2 * i].x);
                const bool medium = hasMedium && hasMedium[i] != 0;
                unsigned cls = NNBVH_CLASS_BASIC;
                if (prim >= 0 && primClass && (long)prim < nPrimClass) cls = primClass[prim];
                d = wf_destinations(prim, medium, cls);
            }
            dest[k] = d;
#pragma unroll
            for (int q = 0; q < kWfQueues; ++q) count[q] += __popcll(__ballot((d >> q) & 1u));
        }
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < kWfQueues; ++q) waveCount[wave][q] = count[q];
        __syncthreads();
        if (threadIdx.x < kWfQueues) {
            const int q = threadIdx.x;
            int total = 0;
            for (int w = 0; w < kWfBlock / 64; ++w) total += waveCount[w][q];
            int base = 0;
            if (total > 0 && queues[q].size) base = atomicAdd(queues[q].size, total);
            for (int w = 0; w < kWfBlock / 64; ++w) {
                waveBase[w][q] = base;
                base += waveCount[w][q];
            }
        }
        __syncthreads();
        int run[kWfQueues];
#pragma unroll
        for (int q = 0; q < kWfQueues; ++q) run[q] = waveBase[wave][q];
#pragma unroll
        for (int k = 0; k < kWfItems; ++k) {
            const int i = (int)(chunk + k * kWfBlock + threadIdx.x);
#pragma unroll
            for (int q = 0; q < kWfQueues; ++q) {
                const bool push = (dest[k] >> q) & 1u;
                const unsigned long long mask = __ballot(push);
                if (push && queues[q].size) {
                    const int at = run[q] + (int)__builtin_amdgcn_mbcnt_hi(
                                                (unsigned)(mask >> 32),
                                                __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (at < queues[q].capacity) queues[q].items[at] = i;  // beyond: counted only
                }
                run[q] += __popcll(mask);
            }
        }
        __syncthreads();  // the next chunk reuses the LDS counters
    }
}

// RecordShadowRayResult (wavefront/intersect.h:32-47): an unoccluded ray adds
// Ld / (r_u + r_l).Average() to its pixel sample's L (SampledSpectrum = 4 floats,
// util/spectrum.h:36; Average() sums left to right then divides by 4, spectrum.h:256-261).
__global__ __launch_bounds__(kWfBlock) void wf_record_shadow(
    const uint8_t *__restrict__ occluded, WavefrontCount cnt, const float4 *__restrict__ Ld,
    const float4 *__restrict__ ru, const float4 *__restrict__ rl,
    const int32_t *__restrict__ pixelIndex, float *L, long nPixels) {
    const int n = wf_count(cnt);
    for (int i = blockIdx.x * kWfBlock + threadIdx.x; i < n; i += gridDim.x * kWfBlock) {
        if (occluded[i] != 0) continue;  // 1 = occluded; 2 = "needs the host": the caller's to finish
        const float4 ld = Ld[i], u = ru[i], l = rl[i];
        const float s0 = u.x + l.x, s1 = u.y + l.y, s2 = u.z + l.z, s3 = u.w + l.w;
        float sum = s0;
        sum += s1;
        sum += s2;
        sum += s3;
        const float avg = sum / 4.0f;
        const long px = pixelIndex[i];
        if (px < 0 || px >= nPixels) continue;
        // plain read-modify-write, as the reference does: pixelIndex is unique within a stage
        // (one shadow ray per pixel sample), SOA<SampledSpectrum> keeps L as one float4 per pixel
        float4 *dst = reinterpret_cast<float4 *>(L) + px;
        float4 v = *dst;
        v.x = v.x + ld.x / avg;
        v.y = v.y + ld.y / avg;
        v.z = v.z + ld.z / avg;
        v.w = v.w + ld.w / avg;
        *dst = v;
    }
}

static int wf_grid(int n, int maxBlocks) {
    int blocks = (n + kWfBlock - 1) / kWfBlock;
    if (blocks < 1) blocks = 1;
    return blocks < maxBlocks ? blocks : maxBlocks;
}

hipError_t launch_wf_gather(const nnbvh_ray_soa &q, WavefrontCount cnt, void *rays, int maxBlocks,
                            hipStream_t stream) {
    hipLaunchKernelGGL(wf_gather_rays, dim3(wf_grid(cnt.n, maxBlocks)), dim3(kWfBlock), 0, stream, q,
                       cnt, (float4 *)rays);
    return hipGetLastError();
}

hipError_t launch_wf_enqueue_closest(const void *hits, WavefrontCount cnt, const uint8_t *hasMedium,
                                     const uint8_t *primClass, long nPrimClass,
                                     const nnbvh_closest_queues &out, int maxBlocks,
                                     hipStream_t stream) {
    int blocks = (cnt.n + kWfChunk - 1) / kWfChunk;
    blocks = blocks < 1 ? 1 : (blocks < maxBlocks ? blocks : maxBlocks);
    hipLaunchKernelGGL(wf_enqueue_closest, dim3(blocks), dim3(kWfBlock), 0, stream,
                       (const float4 *)hits, cnt, hasMedium, primClass, nPrimClass, out);
    return hipGetLastError();
}

hipError_t launch_wf_record_shadow(const uint8_t *occluded, WavefrontCount cnt, const float *Ld,
                                   const float *ru, const float *rl, const int32_t *pixelIndex,
                                   float *L, long nPixels, int maxBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_record_shadow, dim3(wf_grid(cnt.n, maxBlocks)), dim3(kWfBlock), 0, stream,
                       occluded, cnt, (const float4 *)Ld, (const float4 *)ru, (const float4 *)rl,
                       pixelIndex, L, nPixels);
    return hipGetLastError();
}

}  // namespace nnbvh
