// wavefront.hip — the queue side of the batched caller (WavefrontAggregate,
// /root/reference/src/pbrt/wavefront/integrator.h:32-54): gather a SOA ray queue into the trace
// kernel's 32-B ray records, and after the trace apply the reference's enqueue rules
// (wavefront/intersect.h:16-30, 49-156) and shadow-ray bookkeeping (intersect.h:32-47) on the
// device, so a wavefront stage never round-trips through the host.
//
// All three kernels are streaming passes over n work items (64-100 B each): HBM-bound, a few
// percent of the traversal they bracket.  Queue pushes are wave-aggregated: one atomicAdd per
// wavefront and destination queue, lanes take consecutive slots by their rank in the ballot.
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

// WorkQueue::Push (wavefront/workqueue.h:78-99) for a whole wavefront at once.  Items beyond the
// queue's capacity are counted but not stored (the reference DCHECKs instead).
__device__ __forceinline__ void wf_push(const nnbvh_work_queue &q, bool push, int item) {
    const unsigned long long mask = __ballot(push);
    if (mask == 0ull || q.size == nullptr) return;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(q.size, __popcll(mask));
    base = __shfl(base, leader);
    if (push) {
        const int at = base + lane;
        if (at < q.capacity) q.items[at] = item;
    }
}

// EnqueueWorkAfterMiss / EnqueueWorkAfterIntersection (wavefront/intersect.h:16-30, 49-156) on
// work-item indices.  prim_class[prim id] carries what the reference reads off the hit's
// SurfaceInteraction: material present?, evaluable with the basic texture evaluator?, area light?
__global__ __launch_bounds__(kWfBlock) void wf_enqueue_closest(
    const float4 *__restrict__ hits, WavefrontCount cnt, const uint8_t *__restrict__ hasMedium,
    const uint8_t *__restrict__ primClass, long nPrimClass, nnbvh_closest_queues out) {
    const int n = wf_count(cnt);
    const int nRound = (n + 63) & ~63;  // whole wavefronts take part in the ballots
    for (int i = blockIdx.x * kWfBlock + threadIdx.x; i < nRound; i += gridDim.x * kWfBlock) {
        const bool live = i < n;
        int prim = -1;
        bool medium = false;
        if (live) {
            prim = __float_as_int(hits[2 * (long)i].x);
            medium = hasMedium && hasMedium[i] != 0;
        }
        unsigned cls = NNBVH_CLASS_BASIC;
        if (live && prim >= 0 && primClass && (long)prim < nPrimClass) cls = primClass[prim];
        const bool miss = live && prim < 0;
        const bool hit = live && prim >= 0;
        // intersect.h:19-29 (miss) and :56-90 (hit): a ray inside a medium goes to the medium
        // sampler whatever it hit
        wf_push(out.medium_sample, live && medium, i);
        wf_push(out.escaped, miss && !medium, i);
        const bool surface = hit && !medium;
        // :103-111 no material = interface between media: the ray continues
        const bool iface = surface && (cls & NNBVH_CLASS_INTERFACE);
        wf_push(out.next_ray, iface, i);
        // :113-120 emissive surface
        wf_push(out.hit_area_light, surface && !iface && (cls & NNBVH_CLASS_AREA_LIGHT), i);
        // :122-128 which material-evaluation queue
        const bool universal = (cls & NNBVH_CLASS_UNIVERSAL) != 0;
        wf_push(out.basic_eval_material, surface && !iface && !universal, i);
        wf_push(out.universal_eval_material, surface && !iface && universal, i);
    }
}

// RecordShadowRayResult (wavefront/intersect.h:32-47): an unoccluded ray adds
// Ld / (r_u + r_l).Average() to its pixel sample's L (SampledSpectrum = 4 floats,
// util/spectrum.h:36; Average() sums left to right then divides by 4, spectrum.h:256-261).
__global__ __launch_bounds__(kWfBlock) void wf_record_shadow(
    const uint8_t *__restrict__ occluded, WavefrontCount cnt, const float4 *__restrict__ Ld,
    const float4 *__restrict__ ru, const float4 *__restrict__ rl,
    const int32_t *__restrict__ pixelIndex, float *__restrict__ L, long nPixels) {
    const int n = wf_count(cnt);
    for (int i = blockIdx.x * kWfBlock + threadIdx.x; i < n; i += gridDim.x * kWfBlock) {
        if (occluded[i] == 1) continue;  // 2 = "needs the host" is left to the caller as well
        if (occluded[i] != 0) continue;
        const float4 ld = Ld[i], u = ru[i], l = rl[i];
        const float s0 = u.x + l.x, s1 = u.y + l.y, s2 = u.z + l.z, s3 = u.w + l.w;
        float sum = s0;
        sum += s1;
        sum += s2;
        sum += s3;
        const float avg = sum / 4.0f;
        const long px = pixelIndex[i];
        if (px < 0 || px >= nPixels) continue;
        float *dst = L + 4 * px;
        // pixelIndex is unique within a stage in the reference (one shadow ray per pixel sample),
        // so one add per component; atomics only make a caller's duplicates complete, not racy.
        atomicAdd(dst + 0, ld.x / avg);
        atomicAdd(dst + 1, ld.y / avg);
        atomicAdd(dst + 2, ld.z / avg);
        atomicAdd(dst + 3, ld.w / avg);
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
    hipLaunchKernelGGL(wf_enqueue_closest, dim3(wf_grid(cnt.n, maxBlocks)), dim3(kWfBlock), 0, stream,
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
