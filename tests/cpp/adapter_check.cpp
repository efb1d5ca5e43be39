// Exercises include/nnbvh_aggregate.hpp the way a pbrt integrator would: per-ray Intersect /
// IntersectP (Primitive interface) and the batched WavefrontAggregate-shaped calls, and checks
// that the two agree with each other.  Built by tests/test_cpp_adapter.py with g++ against
// libnnbvh_hip.so; run only where a GPU is present.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "nnbvh_aggregate.hpp"

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    const int nTris = 500;
    std::vector<float> verts;
    std::vector<nnbvh_prim> prims;
    for (int i = 0; i < nTris; ++i) {
        float c[3] = {5 * U(rng), 5 * U(rng), 5 * U(rng)};
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) verts.push_back(c[a] + 0.5f * U(rng));
        prims.push_back(nnbvh_prim{NNBVH_PRIM_TRIANGLE, i, {3 * i, 3 * i + 1, 3 * i + 2, 0}});
    }
    nnbvh::HipBVHAggregate agg(prims, verts);  // sah, maxnodeprims 4, like BVHAggregate::Create
    nnbvh::Bounds3f b = agg.Bounds();
    if (!(b.pMin.x < b.pMax.x)) return 10;

    const int nRays = 200;
    std::vector<nnbvh_ray> rays(nRays);
    for (auto &r : rays) {
        float o[3] = {6 * U(rng), 6 * U(rng), 6 * U(rng)}, t[3] = {3 * U(rng), 3 * U(rng), 3 * U(rng)};
        for (int a = 0; a < 3; ++a) {
            r.o[a] = o[a];
            r.d[a] = t[a] - o[a];
        }
        r.tmax = INFINITY;
        r.time = 0;
    }
    std::vector<nnbvh_hit> hits(nRays);
    std::vector<uint8_t> occ(nRays);
    agg.IntersectClosest(rays.data(), nRays, hits.data());
    agg.IntersectShadow(rays.data(), nRays, occ.data());
    int nHit = 0;
    for (int i = 0; i < nRays; ++i) {
        nnbvh::Ray ray{{rays[i].o[0], rays[i].o[1], rays[i].o[2]},
                       {rays[i].d[0], rays[i].d[1], rays[i].d[2]}, 0};
        auto si = agg.Intersect(ray);
        bool p = agg.IntersectP(ray);
        if (si.has_value() != (hits[i].prim >= 0)) return 1;
        if (si && (si->prim != hits[i].prim || std::memcmp(&si->tHit, &hits[i].t, 4))) return 2;
        if (p != (occ[i] != 0) || p != si.has_value()) return 3;
        nHit += si.has_value();
    }
    // hit records -> SurfaceInteraction (Triangle::InteractionFromIntersection)
    std::vector<int32_t> triVerts;
    for (int i = 0; i < nTris; ++i)
        for (int k = 0; k < 3; ++k) triVerts.push_back(3 * i + k);
    nnbvh::HipShadingMesh mesh(verts.data(), (int)verts.size() / 3, triVerts.data(), nTris);
    std::vector<nnbvh_interaction> intr(nRays);
    mesh.Interactions(rays.data(), hits.data(), nRays, intr.data());
    for (int i = 0; i < nRays; ++i) {
        if (intr[i].prim != hits[i].prim) return 5;
        if ((hits[i].prim >= 0) != (intr[i].status == NNBVH_INTERACTION_TRIANGLE)) return 6;
        if (hits[i].prim < 0) continue;
        // the hit point lies on the ray at tHit, inside pi, and n is a unit vector
        for (int a = 0; a < 3; ++a) {
            float p = rays[i].o[a] + hits[i].t * rays[i].d[a];
            float mid = 0.5f * (intr[i].pi_lo[a] + intr[i].pi_hi[a]);
            if (std::fabs(p - mid) > 1e-3f * (1 + std::fabs(p))) return 7;
        }
        float n2 = intr[i].n[0] * intr[i].n[0] + intr[i].n[1] * intr[i].n[1] + intr[i].n[2] * intr[i].n[2];
        if (std::fabs(n2 - 1) > 1e-5f) return 8;
    }
    // the other accelerator through the same Primitive-shaped interface: same hits, same t
    nnbvh::HipKdTreeAggregate kd(prims, verts);  // KdTreeAggregate::Create's defaults
    std::vector<nnbvh_hit> kdHits(nRays);
    kd.IntersectClosest(rays.data(), nRays, kdHits.data());
    for (int i = 0; i < nRays; ++i) {
        if ((kdHits[i].prim >= 0) != (hits[i].prim >= 0)) return 11;
        if (hits[i].prim >= 0 && std::memcmp(&kdHits[i].t, &hits[i].t, 4)) return 12;
        nnbvh::Ray ray{{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, {rays[i].d[0], rays[i].d[1], rays[i].d[2]}, 0};
        if (kd.IntersectP(ray) != (occ[i] != 0)) return 13;
    }
    if (!(kd.Bounds().pMin.x <= b.pMin.x + 1e-6f)) return 14;
    // a scene with a host-only primitive in the way: the single-ray adapters must SAY that the ray is void
    // (ADVICE r2), in both accelerators — never answer "occluded" or "miss" for it
    {
        std::vector<float> v2 = {-1, -1, 0, 1, -1, 0, 0, 1, 0,    // z = 0: a triangle only the host can intersect
                                 -1, -1, 2, 1, -1, 2, 0, 1, 2};   // z = 2: an ordinary triangle behind it
        std::vector<nnbvh_prim> p2 = {nnbvh_prim{NNBVH_PRIM_HOST, 0, {0, 1, 2, 0}},
                                      nnbvh_prim{NNBVH_PRIM_TRIANGLE, 1, {3, 4, 5, 0}}};
        std::vector<float> pb = {-1, -1, 0, 1, 1, 0, /* read for the host primitive only */ 0, 0, 0, 0, 0, 0};
        nnbvh::HipBVHAggregate a2(p2, v2, 4, "sah", 0, &pb);
        nnbvh::HipKdTreeAggregate k2(p2, v2, 5, 1, 0.5f, 1, -1, 0, &pb);
        // `through` crosses the host primitive and then the triangle; `grazing` crosses the host primitive's
        // bounds only (an any-hit ray that a device primitive occludes is occluded whatever the host one says)
        nnbvh::Ray through{{0, -0.5f, -1}, {0, 0, 1}, 0}, grazing{{0.9f, 0.9f, -1}, {0, 0, 1}, 0},
            beside{{5, 5, -1}, {0, 0, 1}, 0};
        bool nh = false;
        if (a2.Intersect(through, INFINITY, &nh).has_value() || !nh) return 20;
        if (a2.IntersectP(grazing, INFINITY, &nh) || !nh) return 21;
        if (!a2.IntersectP(through, INFINITY, &nh) || nh) return 27;
        if (k2.Intersect(through, INFINITY, &nh).has_value() || !nh) return 22;
        if (k2.IntersectP(grazing, INFINITY, &nh) || !nh) return 23;
        nh = true;
        if (a2.Intersect(beside, INFINITY, &nh).has_value() || nh) return 24;
        nh = true;
        if (k2.IntersectP(beside, INFINITY, &nh) || nh) return 25;
        // without the out-parameter a void ray is fatal
        static int fatals = 0;
        auto prev = nnbvh::HipBVHAggregate::fatal_handler();
        nnbvh::HipBVHAggregate::fatal_handler() = [](const char *) { ++fatals; };
        (void)a2.IntersectP(grazing);
        (void)k2.Intersect(through);
        nnbvh::HipBVHAggregate::fatal_handler() = prev;
        if (fatals != 2) return 26;
    }
    // per-ray calls from many threads (the CPU integrators' shape), coalesced into batches: every thread gets
    // exactly the batched results, and far fewer batches than rays were launched
    {
        nnbvh::CoalescingAggregate<nnbvh::HipBVHAggregate> co(agg, 64, std::chrono::microseconds(200));
        const int nThreads = 8;
        std::vector<int> bad(nThreads, 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < nThreads; ++t)
            pool.emplace_back([&, t] {
                for (int rep = 0; rep < 4; ++rep)
                    for (int i = t; i < nRays; i += nThreads) {
                        nnbvh::Ray ray{{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, {rays[i].d[0], rays[i].d[1], rays[i].d[2]}, 0};
                        auto si = co.Intersect(ray);
                        if (si.has_value() != (hits[i].prim >= 0)) ++bad[t];
                        else if (si && (si->prim != hits[i].prim || std::memcmp(&si->tHit, &hits[i].t, 4))) ++bad[t];
                        if (co.IntersectP(ray) != (occ[i] != 0)) ++bad[t];
                    }
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < nThreads; ++t)
            if (bad[t]) return 30;
        uint64_t batches = 0, traced = 0;
        co.Stats(&batches, &traced);
        if (traced != 2ull * 4 * nRays) return 31;
        if (batches * 2 > traced) return 32;  // on average more than two rays per launch
        std::printf("coalesced: %llu rays in %llu batches\n", (unsigned long long)traced, (unsigned long long)batches);
    }
    // an alpha-tested bilinear patch (per-primitive alpha array, reordered with the primitives by the adapter):
    // opaque it stops the ray, fully transparent it lets the ray through to the triangle behind it
    {
        std::vector<float> v = {-1, -1, 0, 1, -1, 0, -1, 1, 0, 1, 1, 0.3f,  // a (slightly twisted) patch in z ~ 0
                                -1, -1, 2, 1, -1, 2, 0, 1, 2};             // a triangle behind it, z = 2
        std::vector<nnbvh_prim> pr = {nnbvh_prim{NNBVH_PRIM_TRIANGLE, 70, {4, 5, 6, 0}},
                                      nnbvh_prim{NNBVH_PRIM_ALPHA_PATCH, 71, {0, 1, 2, 3}}};
        const nnbvh::Ray down{{0.1f, -0.2f, -3}, {0, 0, 1}, 0};
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<float> alpha = {0.0f, pass == 0 ? 1.0f : 0.0f};
            nnbvh::HipBVHAggregate a(pr, v, 1, "sah", 0, nullptr, nullptr, &alpha);
            auto si = a.Intersect(down);
            if (!si || si->prim != (pass == 0 ? 71 : 70)) return 40 + pass;
            if (!a.IntersectP(down)) return 42;
        }
    }
    // the film: an empty film reads back as zeros (accumulation itself: tests/test_film.py)
    nnbvh::HipFilm film(0, 0, 8, 4);
    std::vector<double> px = film.Read();
    if (px.size() != 8 * 4 * 4) return 15;
    for (double v : px)
        if (v != 0.0) return 16;
    std::printf("adapter ok: %d rays, %d hits\n", nRays, nHit);
    return nHit > 0 ? 0 : 4;
}
