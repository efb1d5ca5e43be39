// bvh_layout.cpp — memory order of the baked arrays.  A child reference is a record NUMBER (interior) or
// a slot NUMBER (leaf), so the interior records and the leaves' slot runs may be stored in any order
// without touching what a ray computes (same boxes, same tests, same counts: aggregates.cpp:529-624 see
// the nodes in the same sequence); the order only decides which records share a 128-B cache line.
//
//   records: 0 the reference's DFS order (as baked: a node's first child follows it)
//            1 sibling pairs: the two children of a node in ONE 128-B line, pairs in DFS order
//            2 the top levels breadth-first (one contiguous block), sibling pairs in DFS order below
//            9 a random permutation (calibration: what locality of the order is worth at all)
//   leaves (+16): a leaf's slots never straddle a 128-B line if they fit into one
// Host code; runs once at scene creation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "nnbvh_internal.h"

namespace nnbvh {

namespace {

struct PairAlloc {  // hands out record positions: aligned pairs, singles into the holes pairs leave
    int pos = 0;
    std::vector<int> holes;
    int single() {
        if (!holes.empty()) {
            const int h = holes.back();
            holes.pop_back();
            return h;
        }
        return pos++;
    }
    int pair() {
        if (pos & 1) holes.push_back(pos++);
        const int at = pos;
        pos += 2;
        return at;
    }
};

// positions of the subtree below record `root` (already placed): children as aligned pairs, depth first
void place_pairs_dfs(const std::vector<WideNode> &w, int root, PairAlloc &al, std::vector<int> &at) {
    std::vector<int> st{root};
    while (!st.empty()) {
        const int n = st.back();
        st.pop_back();
        const int a = w[n].ref0, b = w[n].ref1;
        if (a >= 0 && b >= 0) {
            const int p = al.pair();
            at[a] = p;
            at[b] = p + 1;
            st.push_back(b);
            st.push_back(a);
        } else if (a >= 0 || b >= 0) {
            const int c = a >= 0 ? a : b;
            at[c] = al.single();
            st.push_back(c);
        }
    }
}

// mode 32 (experiment, NNBVH_FAT builds of the lean kernel instances only): 192-B "fat" records — a node's own
// record followed by COPIES of its two children's records, so that one trip to memory serves the node and the
// child the ray enters next (the reference's binary order replayed from the copy: same boxes, same arithmetic,
// same counts).  Interior references become (record << 2) | axis of the referenced node: which copy a ray needs
// (its near child's, aggregates.cpp:562-568) is then known before the record arrives.
bool fatten_scene(float4 **d_wide, float4 **d_prims, int *n_interior, int64_t *n_slots, int *root_ref,
                  std::string *error) {
    const int n = *n_interior;
    const int64_t ns = *n_slots;
    if (n <= 0 || *root_ref < 0) return true;
    if (n >= (1 << 22)) {
        *error = "fat records: more than 2^22 interior records (192-B records behind 32-bit offsets)";
        return false;
    }
    std::vector<WideNode> w((size_t)n);
    std::vector<float4> stream((size_t)ns);
    auto fail = [&](const char *what, hipError_t e) {
        *error = std::string("fat records: ") + what + ": " + hipGetErrorString(e);
        return false;
    };
    hipError_t e = hipMemcpy(w.data(), *d_wide, (size_t)n * sizeof(WideNode), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("read records", e);
    e = hipMemcpy(stream.data(), *d_prims, (size_t)ns * 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("read stream", e);
    auto enc = [&](int ref) { return ref >= 0 ? ((ref << 2) | (w[(size_t)ref].axis & 3)) : ref; };
    std::vector<WideNode> fat((size_t)n * 3);
    std::memset(fat.data(), 0, fat.size() * sizeof(WideNode));
    for (int i = 0; i < n; ++i) {
        WideNode r = w[(size_t)i];
        r.ref0 = enc(w[(size_t)i].ref0);
        r.ref1 = enc(w[(size_t)i].ref1);
        fat[(size_t)i * 3] = r;
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 2; ++c) {
            const int ref = c ? w[(size_t)i].ref1 : w[(size_t)i].ref0;
            if (ref >= 0) fat[(size_t)i * 3 + 1 + c] = fat[(size_t)ref * 3];
        }
    const size_t wideBytes = (fat.size() * sizeof(WideNode) + 255) & ~(size_t)255;
    void *arena = nullptr;
    e = hipMalloc(&arena, wideBytes + (size_t)ns * 16 + 64);
    if (e != hipSuccess) return fail("hipMalloc", e);
    char *ps = (char *)arena + wideBytes;
    e = hipMemcpy(arena, fat.data(), fat.size() * sizeof(WideNode), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ps, stream.data(), (size_t)ns * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(ps + (size_t)ns * 16, 0, 64);
    if (e != hipSuccess) {
        (void)hipFree(arena);
        return fail("upload", e);
    }
    (void)hipFree(*d_wide);
    *d_wide = (float4 *)arena;
    *d_prims = (float4 *)ps;
    *n_interior = 3 * n;  // records of 64 B the allocation holds (sizes, the 4-GiB check)
    *root_ref = enc(*root_ref);
    return true;
}

}  // namespace

// Returns false (error set) on a device error; otherwise *d_wide / *d_prims / *n_interior / *n_slots
// describe the new allocation (records, 256-B aligned stream, 64 B padding) and the old one is freed.
bool relayout_scene(int mode, float4 **d_wide, float4 **d_prims, int *n_interior, int64_t *n_slots, int *root_ref,
                    int top_levels, std::string *error) {
    if (mode == 32) return fatten_scene(d_wide, d_prims, n_interior, n_slots, root_ref, error);
    const int recMode = mode & 15;
    const bool alignLeaves = (mode & 16) != 0;
    if (recMode == 0 && !alignLeaves) return true;
    const int n = *n_interior;
    const int64_t ns = *n_slots;
    std::vector<WideNode> w((size_t)std::max(n, 1));
    std::vector<float4> stream((size_t)ns);
    auto fail = [&](const char *what, hipError_t e) {
        *error = std::string("relayout: ") + what + ": " + hipGetErrorString(e);
        return false;
    };
    hipError_t e = hipMemcpy(w.data(), *d_wide, (size_t)n * sizeof(WideNode), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("read records", e);
    e = hipMemcpy(stream.data(), *d_prims, (size_t)ns * 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail("read stream", e);

    // ---- records
    std::vector<int> at((size_t)std::max(n, 1), -1);
    int nNew = n;
    if (n > 0 && *root_ref >= 0 && recMode != 0) {
        if (recMode == 9) {
            std::vector<int> perm((size_t)n - 1);
            std::iota(perm.begin(), perm.end(), 1);
            std::mt19937 rng(12345);
            std::shuffle(perm.begin(), perm.end(), rng);
            at[0] = 0;
            for (int i = 1; i < n; ++i) at[i] = perm[(size_t)i - 1];
        } else {
            PairAlloc al;
            at[0] = al.single();
            std::vector<int> frontier{0};
            if (recMode == 2) {  // breadth first down to top_levels, pairs kept together
                for (int lvl = 0; lvl < top_levels && !frontier.empty(); ++lvl) {
                    std::vector<int> next;
                    for (int nd : frontier) {
                        const int a = w[nd].ref0, b = w[nd].ref1;
                        if (a >= 0 && b >= 0) {
                            const int p = al.pair();
                            at[a] = p;
                            at[b] = p + 1;
                            next.push_back(a);
                            next.push_back(b);
                        } else if (a >= 0 || b >= 0) {
                            const int c = a >= 0 ? a : b;
                            at[c] = al.single();
                            next.push_back(c);
                        }
                    }
                    frontier.swap(next);
                }
            }
            for (int nd : frontier) place_pairs_dfs(w, nd, al, at);
            nNew = al.pos;
        }
    } else {
        std::iota(at.begin(), at.end(), 0);
    }

    // ---- leaves: new first slot of every leaf (slot -> slot map only at leaf starts)
    std::vector<float4> streamNew;
    std::vector<int64_t> slotAt;  // indexed by old first slot
    if (alignLeaves) {
        slotAt.assign((size_t)ns + 1, -1);
        streamNew.reserve((size_t)ns + (size_t)ns / 3);
        int64_t i = 0;
        while (i < ns) {  // leaves are the runs that end with a kPrimLast primitive
            int64_t j = i;
            for (;;) {
                const unsigned flags = (unsigned)__builtin_bit_cast(int, stream[(size_t)j + 1].w);
                const int len = (flags & (kPrimInstance | kPrimSmooth)) ? 6 : ((flags & kPrimPatch) ? 4 : 3);
                j += len;
                if ((flags & kPrimLast) || j >= ns) break;
            }
            const int64_t len = j - i;
            int64_t posn = (int64_t)streamNew.size();
            if (len <= 8 && (posn & 7) + len > 8) posn = (posn + 7) & ~(int64_t)7;
            streamNew.resize((size_t)posn, float4{0, 0, 0, 0});
            slotAt[(size_t)i] = posn;
            streamNew.insert(streamNew.end(), stream.begin() + i, stream.begin() + j);
            i = j;
        }
    }
    auto leaf_ref = [&](int ref) {
        if (ref >= 0 || !alignLeaves) return ref;
        return (int)~slotAt[(size_t)~ref];
    };
    std::vector<WideNode> wNew((size_t)std::max(nNew, 1));
    std::memset(wNew.data(), 0, wNew.size() * sizeof(WideNode));
    for (int i = 0; i < n; ++i) {
        WideNode r = w[i];
        r.ref0 = r.ref0 >= 0 ? at[r.ref0] : leaf_ref(r.ref0);
        r.ref1 = r.ref1 >= 0 ? at[r.ref1] : leaf_ref(r.ref1);
        wNew[(size_t)at[i]] = r;
    }
    // holes keep a harmless self-contained record (never referenced)
    if (*root_ref < 0) *root_ref = leaf_ref(*root_ref);
    const std::vector<float4> &so = alignLeaves ? streamNew : stream;
    const int64_t nsNew = (int64_t)so.size();
    if (nsNew >= 0x7ffffffe) {
        *error = "relayout: primitive stream exceeds 2^31 slots";
        return false;
    }
    const size_t wideBytes = (wNew.size() * sizeof(WideNode) + 255) & ~(size_t)255;
    void *arena = nullptr;
    e = hipMalloc(&arena, wideBytes + (size_t)nsNew * 16 + 64);
    if (e != hipSuccess) return fail("hipMalloc", e);
    char *ps = (char *)arena + wideBytes;
    e = hipMemcpy(arena, wNew.data(), wNew.size() * sizeof(WideNode), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ps, so.data(), (size_t)nsNew * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(ps + (size_t)nsNew * 16, 0, 64);
    if (e != hipSuccess) {
        (void)hipFree(arena);
        return fail("upload", e);
    }
    (void)hipFree(*d_wide);
    *d_wide = (float4 *)arena;
    *d_prims = (float4 *)ps;
    *n_interior = nNew;
    *n_slots = nsNew;
    return true;
}

}  // namespace nnbvh
