// bvh.cpp -- native binned-SAH BVH2 builder producing the reference's flat layout
// (reference: bvh.cpp:5-220, behaviour summarised in SURVEY.md 2.2).
//
// Same decisions, same f32 arithmetic (8 centroid bins per axis, vertex-AABB bins, the
// reference's right-sweep box indexing, strict `<` on the plane cost, no traversal-cost term,
// in-place two-cursor partition, children allocated adjacent), so node numbering and triIdx
// order are identical.  Different structure: an explicit LIFO work list replaces recursion,
// split-time computation of both children's bounds replaces the interleaved
// UpdateNodeBounds calls, centroids are kept SoA for the binning loops, and the sub-trees
// rooted at tree depth 4 run as independent OpenMP tasks inside pre-reserved node ranges
// (numbering therefore does not depend on the thread count).
#include "bvh.h"
#include "mesh.h"

#include <cstdlib>
#include <cstring>
#include <vector>

using namespace Tmpl8;

namespace {

inline float lane_min(float a, float b) { return a < b ? a : b; }   // _mm_min_ps lane
inline float lane_max(float a, float b) { return a > b ? a : b; }   // _mm_max_ps lane

struct Box3 {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; ++a) { lo[a] = 1e30f; hi[a] = -1e30f; } }
    void grow(const float3_strict& v)
    {
        lo[0] = lane_min(lo[0], v.x); hi[0] = lane_max(hi[0], v.x);
        lo[1] = lane_min(lo[1], v.y); hi[1] = lane_max(hi[1], v.y);
        lo[2] = lane_min(lo[2], v.z); hi[2] = lane_max(hi[2], v.z);
    }
    void grow(const Box3& b)
    {
        for (int a = 0; a < 3; ++a) { lo[a] = lane_min(lo[a], b.lo[a]); hi[a] = lane_max(hi[a], b.hi[a]); }
    }
    float halfArea() const
    {
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return ex * ey + ey * ez + ez * ex;
    }
};

struct Work {
    uint node;
    uint depth;
    float cmin[3], cmax[3];   // centroid bounds of the node's triangles
};

struct Builder {
    Tri* tris;
    uint* triIdx;
    BVHNode* nodes;
    uint pool;
    const float* cen[3];      // SoA centroids by triangle id
    bool overflow = false;

    // vertex AABB into the node, centroid bounds out (bvh.cpp:181-200)
    void bounds(uint nodeIdx, float cmin[3], float cmax[3]) const
    {
        BVHNode& n = nodes[nodeIdx];
        Box3 vb, cb;
        vb.clear();
        cb.clear();
        for (uint i = 0; i < n.triCount; ++i) {
            const uint id = triIdx[n.leftFirst + i];
            const Tri& t = tris[id];
            vb.grow(t.vertex0);
            vb.grow(t.vertex1);
            vb.grow(t.vertex2);
            for (int a = 0; a < 3; ++a) {
                cb.lo[a] = lane_min(cb.lo[a], cen[a][id]);
                cb.hi[a] = lane_max(cb.hi[a], cen[a][id]);
            }
        }
        n.aabbMin = make_float3_strict(vb.lo[0], vb.lo[1], vb.lo[2]);
        n.aabbMax = make_float3_strict(vb.hi[0], vb.hi[1], vb.hi[2]);
        for (int a = 0; a < 3; ++a) { cmin[a] = cb.lo[a]; cmax[a] = cb.hi[a]; }
    }

    static inline int bin_of(float c, float lo, float scale)
    {
        const int b = (int)((c - lo) * scale);
        return b < BINS - 1 ? b : BINS - 1;
    }

    // bvh.cpp:98-179
    float best_split(const BVHNode& n, int& axis, int& splitPos, const float cmin[3], const float cmax[3]) const
    {
        float best = 1e30f;
        for (int a = 0; a < 3; ++a) {
            const float lo = cmin[a], hi = cmax[a];
            if (lo == hi) continue;
            const float scale = (float)BINS / (hi - lo);
            Box3 bin[BINS];
            uint cnt[BINS];
            for (int b = 0; b < BINS; ++b) { bin[b].clear(); cnt[b] = 0; }
            const float* ca = cen[a];
            for (uint i = 0; i < n.triCount; ++i) {
                const uint id = triIdx[n.leftFirst + i];
                const int b = bin_of(ca[id], lo, scale);
                const Tri& t = tris[id];
                ++cnt[b];
                bin[b].grow(t.vertex0);
                bin[b].grow(t.vertex1);
                bin[b].grow(t.vertex2);
            }
            // Sweep.  The reference accumulates the right-hand COUNT over bins 7, 6, ... but
            // the right-hand BOX over bins 6, 5, ... (bvh.cpp:134-138); kept as is.
            float leftCost[BINS - 1], rightCost[BINS - 1];
            Box3 lb, rb;
            lb.clear();
            rb.clear();
            int lsum = 0, rsum = 0;
            for (int i = 0; i < BINS - 1; ++i) {
                lsum += (int)cnt[i];
                rsum += (int)cnt[BINS - 1 - i];
                lb.grow(bin[i]);
                rb.grow(bin[BINS - 2 - i]);
                leftCost[i] = (float)lsum * lb.halfArea();
                rightCost[BINS - 2 - i] = (float)rsum * rb.halfArea();
            }
            for (int i = 0; i < BINS - 1; ++i) {
                const float cost = leftCost[i] + rightCost[i];
                if (cost < best) { axis = a; splitPos = i + 1; best = cost; }
            }
        }
        return best;
    }

    // Subdivide the sub-tree under `root` (bvh.cpp:46-96).  `next` is the allocation cursor of
    // this sub-tree's node range.  When `defer` is given, children created by a depth-3 split
    // are queued there instead of being processed (bvh.cpp:79-94).
    void run(const Work& root, uint& next, std::vector<Work>* defer, uint& extent)
    {
        std::vector<Work> todo;
        todo.reserve(64);
        todo.push_back(root);
        while (!todo.empty()) {
            const Work w = todo.back();
            todo.pop_back();
            BVHNode& n = nodes[w.node];
            int axis = 0, splitPos = 0;
            const float splitCost = best_split(n, axis, splitPos, w.cmin, w.cmax);
            const float ex = n.aabbMax.x - n.aabbMin.x, ey = n.aabbMax.y - n.aabbMin.y,
                        ez = n.aabbMax.z - n.aabbMin.z;
            const float nosplitCost = (ex * ey + ey * ez + ez * ex) * (float)n.triCount;   // bvh.h:16-20
            if (splitCost >= nosplitCost) continue;
            // in-place partition (bvh.cpp:56-64)
            int i = (int)n.leftFirst;
            int j = i + (int)n.triCount - 1;
            const float lo = w.cmin[axis];
            const float scale = (float)BINS / (w.cmax[axis] - lo);
            const float* ca = cen[axis];
            while (i <= j) {
                if (bin_of(ca[triIdx[i]], lo, scale) < splitPos) ++i;
                else { const uint t = triIdx[i]; triIdx[i] = triIdx[j]; triIdx[j] = t; --j; }
            }
            const int leftCount = i - (int)n.leftFirst;
            if (leftCount == 0 || leftCount == (int)n.triCount) continue;
            if (next + 2 > pool) { overflow = true; return; }
            const uint left = next++, right = next++;
            if (right + 1 > extent) extent = right + 1;
            nodes[left].leftFirst = n.leftFirst;
            nodes[left].triCount = (uint)leftCount;
            nodes[right].leftFirst = (uint)i;
            nodes[right].triCount = n.triCount - (uint)leftCount;
            n.leftFirst = left;
            n.triCount = 0;
            Work wl, wr;
            wl.node = left;
            wr.node = right;
            wl.depth = wr.depth = w.depth + 1;
            bounds(left, wl.cmin, wl.cmax);
            bounds(right, wr.cmin, wr.cmax);
            if (defer && w.depth == 3) {
                defer->push_back(wl);
                defer->push_back(wr);
            } else {
                todo.push_back(wr);   // LIFO: the left sub-tree is numbered first
                todo.push_back(wl);
            }
        }
    }
};

}  // namespace

BVH::BVH(Mesh* m)
{
    mesh = m;
    poolSize = (uint)m->triangleCount * 2 + 64;
    bvhNode = (BVHNode*)aligned_alloc(64, sizeof(BVHNode) * (size_t)poolSize);
    triIdx = new uint[m->triangleCount];
    Build();
}

BVH::~BVH()
{
    free(bvhNode);
    delete[] triIdx;
}

void BVH::Build()
{
    const int T = mesh->triangleCount;
    Tri* tris = mesh->triangles;
    memset(bvhNode, 0, sizeof(BVHNode) * (size_t)poolSize);
    std::vector<float> cx(T), cy(T), cz(T);
    for (int i = 0; i < T; ++i) {
        triIdx[i] = (uint)i;
        // bvh.cpp:23 -- (v0 + v1 + v2) * 0.3333f per component
        Tri& t = tris[i];
        t.centroid.x = cx[i] = (t.vertex0.x + t.vertex1.x + t.vertex2.x) * 0.3333f;
        t.centroid.y = cy[i] = (t.vertex0.y + t.vertex1.y + t.vertex2.y) * 0.3333f;
        t.centroid.z = cz[i] = (t.vertex0.z + t.vertex1.z + t.vertex2.z) * 0.3333f;
    }
    Builder b;
    b.tris = tris;
    b.triIdx = triIdx;
    b.nodes = bvhNode;
    b.pool = poolSize;
    b.cen[0] = cx.data();
    b.cen[1] = cy.data();
    b.cen[2] = cz.data();

    bvhNode[0].leftFirst = 0;
    bvhNode[0].triCount = (uint)T;
    Work root;
    root.node = 0;
    root.depth = 0;
    b.bounds(0, root.cmin, root.cmax);

    uint next = 2;   // node 1 stays unused so sibling pairs are 64-byte aligned (bvh.cpp:16)
    uint extent = 1;
    std::vector<Work> jobs;
    b.run(root, next, &jobs, extent);

    // pre-reserved node ranges (bvh.cpp:33-36), then the sub-trees in parallel
    const int N = (int)jobs.size();
    std::vector<uint> cursor(N > 0 ? N : 1), ext(N > 0 ? N : 1, 0u);
    if (N > 0) cursor[0] = next;
    for (int i = 1; i < N; ++i) cursor[i] = cursor[i - 1] + bvhNode[jobs[i - 1].node].triCount * 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < N; ++i) {
        Work w = jobs[i];
        w.depth = 99;
        Builder local = b;
        local.run(w, cursor[i], nullptr, ext[i]);
        if (local.overflow) {
#pragma omp critical
            b.overflow = true;
        }
    }
    for (int i = 0; i < N; ++i)
        if (ext[i] > extent) extent = ext[i];
    nodesUsed = b.overflow ? 0 : extent;
}
