// bvh.h -- host mirror of the reference's flat BVH (bvh.h:11-49): 32-byte nodes, children
// adjacent, in-place triIdx partition.  The builder reproduces the reference's node numbering
// and triIdx order exactly (closest-hit ties are broken by traversal order),
// but is written for a many-core host: explicit work stack instead of recursion, deferred
// depth-4 sub-trees built as OpenMP tasks into pre-reserved node ranges.
#pragma once
#include "template_types.h"

namespace Tmpl8 {
class Mesh;
}

// bvh.h:11-21
struct BVHNode {
    float3_strict aabbMin; uint leftFirst;
    float3_strict aabbMax; uint triCount;
    bool isLeaf() const { return triCount > 0; }
};
static_assert(sizeof(BVHNode) == 32, "BVHNode is 32 bytes");

#define BINS 8   // bvh.h:26

class BVH {
public:
    BVH() = default;
    explicit BVH(Tmpl8::Mesh* mesh);   // bvh.cpp:5-11: allocate + Build()
    ~BVH();
    BVH(const BVH&) = delete;
    BVH& operator=(const BVH&) = delete;
    void Build();                      // bvh.cpp:13-44

    uint* triIdx = nullptr;
    // Highest written node index + 1.  The reference reports 2*T here (bvh.cpp:43) although
    // its last sub-tree job writes past that; the pool here is 2*T+64 nodes and
    // nodesUsed is the true extent, so the whole tree is uploaded.
    uint nodesUsed = 0;
    BVHNode* bvhNode = nullptr;

private:
    Tmpl8::Mesh* mesh = nullptr;
    uint poolSize = 0;
};
