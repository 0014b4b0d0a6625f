// mesh.h -- host mirror of the reference's Mesh (mesh.h:6-33): the scene-side input of
// RayTracer::Init.  GL members (VAO/VBO/texture) are not part of the compute path and are
// dropped; everything RayTracer reads is here under the reference's names.
#pragma once
#include "template_types.h"
#include "bvh.h"

#include <string>

namespace Tmpl8 {

// 64 bytes, same offsets as the reference's union-of-__m128 version (mesh.h:6-13):
// vertex0 @0, vertex1 @16, vertex2 @32, centroid @48.  Pads are zeroed here (the reference
// leaves them uninitialised).
struct alignas(64) Tri {
    float3_strict vertex0; float pad0;
    float3_strict vertex1; float pad1;
    float3_strict vertex2; float pad2;
    float3_strict centroid; float pad3;
};
static_assert(sizeof(Tri) == 64, "Tri is 64 bytes");

class Mesh {
public:
    Mesh() = default;
    ~Mesh();
    Mesh(const Mesh&) = delete;
    Mesh& operator=(const Mesh&) = delete;

    // mesh.cpp:5-98: "rooms/<modelFile>.glb", primitive 0 of mesh 0, POSITION + u16/u32
    // indices, then DetermineFloorHeight() and the BVH.  Returns false (and prints, like the
    // reference) when the file cannot be parsed.
    bool LoadMesh();
    // same, from an explicit path
    bool LoadMeshFromFile(const char* path);
    // adopt an already expanded triangle list (synthetic scenes, calibration tests)
    void SetTriangles(const Tri* tris, int count);
    void DetermineFloorHeight();   // mesh.cpp:100-136

    char modelFile[32] = "C046_1";      // mesh.h:21
    std::string roomsDir = "rooms/";    // mesh.cpp:13 prefix

    Tri* triangles = nullptr;
    int triangleCount = 0;
    float* vertices = nullptr;   // 9 floats per triangle (mesh.cpp:72-80)
    int vertexCount = 0;         // number of floats in `vertices` (mesh.cpp:89)
    uint dosageBufferID = 0;     // GL buffer id in the reference; unused here
    bool loadedMesh = false;
    float floorHeight = 0.0f;

    BVH* bvh = nullptr;
    std::string lastError;

private:
    void release();
    void finish();
};

}  // namespace Tmpl8
