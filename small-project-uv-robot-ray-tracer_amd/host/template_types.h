// template_types.h -- the few types of the reference's template/precomp.h that the
// RayTracer / Mesh / BVH surface exposes (float2, float3, float3_strict, uint, Timer), with
// the same names, layouts and namespace, so callers written against the reference compile.
//   float3 / float3_strict layouts: template/precomp.h:166-181
//   Timer:                          template/precomp.h:277-288
#pragma once
#include <chrono>
#include <cstdint>

typedef unsigned int uint;

struct float2 {
    float x, y;
};
inline float2 make_float2(float a, float b) { float2 f; f.x = a; f.y = b; return f; }

struct alignas(16) float3 {
    float x, y, z, dummy;
};
inline float3 make_float3(float a, float b, float c) { float3 f; f.x = a; f.y = b; f.z = c; f.dummy = 0; return f; }

struct float3_strict {
    float x, y, z;
    float operator[](int n) const { return (&x)[n]; }
};
inline float3_strict make_float3_strict(float a, float b, float c) { float3_strict f; f.x = a; f.y = b; f.z = c; return f; }

namespace Tmpl8 {

struct Timer {
    Timer() { reset(); }
    float elapsed() const
    {
        auto t2 = std::chrono::high_resolution_clock::now();
        return (float)std::chrono::duration_cast<std::chrono::duration<double>>(t2 - start).count();
    }
    void reset() { start = std::chrono::high_resolution_clock::now(); }
    std::chrono::high_resolution_clock::time_point start;
};

}  // namespace Tmpl8
