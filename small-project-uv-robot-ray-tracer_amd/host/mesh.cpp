// mesh.cpp -- scene input of the hot path: binary glTF -> Tri[] + floor height + BVH
// (reference: mesh.cpp:5-136).  A minimal GLB reader replaces tinygltf: 12-byte header, JSON
// chunk, BIN chunk; only what the reference reads is interpreted -- meshes[0].primitives[0],
// its POSITION accessor (VEC3 f32) and its index accessor (u16 or u32); no node transforms, no
// byteStride (mesh.cpp:28-51).
#include "mesh.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <vector>

using namespace Tmpl8;

namespace {

// ---- a small JSON value tree (objects, arrays, numbers, strings, literals) ----
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;

    const JValue* get(const char* key) const
    {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const JValue* at(size_t i) const { return (kind == Arr && i < arr.size()) ? &arr[i] : nullptr; }
    long long integer(long long dflt) const { return kind == Num ? (long long)num : dflt; }
};

struct JParser {
    const char* p;
    const char* end;
    bool ok = true;

    void ws() { while (p < end && isspace((unsigned char)*p)) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }

    JValue value()
    {
        JValue v;
        ws();
        if (p >= end) { ok = false; return v; }
        if (*p == '{') {
            ++p;
            v.kind = JValue::Obj;
            if (eat('}')) return v;
            do {
                ws();
                JValue k = value();
                if (k.kind != JValue::Str || !eat(':')) { ok = false; return v; }
                v.obj.emplace_back(k.str, value());
                if (!ok) return v;
            } while (eat(','));
            if (!eat('}')) ok = false;
        } else if (*p == '[') {
            ++p;
            v.kind = JValue::Arr;
            if (eat(']')) return v;
            do {
                v.arr.push_back(value());
                if (!ok) return v;
            } while (eat(','));
            if (!eat(']')) ok = false;
        } else if (*p == '"') {
            ++p;
            v.kind = JValue::Str;
            while (p < end && *p != '"') {
                if (*p == '\\' && p + 1 < end) {
                    ++p;
                    switch (*p) {
                        case 'n': v.str += '\n'; break;
                        case 't': v.str += '\t'; break;
                        case 'u': v.str += '?'; p += (end - p > 4) ? 4 : 0; break;
                        default: v.str += *p;
                    }
                    ++p;
                } else v.str += *p++;
            }
            if (p >= end) ok = false; else ++p;
        } else if (!strncmp(p, "true", 4)) { v.kind = JValue::Bool; v.b = true; p += 4; }
        else if (!strncmp(p, "false", 5)) { v.kind = JValue::Bool; p += 5; }
        else if (!strncmp(p, "null", 4)) { p += 4; }
        else {
            char* q = nullptr;
            v.num = strtod(p, &q);
            if (q == p) { ok = false; return v; }
            v.kind = JValue::Num;
            p = q;
        }
        return v;
    }
};

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

Mesh::~Mesh() { release(); }

void Mesh::release()
{
    delete bvh;
    bvh = nullptr;
    free(triangles);
    triangles = nullptr;
    delete[] vertices;
    vertices = nullptr;
    triangleCount = vertexCount = 0;
    loadedMesh = false;
}

bool Mesh::LoadMesh()
{
    const std::string path = roomsDir + modelFile + ".glb";   // mesh.cpp:13-14
    return LoadMeshFromFile(path.c_str());
}

bool Mesh::LoadMeshFromFile(const char* path)
{
    std::cout << "Loading mesh " << std::endl;
    std::ifstream f(path, std::ios::binary);
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto bad = [&](const char* why) {
        lastError = std::string(why) + ": " + path;
        printf("Err: %s\n", lastError.c_str());
        printf("Failed to parse glTF\n");   // mesh.cpp:23-26
        return false;
    };
    if (!f || buf.size() < 20) return bad("cannot read file");
    if (memcmp(buf.data(), "glTF", 4) != 0) return bad("not a binary glTF");
    const uint32_t jlen = rd32(&buf[12]);
    if (memcmp(&buf[16], "JSON", 4) != 0 || 20 + (size_t)jlen + 8 > buf.size()) return bad("missing JSON chunk");
    JParser jp{(const char*)&buf[20], (const char*)&buf[20] + jlen};
    const JValue doc = jp.value();
    if (!jp.ok) return bad("malformed JSON chunk");
    const size_t boff = 20 + jlen;
    const uint32_t blen = rd32(&buf[boff]);
    if (memcmp(&buf[boff + 4], "BIN\0", 4) != 0 || boff + 8 + (size_t)blen > buf.size()) return bad("missing BIN chunk");
    const unsigned char* bin = &buf[boff + 8];

    const JValue* meshes = doc.get("meshes");
    const JValue* prim = meshes && meshes->at(0) && meshes->at(0)->get("primitives")
                             ? meshes->at(0)->get("primitives")->at(0) : nullptr;
    const JValue* accessors = doc.get("accessors");
    const JValue* views = doc.get("bufferViews");
    if (!prim || !accessors || !views) return bad("no meshes[0].primitives[0]");
    const JValue* attrs = prim->get("attributes");
    const JValue* posIdx = attrs ? attrs->get("POSITION") : nullptr;
    const JValue* indIdx = prim->get("indices");
    if (!posIdx || !indIdx) return bad("primitive lacks POSITION or indices");
    const JValue* pacc = accessors->at((size_t)posIdx->integer(-1));
    const JValue* iacc = accessors->at((size_t)indIdx->integer(-1));
    if (!pacc || !iacc) return bad("accessor out of range");
    auto view_offset = [&](const JValue* acc, size_t& off, size_t& avail) -> bool {
        const JValue* bv = acc->get("bufferView");
        const JValue* view = bv ? views->at((size_t)bv->integer(-1)) : nullptr;
        if (!view) return false;
        const long long vo = view->get("byteOffset") ? view->get("byteOffset")->integer(0) : 0;
        const long long ao = acc->get("byteOffset") ? acc->get("byteOffset")->integer(0) : 0;
        if (vo < 0 || ao < 0 || (size_t)(vo + ao) > blen) return false;
        off = (size_t)(vo + ao);
        avail = blen - off;
        return true;
    };
    size_t poff, pavail, ioff, iavail;
    if (!view_offset(pacc, poff, pavail) || !view_offset(iacc, ioff, iavail)) return bad("bad bufferView");
    const long long pcount = pacc->get("count") ? pacc->get("count")->integer(0) : 0;
    const long long icount = iacc->get("count") ? iacc->get("count")->integer(0) : 0;
    const long long ctype = iacc->get("componentType") ? iacc->get("componentType")->integer(0) : 0;
    const bool shortIndices = ctype == 5123;              // TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT
    if (!shortIndices && ctype != 5125) return bad("indices must be u16 or u32");
    if ((size_t)pcount * 12 > pavail || (size_t)icount * (shortIndices ? 2 : 4) > iavail) return bad("accessor beyond BIN chunk");
    const long long T = icount / 3;
    if (T <= 0) return bad("no triangles");

    std::vector<float> positions((size_t)pcount * 3);
    memcpy(positions.data(), bin + poff, (size_t)pcount * 12);
    auto index_at = [&](long long k) -> long long {
        if (shortIndices) { uint16_t v; memcpy(&v, bin + ioff + 2 * k, 2); return v; }
        uint32_t v; memcpy(&v, bin + ioff + 4 * k, 4); return v;
    };

    release();
    triangles = (Tri*)aligned_alloc(64, sizeof(Tri) * (size_t)T);
    memset((void*)triangles, 0, sizeof(Tri) * (size_t)T);
    for (long long i = 0; i < T; ++i) {                   // mesh.cpp:56-71
        float3_strict* v[3] = {&triangles[i].vertex0, &triangles[i].vertex1, &triangles[i].vertex2};
        for (int k = 0; k < 3; ++k) {
            const long long id = index_at(i * 3 + k);
            if (id < 0 || id >= pcount) { release(); return bad("vertex index out of range"); }
            *v[k] = make_float3_strict(positions[id * 3 + 0], positions[id * 3 + 1], positions[id * 3 + 2]);
        }
    }
    triangleCount = (int)T;
    finish();
    return true;
}

void Mesh::SetTriangles(const Tri* tris, int count)
{
    release();
    triangles = (Tri*)aligned_alloc(64, sizeof(Tri) * (size_t)count);
    memcpy((void*)triangles, tris, sizeof(Tri) * (size_t)count);
    triangleCount = count;
    finish();
}

void Mesh::finish()
{
    vertexCount = triangleCount * 9;                      // mesh.cpp:89 (= indices.count * 3)
    vertices = new float[(size_t)vertexCount];
    for (int i = 0; i < triangleCount; ++i) {             // mesh.cpp:72-80
        const Tri& t = triangles[i];
        float* o = vertices + (size_t)i * 9;
        o[0] = t.vertex0.x; o[1] = t.vertex0.y; o[2] = t.vertex0.z;
        o[3] = t.vertex1.x; o[4] = t.vertex1.y; o[5] = t.vertex1.z;
        o[6] = t.vertex2.x; o[7] = t.vertex2.y; o[8] = t.vertex2.z;
    }
    DetermineFloorHeight();
    std::cout << "Vertex count: " << vertexCount << " triangle count: " << triangleCount << std::endl;
    bvh = new BVH(this);
    std::cout << "BVH size: " << bvh->nodesUsed << std::endl;
    loadedMesh = true;
}

// mesh.cpp:100-136: floor = centre of the fullest of 48 height bins between the lowest vertex
// and y = 0 (maxVal is never raised above 0 in the reference; kept).  A vertex is counted in
// bin j when  j*range/48 + min < y < (j+1)*range/48 + min  (strict on both sides, f32).
// The reference scans all 48 bins per vertex; here the candidate bin is located directly and
// its two neighbours are tested with the very same f32 expressions, which gives the same counts.
void Mesh::DetermineFloorHeight()
{
    const int binCount = 48;
    float maxVal = 0.0f, minVal = 0.0f;
    int hist[binCount] = {0};
    const int nv = vertexCount / 3;
    for (int i = 0; i < nv; ++i)
        if (vertices[i * 3 + 1] < minVal) minVal = vertices[i * 3 + 1];
    const float range = maxVal - minVal;
    auto lower = [&](int j) { return (float)j * range / (float)binCount + minVal; };
    for (int i = 0; i < nv; ++i) {
        const float y = vertices[i * 3 + 1];
        if (!(range > 0.0f)) break;
        float g = (y - minVal) / range * (float)binCount;
        if (!(g > -4.0f)) g = -4.0f;
        if (g > (float)binCount + 4.0f) g = (float)binCount + 4.0f;
        const int guess = (int)g;
        for (int j = guess - 2; j <= guess + 2; ++j) {
            if (j < 0 || j >= binCount) continue;
            if (lower(j) < y && y < lower(j + 1)) hist[j]++;
        }
    }
    int maxCount = 0, maxIndex = -1;
    for (int i = 0; i < binCount; ++i)
        if (hist[i] > maxCount) { maxIndex = i; maxCount = hist[i]; }
    floorHeight = ((float)maxIndex + 0.5f) * range / (float)binCount + minVal;
}
