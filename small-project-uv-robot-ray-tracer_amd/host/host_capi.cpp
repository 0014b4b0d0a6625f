// host_capi.cpp -- flat C wrappers over Mesh / BVH / RayTracer so that tests and bench.py can
// drive the C++ host layer through ctypes.  Plumbing only; no arithmetic lives here.
#include "raytracer.h"

#include <cstring>
#include <vector>

using namespace Tmpl8;

extern "C" {

// ---- Mesh ----
void* uvrt_host_mesh_load(const char* glb_path)
{
    Mesh* m = new Mesh();
    if (!m->LoadMeshFromFile(glb_path)) { delete m; return nullptr; }
    return m;
}
void* uvrt_host_mesh_from_tris(const void* tris64, int count)
{
    Mesh* m = new Mesh();
    m->SetTriangles((const Tri*)tris64, count);
    return m;
}
void uvrt_host_mesh_free(void* m) { delete (Mesh*)m; }
int uvrt_host_mesh_tri_count(void* m) { return ((Mesh*)m)->triangleCount; }
float uvrt_host_mesh_floor_height(void* m) { return ((Mesh*)m)->floorHeight; }
const void* uvrt_host_mesh_tris(void* m) { return ((Mesh*)m)->triangles; }
const void* uvrt_host_mesh_nodes(void* m) { return ((Mesh*)m)->bvh->bvhNode; }
unsigned uvrt_host_mesh_nodes_used(void* m) { return ((Mesh*)m)->bvh->nodesUsed; }
const unsigned* uvrt_host_mesh_tri_idx(void* m) { return ((Mesh*)m)->bvh->triIdx; }
void uvrt_host_mesh_rebuild_bvh(void* m) { ((Mesh*)m)->bvh->Build(); }

// ---- RayTracer ----
void* uvrt_host_rt_new(void) { return new RayTracer(); }
void uvrt_host_rt_free(void* r) { delete (RayTracer*)r; }
void uvrt_host_rt_set_route_dir(void* r, const char* dir) { ((RayTracer*)r)->routeDir = dir; }
void uvrt_host_rt_set_default_route(void* r, const char* name)
{
    RayTracer* rt = (RayTracer*)r;
    strncpy(rt->defaultRouteFile, name, 31);
    rt->defaultRouteFile[31] = 0;
}
void uvrt_host_rt_set_device(void* r, int dev) { ((RayTracer*)r)->deviceId = dev; }
void uvrt_host_rt_set_auto_save(void* r, int on) { ((RayTracer*)r)->autoSaveRoute = on != 0; }
void uvrt_host_rt_init(void* r, void* mesh) { ((RayTracer*)r)->Init((Mesh*)mesh); }
void uvrt_host_rt_load_route(void* r, const char* name)
{
    char buf[32];
    strncpy(buf, name, 31);
    buf[31] = 0;
    ((RayTracer*)r)->LoadRoute(buf);
}
void uvrt_host_rt_save_route(void* r, const char* name)
{
    char buf[32];
    strncpy(buf, name, 31);
    buf[31] = 0;
    ((RayTracer*)r)->SaveRoute(buf);
}
void uvrt_host_rt_update_photons_per_light(void* r) { ((RayTracer*)r)->UpdatePhotonsPerLight(); }
void uvrt_host_rt_reset_dosage_map(void* r) { ((RayTracer*)r)->ResetDosageMap(); }
void uvrt_host_rt_clear_buffers(void* r, int reset_color) { ((RayTracer*)r)->ClearBuffers(reset_color != 0); }
void uvrt_host_rt_compute_dosage_map(void* r) { ((RayTracer*)r)->ComputeDosageMap(); }
void uvrt_host_rt_compute_single(void* r, float x, float y, float duration, int photons, int tris)
{
    LightPos lp;
    lp.position = make_float2(x, y);
    lp.duration = duration;
    ((RayTracer*)r)->ComputeSingleLightDosageMap(lp, photons, tris);
}
void uvrt_host_rt_shade(void* r) { ((RayTracer*)r)->Shade(); }
void uvrt_host_rt_add_lamp(void* r) { ((RayTracer*)r)->AddLamp(); }
void uvrt_host_rt_calibrate(void* r, float p, float h, float d) { ((RayTracer*)r)->CalibratePower(p, h, d); }
void uvrt_host_rt_sync(void* r) { ((RayTracer*)r)->Sync(); }
void uvrt_host_rt_read_dosage(void* r, float* out, int first, int count) { ((RayTracer*)r)->ReadDosage(out, first, count); }
void* uvrt_host_rt_ctx(void* r) { return ((RayTracer*)r)->ctx; }
void uvrt_host_rt_set_shard(void* r, int rank, int world)
{
    ((RayTracer*)r)->shardRank = rank;
    ((RayTracer*)r)->shardWorld = world;
    ((RayTracer*)r)->launchIndex = 0;
}

void uvrt_host_rt_compute_batched(void* r, int iterations) { ((RayTracer*)r)->ComputeIterationsBatched(iterations); }
void uvrt_host_rt_compute_batched_group(void** rs, int n, int iterations)
{
    std::vector<RayTracer*> g;
    for (int i = 0; i < n; ++i) g.push_back((RayTracer*)rs[i]);
    RayTracer::ComputeIterationsBatched(g, iterations);
}
void uvrt_host_rt_set_ray_range(void* r, int rank, int world) { ((RayTracer*)r)->SetRayRange(rank, world); }
void uvrt_host_rt_set_reduce_over_comm(void* r, int on) { ((RayTracer*)r)->reduceOverComm = on != 0; }

int uvrt_host_rt_lamp_count(void* r) { return (int)((RayTracer*)r)->lightPositions.size(); }
void uvrt_host_rt_get_lamp(void* r, int i, float* xyd)
{
    const LightPos& lp = ((RayTracer*)r)->lightPositions[i];
    xyd[0] = lp.position.x; xyd[1] = lp.position.y; xyd[2] = lp.duration;
}
void uvrt_host_rt_set_lamps(void* r, const float* xyd, int n)
{
    RayTracer* rt = (RayTracer*)r;
    rt->lightPositions.clear();
    for (int i = 0; i < n; ++i) {
        LightPos lp;
        lp.position = make_float2(xyd[3 * i], xyd[3 * i + 1]);
        lp.duration = xyd[3 * i + 2];
        rt->lightPositions.push_back(lp);
    }
    rt->UpdatePhotonsPerLight();
}

// scalar fields, by name (keeps the ctypes surface small)
static int field(RayTracer* rt, const char* n, double* v, int set)
{
#define F(name, type) if (!strcmp(n, #name)) { if (set) rt->name = (type)*v; else *v = (double)rt->name; return 0; }
    F(lightLength, float) F(lightHeight, float) F(maxPhotonCount, int) F(photonCount, int)
    F(maxIterations, int) F(currIterations, int) F(lightIntensity, float) F(minDosage, float)
    F(minPower, float) F(photonsPerLight, int) F(compTime, float) F(progress, float)
    F(finishedComputation, bool) F(thresholdView, bool) F(startedComputation, bool)
    F(calibratedPower, float) F(photonMapSize, int)
#undef F
    if (!strcmp(n, "viewMode")) { if (set) rt->viewMode = (ViewMode)(int)*v; else *v = (double)rt->viewMode; return 0; }
    return -1;
}
int uvrt_host_rt_get(void* r, const char* name, double* v) { return field((RayTracer*)r, name, v, 0); }
int uvrt_host_rt_set(void* r, const char* name, double v) { return field((RayTracer*)r, name, &v, 1); }

}  // extern "C"
