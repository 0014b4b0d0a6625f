// raytracer.h -- host mirror of the reference's RayTracer (raytracer.h:5-59): same class
// name, namespace, method signatures and public data members, so MyApp / UserInterface code
// written against the reference drives this one unchanged.  The OpenCL `Kernel*` / `Buffer*`
// members are replaced by one opaque uvrt_ctx (include/uvrt.h); everything else is kept.
#pragma once
#include "template_types.h"
#include "mesh.h"

#include <string>
#include <vector>

struct uvrt_ctx;

namespace Tmpl8 {

struct LightPos {              // raytracer.h:5-9
    float2 position;
    float duration;
};

enum ViewMode { dosage, maxpower, texture };   // raytracer.h:11

class RayTracer {
public:
    RayTracer() = default;
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    void Init(Mesh* mesh);                                   // raytracer.cpp:12-59
    void UpdatePhotonsPerLight();                            // :61-64
    void ComputeDosageMap();                                 // :66-72
    void ComputeSingleLightDosageMap(LightPos lightPos, int photonsPerLight, int triangleCount);   // :75-88
    void Shade();                                            // :93-120
    void ResetDosageMap();                                   // :122-131
    void ClearBuffers(bool resetColor);                      // :133-143
    void AddLamp();                                          // :3-10
    void CalibratePower(float measurePower, float measureHeight, float measureDist);   // :151-227
    void SaveRoute(char fileName[32]);                       // :233-259
    void LoadRoute(char fileName[32]);                       // :261-300

    float lightLength = 1.0f;
    float lightHeight = 0.8f;
    int maxPhotonCount = (1 << 26);
    int photonCount = (1 << 25);
    int maxIterations = 10;
    int currIterations = 0;   // The number of computed iterations
    float lightIntensity = 450;
    float minDosage = 100, minPower = 1500;
    char defaultRouteFile[32] = "route";
    char newRouteFile[32] = "new_route";

    Mesh* mesh = nullptr;
    float* dosageMap = new float[2];
    std::vector<LightPos> lightPositions;
    int photonsPerLight = 0;   // The number of photons per light of a single iteration
    float compTime = 0;
    float progressTextTimer = 0;
    float progress = 0;
    Timer timerClock;
    bool finishedComputation = true;
    ViewMode viewMode = texture;
    bool thresholdView = false;
    bool startedComputation = false;
    float calibratedPower = 0;
    int photonMapSize = 0;
    void* simpleShader = nullptr;   // ShaderGL* in the reference; not used by the compute path

    // ---- additions of the headless build (not in the reference) ----
    uvrt_ctx* ctx = nullptr;            // replaces the six Kernel* and nine Buffer* members
    int deviceId = 0;                   // HIP device the context is created on
    std::string routeDir = "positions/";   // prefix of SaveRoute/LoadRoute (raytracer.cpp:257,263)
    bool autoSaveRoute = true;          // ResetDosageMap rewrites positions/route.xml (:126)
    // Multi-GPU launch sharding (DESIGN.md "Multi-GPU"): of the global sequence of lamp launches
    // (every ComputeSingleLightDosageMap call, in order) this instance runs those with
    // index % shardWorld == shardRank and only advances the SEED chain for the others, so the
    // union over ranks is the single-GPU computation.  The owner of the ranks then reduces
    // photonMap with SUM and maxPhotonMap with MAX (both exact) before Shade().
    int shardRank = 0, shardWorld = 1;
    long long launchIndex = 0;
    // Batched computation (include/uvrt.h "batched tracing"): what MyApp::Tick does over `iterations`
    // frames (myapp.cpp:156-163: ComputeDosageMap(); Shade(); currIterations++), with all launches of a
    // batch traced first and accumulate + Shade replayed per launch afterwards: same maps, dose and colours
    // bit for bit, far fewer kernel launches.  `group`: instances that together trace every launch by
    // global-id range (rangeFirst / rangeCount; BASELINE configs[3] "pixel tiles") inside ONE process --
    // their planes are summed by uvrt_reduce_batch_group; an instance whose context holds a communicator
    // (one process per GPU, uvrt_comm_init_rank) reduces over it.
    void ComputeIterationsBatched(int iterations);
    static void ComputeIterationsBatched(const std::vector<RayTracer*>& group, int iterations);
    long long rangeFirst = 0, rangeCount = -1;      // -1: the whole launch
    bool reduceOverComm = false;                    // ctx has a communicator: all-reduce the planes of every batch
    void SetRayRange(int rank, int world);          // contiguous share of [0, photonsPerLight) for rank of world
    // The reference never reads the dose back; the headless build does.
    void ReadDosage(float* out, int first, int count);
    void Sync();                        // clFinish(Kernel::GetQueue()), myapp.cpp:165
};

}  // namespace Tmpl8
