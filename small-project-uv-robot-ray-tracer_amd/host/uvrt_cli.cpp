// uvrt_cli.cpp -- headless equivalent of the reference's per-frame compute block
// (MyApp::Init myapp.cpp:36-39 + MyApp::Tick myapp.cpp:156-175): load the room, Init the
// RayTracer, ResetDosageMap, then {ComputeDosageMap; Shade; currIterations++; sync; progress
// line} until maxIterations, and dump the per-triangle dose.
//
//   uvrt_cli --room rooms/testroomopt.glb [--route-dir positions/] [--route lange_route]
//            [--photons N] [--iterations K] [--lamps L] [--view dosage|maxpower]
//            [--calibrate POWER HEIGHT DIST] [--device D] [--save-route name]
//            [--dump dose.f32 | dose.npy]   raw little-endian f32[T] or NumPy .npy (by extension)
//            [--ply heatmap.ply]            the room with per-triangle heat-map colours
//                                           (dosageToColor output; what the reference shows in GL)
//            [--flavour 0|1|2]              arithmetic of extend (include/uvrt.h uvrt_set_flavour): 0 strict (default), 1 the fused
//                                           forms of the reference's strict build on gfx950, 2 what the reference's OWN build
//                                           flags (-cl-fast-relaxed-math, template.cpp:1192) compute on gfx950 (+11 %)
//            [--batch K]                    trace K iterations per batch (RayTracer::ComputeIterationsBatched:
//                                           all launches first, accumulate + Shade replayed; same dose bits)
//            [--gpus N]                     one process, N contexts: every launch split by global-id range
//                                           ("pixel tiles"), ONE RCCL all-reduce of the count planes per batch
//                                           (contexts share a device when the box has fewer than N: no RCCL then)
#include "raytracer.h"
#include "../../include/uvrt.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

using namespace Tmpl8;

int main(int argc, char** argv)
{
    std::string room, routeDir = "positions/", route = "route", dump, saveRoute, ply;
    long long photons = -1;
    int iterations = -1, lamps = -1, device = 0, gpus = 1, batch = 0, flavour = 0;
    bool calibrate = false;
    float calP = 2909.0f, calH = 0.8f, calD = 1.0f;   // userinterface.cpp:107-109 defaults
    ViewMode view = dosage;
    for (int i = 1; i < argc; ++i) {
        auto need = [&](int k) { if (i + k >= argc) { fprintf(stderr, "missing value for %s\n", argv[i]); exit(2); } };
        if (!strcmp(argv[i], "--room")) { need(1); room = argv[++i]; }
        else if (!strcmp(argv[i], "--route-dir")) { need(1); routeDir = argv[++i]; }
        else if (!strcmp(argv[i], "--route")) { need(1); route = argv[++i]; }
        else if (!strcmp(argv[i], "--photons")) { need(1); photons = atoll(argv[++i]); }
        else if (!strcmp(argv[i], "--iterations")) { need(1); iterations = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--lamps")) { need(1); lamps = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--device")) { need(1); device = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--gpus")) { need(1); gpus = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--batch")) { need(1); batch = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--flavour")) { need(1); flavour = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--dump")) { need(1); dump = argv[++i]; }
        else if (!strcmp(argv[i], "--ply")) { need(1); ply = argv[++i]; }
        else if (!strcmp(argv[i], "--save-route")) { need(1); saveRoute = argv[++i]; }
        else if (!strcmp(argv[i], "--view")) { need(1); view = !strcmp(argv[++i], "maxpower") ? maxpower : dosage; }
        else if (!strcmp(argv[i], "--calibrate")) { need(3); calibrate = true; calP = (float)atof(argv[++i]); calH = (float)atof(argv[++i]); calD = (float)atof(argv[++i]); }
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (room.empty()) { fprintf(stderr, "usage: uvrt_cli --room file.glb [options]\n"); return 2; }
    if (!routeDir.empty() && routeDir.back() != '/') routeDir += '/';

    Mesh mesh;
    if (!mesh.LoadMeshFromFile(room.c_str())) return 1;

    RayTracer rayTracer;
    rayTracer.deviceId = device;
    rayTracer.routeDir = routeDir;
    rayTracer.autoSaveRoute = false;
    strncpy(rayTracer.defaultRouteFile, route.c_str(), 31);
    rayTracer.Init(&mesh);                                   // myapp.cpp:39
    if (rayTracer.lightPositions.empty()) rayTracer.AddLamp();
    if (lamps > 0 && lamps < (int)rayTracer.lightPositions.size()) rayTracer.lightPositions.resize(lamps);
    if (photons > 0) rayTracer.photonCount = (int)photons;
    if (iterations > 0) rayTracer.maxIterations = iterations;
    rayTracer.UpdatePhotonsPerLight();

    if (calibrate) {
        rayTracer.CalibratePower(calP, calH, calD);          // userinterface.cpp:130-133
        std::cout << "Calibrated lamp power: " << rayTracer.lightIntensity << std::endl;
    }

    // --gpus N: further instances of the same RayTracer, one per context, each with its range of every launch
    std::vector<RayTracer*> group{&rayTracer};
    std::vector<RayTracer*> extra;
    if (gpus < 1 || gpus > 64) { fprintf(stderr, "--gpus must be in [1,64]\n"); return 2; }
    if (gpus > 1) {
        const int ndev = uvrt_device_count();
        if (batch <= 0) batch = 1;
        for (int r = 1; r < gpus; ++r) {
            RayTracer* rt = new RayTracer();
            rt->deviceId = ndev >= gpus ? device + r : device;
            rt->routeDir = routeDir;
            rt->autoSaveRoute = false;
            strncpy(rt->defaultRouteFile, route.c_str(), 31);
            rt->Init(&mesh);
            rt->lightPositions = rayTracer.lightPositions;
            rt->photonCount = rayTracer.photonCount;
            rt->maxIterations = rayTracer.maxIterations;
            rt->lightIntensity = rayTracer.lightIntensity;
            rt->UpdatePhotonsPerLight();
            extra.push_back(rt);
            group.push_back(rt);
        }
        if (ndev >= gpus) {
            std::vector<uvrt_ctx*> ctxs;
            for (RayTracer* rt : group) ctxs.push_back(rt->ctx);
            if (uvrt_comm_init_all(ctxs.data(), gpus) != UVRT_OK) { fprintf(stderr, "comm_init_all: %s\n", uvrt_last_error()); return 1; }
            std::cout << "Sharding every launch over " << gpus << " GPUs, one RCCL all-reduce of the count planes per batch" << std::endl;
        } else {
            std::cout << "Sharding every launch over " << gpus << " contexts on " << ndev << " GPU(s) (rehearsal: no RCCL)" << std::endl;
        }
    }
    for (size_t r = 0; r < group.size(); ++r) {
        if (uvrt_set_flavour(group[r]->ctx, flavour) != UVRT_OK) { fprintf(stderr, "--flavour: %s\n", uvrt_last_error()); return 2; }
        group[r]->ResetDosageMap();                          // userinterface.cpp:247-251
        group[r]->viewMode = view;
        if (gpus > 1) group[r]->SetRayRange((int)r, gpus);
    }
    while (!rayTracer.finishedComputation) {                 // myapp.cpp:156-175
        rayTracer.finishedComputation = rayTracer.currIterations >= rayTracer.maxIterations;
        if (rayTracer.finishedComputation) break;
        if (batch > 0) {
            RayTracer::ComputeIterationsBatched(group, std::min(batch, rayTracer.maxIterations - rayTracer.currIterations));
        } else {
            rayTracer.ComputeDosageMap();
            rayTracer.Shade();
            rayTracer.currIterations++;
        }
        if (rayTracer.viewMode == texture) rayTracer.viewMode = dosage;
        rayTracer.progress = 100.0f * static_cast<float>(rayTracer.currIterations) / static_cast<float>(rayTracer.maxIterations);
        for (RayTracer* rt : group) rt->Sync();
        float time = rayTracer.timerClock.elapsed();
        rayTracer.compTime += time;
        std::cout << "Progress: " << rayTracer.progress << "% photon count: " << rayTracer.photonMapSize
                  << " delta time: " << time * 1000.0f << " total time: " << rayTracer.compTime * 1000.0f << std::endl;
        rayTracer.timerClock.reset();
    }
    const double rays = (double)rayTracer.photonMapSize;
    std::cout << "Traced " << rays << " photons in " << rayTracer.compTime * 1000.0f << " ms = "
              << rays / rayTracer.compTime / 1e6 << " Mray/s" << std::endl;

    std::vector<float> dose(mesh.triangleCount);
    rayTracer.ReadDosage(dose.data(), 0, mesh.triangleCount);
    double sum = 0;
    int nonzero = 0;
    for (float d : dose) { sum += d; nonzero += d != 0.0f; }
    printf("dose: sum %.6f, %d of %d triangles non-zero, dose[0..3] = %.9g %.9g %.9g %.9g\n", sum, nonzero,
           mesh.triangleCount, dose[0], dose.size() > 1 ? dose[1] : 0.f, dose.size() > 2 ? dose[2] : 0.f,
           dose.size() > 3 ? dose[3] : 0.f);
    if (!dump.empty()) {
        std::ofstream f(dump, std::ios::binary);
        if (dump.size() > 4 && dump.substr(dump.size() - 4) == ".npy") {
            // NumPy format 1.0: magic, version, header length, dict padded to a 64-byte boundary
            std::string hdr = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(dose.size()) + ",), }";
            while ((10 + hdr.size() + 1) % 64) hdr += ' ';
            hdr += '\n';
            const unsigned short hl = (unsigned short)hdr.size();
            f.write("\x93NUMPY\x01\x00", 8);
            f.write((const char*)&hl, 2);
            f.write(hdr.data(), (std::streamsize)hdr.size());
        }
        f.write((const char*)dose.data(), (std::streamsize)dose.size() * 4);
    }
    if (!ply.empty()) {
        std::vector<float> color((size_t)mesh.triangleCount * 9);
        if (uvrt_read_color(rayTracer.ctx, color.data(), 0, mesh.triangleCount) != UVRT_OK) {
            fprintf(stderr, "read_color: %s\n", uvrt_last_error());
            return 1;
        }
        std::ofstream f(ply, std::ios::binary);
        f << "ply\nformat binary_little_endian 1.0\ncomment UV dose heat map (dosageToColor)\n"
          << "element vertex " << mesh.triangleCount * 3 << "\nproperty float x\nproperty float y\nproperty float z\n"
          << "property uchar red\nproperty uchar green\nproperty uchar blue\n"
          << "element face " << mesh.triangleCount << "\nproperty list uchar int vertex_indices\nend_header\n";
        auto to8 = [](float v) { v = v < 0 ? 0 : (v > 1 ? 1 : v); return (unsigned char)(v * 255.0f + 0.5f); };
        for (int i = 0; i < mesh.triangleCount; ++i)
            for (int k = 0; k < 3; ++k) {
                f.write((const char*)(mesh.vertices + (size_t)i * 9 + k * 3), 12);
                const unsigned char rgb[3] = {to8(color[(size_t)i * 9 + k * 3]), to8(color[(size_t)i * 9 + k * 3 + 1]),
                                              to8(color[(size_t)i * 9 + k * 3 + 2])};
                f.write((const char*)rgb, 3);
            }
        for (int i = 0; i < mesh.triangleCount; ++i) {
            const unsigned char three = 3;
            const int idx[3] = {3 * i, 3 * i + 1, 3 * i + 2};
            f.write((const char*)&three, 1);
            f.write((const char*)idx, 12);
        }
    }
    for (RayTracer* rt : extra) {
        // every instance holds the same maps after the reduction: check it, then drop the helpers
        std::vector<float> other(mesh.triangleCount);
        rt->ReadDosage(other.data(), 0, mesh.triangleCount);
        if (memcmp(other.data(), dose.data(), dose.size() * 4) != 0) { fprintf(stderr, "rank doses differ\n"); return 1; }
        delete rt;
    }
    if (!saveRoute.empty()) {
        char name[32];
        strncpy(name, saveRoute.c_str(), 31);
        name[31] = 0;
        rayTracer.SaveRoute(name);                           // myapp.cpp:298
    }
    return 0;
}
