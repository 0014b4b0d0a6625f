// raytracer.cpp -- host orchestration of the UV-dose hot path on the HIP C ABI
// (reference: raytracer.cpp:3-300; launch sequence in SURVEY.md 3.2).
//
// Same call order as the reference: per lamp generate -> extend -> accumulate on one in-order
// stream with no host synchronisation in between; Shade = computeDosage + dosageToColor;
// errors of the device layer are fatal (the reference's CHECKCL -> FatalError, exit).
#include "raytracer.h"

#include "../../include/uvrt.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace Tmpl8;

namespace {

// template.cpp:904-917 FatalError: report and terminate
void check(int rc, const char* what)
{
    if (rc == UVRT_OK) return;
    fprintf(stderr, "Fatal error in %s: %s\n", what, uvrt_last_error());
    exit(1);
}
void fatal(const char* what)
{
    fprintf(stderr, "Fatal error: %s\n", what);
    exit(1);
}

// ------------------------------------------------------------ minimal XML for route files
struct XmlElem {
    std::string name, text;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<XmlElem> kids;
    const XmlElem* child(const std::string& n) const
    {
        for (auto& k : kids) if (k.name == n) return &k;
        return nullptr;
    }
    const std::string* attr(const std::string& n) const
    {
        for (auto& a : attrs) if (a.first == n) return &a.second;
        return nullptr;
    }
};

struct XmlParser {
    const std::string& s;
    size_t i = 0;
    bool ok = true;
    explicit XmlParser(const std::string& str) : s(str) {}
    void ws() { while (i < s.size() && isspace((unsigned char)s[i])) ++i; }
    std::string ident()
    {
        size_t b = i;
        while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
        return s.substr(b, i - b);
    }
    void skip_misc()
    {
        for (;;) {
            ws();
            if (s.compare(i, 4, "<!--") == 0) { size_t e = s.find("-->", i); i = e == std::string::npos ? s.size() : e + 3; }
            else if (s.compare(i, 2, "<?") == 0) { size_t e = s.find("?>", i); i = e == std::string::npos ? s.size() : e + 2; }
            else break;
        }
    }
    bool element(XmlElem& out)
    {
        skip_misc();
        if (i >= s.size() || s[i] != '<') return ok = false;
        ++i;
        out.name = ident();
        for (;;) {
            ws();
            if (i >= s.size()) return ok = false;
            if (s[i] == '/') { i += 2; return true; }            // <name ... />
            if (s[i] == '>') { ++i; break; }
            std::string an = ident();
            ws();
            if (an.empty() || i >= s.size() || s[i] != '=') return ok = false;
            ++i;
            ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) return ok = false;
            const char q = s[i++];
            size_t e = s.find(q, i);
            if (e == std::string::npos) return ok = false;
            out.attrs.emplace_back(an, s.substr(i, e - i));
            i = e + 1;
        }
        for (;;) {                                               // content
            size_t lt = s.find('<', i);
            if (lt == std::string::npos) return ok = false;
            out.text += s.substr(i, lt - i);
            i = lt;
            if (s.compare(i, 2, "</") == 0) {
                size_t e = s.find('>', i);
                if (e == std::string::npos) return ok = false;
                i = e + 1;
                return true;
            }
            if (s.compare(i, 4, "<!--") == 0) { skip_misc(); continue; }
            XmlElem kid;
            if (!element(kid)) return false;
            out.kids.push_back(std::move(kid));
        }
    }
};

std::string trim(const std::string& t)
{
    size_t b = 0, e = t.size();
    while (b < e && isspace((unsigned char)t[b])) ++b;
    while (e > b && isspace((unsigned char)t[e - 1])) --e;
    return t.substr(b, e - b);
}
// tinyxml2 XMLUtil::ToInt / ToFloat: sscanf "%d" / "%f"
bool to_int(const std::string& t, int* v) { return sscanf(t.c_str(), "%d", v) == 1; }
bool to_float(const std::string& t, float* v) { return sscanf(t.c_str(), "%f", v) == 1; }
// tinyxml2 XMLUtil::ToStr(float): "%.8g"
std::string float_str(float v) { char b[64]; snprintf(b, sizeof b, "%.8g", (double)v); return b; }

}  // namespace

RayTracer::~RayTracer()
{
    if (ctx) uvrt_destroy(ctx);
    delete[] dosageMap;
}

void RayTracer::AddLamp()                                    // raytracer.cpp:3-10
{
    LightPos initLightPos;
    initLightPos.position = make_float2(0.0f, 0.0f);
    initLightPos.duration = 1;
    lightPositions.push_back(initLightPos);
    UpdatePhotonsPerLight();
}

void RayTracer::Init(Mesh* m)                                // raytracer.cpp:12-59
{
    mesh = m;
    LoadRoute(defaultRouteFile);
    // A second Init (model reload, userinterface.cpp:239-240) rebuilds the kernels in the
    // reference, which restarts SEED at 0; a fresh context does the same (and does not leak).
    if (ctx) { uvrt_destroy(ctx); ctx = nullptr; }
    check(uvrt_create(deviceId, &ctx), "uvrt_create");
    check(uvrt_set_scene(ctx, mesh->triangles, mesh->triangleCount, mesh->bvh->bvhNode,
                         (int)mesh->bvh->nodesUsed, mesh->bvh->triIdx), "uvrt_set_scene");
}

void RayTracer::UpdatePhotonsPerLight()                      // raytracer.cpp:61-64
{
    // Round down to an even number, as the reference does.  (An empty lamp list divides by
    // zero in the reference; kept as the caller's error.)
    if (lightPositions.empty()) { photonsPerLight = 0; return; }   // the reference divides by zero here
    photonsPerLight = (int)(photonCount / lightPositions.size()) & ~1;
}

void RayTracer::ComputeDosageMap()                           // raytracer.cpp:66-72
{
    for (LightPos& lightPosition : lightPositions)
        ComputeSingleLightDosageMap(lightPosition, photonsPerLight, mesh->triangleCount);
}

void RayTracer::ComputeSingleLightDosageMap(LightPos lightPos, int photonsPerLight, int triangleCount)
{
    // raytracer.cpp:77 -- lamp foot in world space; the y sum is one f32 addition
    const float lp[3] = {lightPos.position.x, mesh->floorHeight + lightHeight, lightPos.position.y};
    const bool mine = shardWorld <= 1 || (launchIndex % shardWorld) == shardRank;
    ++launchIndex;
    if (mine) {
        check(uvrt_generate(ctx, lp, lightLength, 0, photonsPerLight), "generate");   // :78-80
        check(uvrt_extend(ctx, photonsPerLight), "extend");                           // :82
        check(uvrt_accumulate(ctx, lightPos.duration, triangleCount), "accumulate");  // :84-85
    } else {
        // another rank traces this launch; keep generate.cl's program-scope SEED in step
        check(uvrt_advance_seed(ctx, lp, lightLength), "advance_seed");
    }
    photonMapSize += photonsPerLight;                                   // :87
}

void RayTracer::SetRayRange(int rank, int world)
{
    // contiguous global-id ranges; the union over the ranks is [0, photonsPerLight)
    const long long n = photonsPerLight, share = (n + world - 1) / world;
    rangeFirst = std::min((long long)rank * share, n);
    rangeCount = std::min(share, n - rangeFirst);
}

void RayTracer::ComputeIterationsBatched(int iterations)
{
    std::vector<RayTracer*> self{this};
    ComputeIterationsBatched(self, iterations);
}

void RayTracer::ComputeIterationsBatched(const std::vector<RayTracer*>& group, int iterations)
{
    RayTracer* r0 = group[0];
    const int L = (int)r0->lightPositions.size();
    const long long total = (long long)iterations * L;
    const int kMax = 64;                                  // launches per uvrt_trace_batch
    std::vector<float> lamps((size_t)kMax * 3);
    std::vector<uvrt_replay_op> ops((size_t)kMax);
    std::vector<uvrt_ctx*> ctxs;
    for (RayTracer* rt : group) ctxs.push_back(rt->ctx);
    long long done = 0;
    while (done < total) {
        int cnt = (int)std::min<long long>(kMax, total - done);
        if (cnt < total - done && cnt >= L) cnt -= (int)((done + cnt) % L);    // end on an iteration where one fits
        for (RayTracer* rt : group) {
            for (int j = 0; j < cnt; ++j) {
                const int li = (int)((done + j) % L);
                const LightPos& lp = rt->lightPositions[li];
                lamps[3 * j + 0] = lp.position.x;                                // raytracer.cpp:77
                lamps[3 * j + 1] = rt->mesh->floorHeight + rt->lightHeight;
                lamps[3 * j + 2] = lp.position.y;
                uvrt_replay_op& op = ops[j];
                op.duration = lp.duration;                                       // :84
                rt->photonMapSize += rt->photonsPerLight;                         // :87
                ++rt->launchIndex;
                op.shade = li == L - 1;                                          // myapp.cpp:160
                if (rt->viewMode == maxpower) {                                  // raytracer.cpp:96-104
                    op.which_map = UVRT_MAP_MAX;
                    op.photons_per_light = rt->photonsPerLight;
                    op.scaled_power = rt->lightIntensity * 100;
                    op.min_value = rt->minPower;
                } else {                                                         // :106-116
                    op.which_map = UVRT_MAP_SUM;
                    op.photons_per_light = rt->photonMapSize / (int)rt->lightPositions.size();
                    op.scaled_power = rt->lightIntensity * 0.1f;
                    op.min_value = rt->minDosage;
                }
                op.threshold_view = rt->thresholdView;
                if (op.shade) {                                                  // myapp.cpp:162-163
                    ++rt->currIterations;
                    rt->progress = 100.0f * (float)rt->currIterations / (float)rt->maxIterations;
                }
            }
            const long long n = rt->rangeCount < 0 ? rt->photonsPerLight : rt->rangeCount;
            if (n > 0)
                check(uvrt_trace_batch(rt->ctx, lamps.data(), rt->lightLength, cnt, rt->rangeFirst, n), "trace_batch");
            else
                fatal("ComputeIterationsBatched: an empty ray range (more ranks than photons)");
        }
        if (group.size() > 1) check(uvrt_reduce_batch_group(ctxs.data(), (int)ctxs.size()), "reduce_batch_group");
        for (RayTracer* rt : group) {
            if (group.size() == 1 && rt->reduceOverComm) check(uvrt_reduce_batch(rt->ctx), "reduce_batch");
            check(uvrt_replay_batch(rt->ctx, ops.data(), cnt, rt->mesh->triangleCount), "replay_batch");
        }
        done += cnt;
    }
}

void RayTracer::Shade()                                      // raytracer.cpp:93-120
{
    if (viewMode == maxpower) {
        // only the photons of one iteration; x100: W/m^2 -> microW/cm^2
        check(uvrt_shade(ctx, UVRT_MAP_MAX, photonsPerLight, lightIntensity * 100, minPower, thresholdView,
                         mesh->triangleCount), "computeDosage + dosageToColor");
    } else {
        // x0.1: J/m^2 -> mJ/cm^2
        check(uvrt_shade(ctx, UVRT_MAP_SUM, photonMapSize / (int)lightPositions.size(), lightIntensity * 0.1f,
                         minDosage, thresholdView, mesh->triangleCount), "computeDosage + dosageToColor");
    }
}

void RayTracer::ResetDosageMap()                             // raytracer.cpp:122-131
{
    startedComputation = true;
    compTime = 0;
    timerClock.reset();
    if (autoSaveRoute) SaveRoute(defaultRouteFile);
    progress = 0;
    finishedComputation = false;
    currIterations = 0;
    launchIndex = 0;
    ClearBuffers(true);
}

void RayTracer::ClearBuffers(bool resetColor)                // raytracer.cpp:133-143
{
    photonMapSize = 0;
    check(uvrt_resize_rays(ctx, photonCount), "resize_rays");
    check(uvrt_reset(ctx, resetColor), "reset");
}

void RayTracer::CalibratePower(float measurePower, float measureHeight, float measureDist)   // :151-227
{
    measureHeight += mesh->floorHeight;

    // A small square as the sample geometry, 0.2 m wide, facing the lamp
    LightPos singleLightPos;
    singleLightPos.position = make_float2(0.0f, 0.0f);
    singleLightPos.duration = 0.0f;   // uninitialised in the reference; only the max map is read
    Tri square[2];
    memset((void*)square, 0, sizeof square);
    const float triWidth = 0.1f;
    const float x = singleLightPos.position.x, z = singleLightPos.position.y + measureDist;
    square[0].vertex0 = make_float3_strict(x + triWidth, measureHeight + triWidth, z);
    square[0].vertex1 = make_float3_strict(x - triWidth, measureHeight + triWidth, z);
    square[0].vertex2 = make_float3_strict(x + triWidth, measureHeight - triWidth, z);
    square[1].vertex0 = make_float3_strict(x - triWidth, measureHeight - triWidth, z);
    square[1].vertex1 = make_float3_strict(x - triWidth, measureHeight + triWidth, z);
    square[1].vertex2 = make_float3_strict(x + triWidth, measureHeight - triWidth, z);
    // single-node BVH whose root is a leaf (raytracer.cpp:173-187); its bounds are never tested
    BVHNode hostNode;
    memset(&hostNode, 0, sizeof hostNode);
    hostNode.leftFirst = 0;
    hostNode.triCount = 2;
    uint hostTriIdx[2] = {0, 1};
    check(uvrt_set_scene(ctx, square, 2, &hostNode, 1, hostTriIdx), "set_scene(calibration)");

    const int keepRank = shardRank, keepWorld = shardWorld;
    const long long keepIndex = launchIndex;
    shardRank = 0;
    shardWorld = 1;   // calibration is replicated: every rank traces every launch
    ClearBuffers(false);
    for (int i = 0; i < maxIterations; ++i)
        ComputeSingleLightDosageMap(singleLightPos, photonCount, 2);
    shardRank = keepRank;
    shardWorld = keepWorld;
    launchIndex = keepIndex;

    // power 1, so measured / traced irradiance is the calibrated power
    check(uvrt_compute_dosage(ctx, UVRT_MAP_MAX, photonCount, 1.0f, 2), "computeDosage");
    check(uvrt_sync(ctx), "sync");
    check(uvrt_read_dosage(ctx, dosageMap, 0, 2), "read_dosage");
    const float avgPower = (dosageMap[0] + dosageMap[1]) / 2.0f;
    calibratedPower = 0.01f * (measurePower / avgPower);
    lightIntensity = calibratedPower;

    // restore the room (raytracer.cpp:212-224)
    check(uvrt_set_scene(ctx, mesh->triangles, mesh->triangleCount, mesh->bvh->bvhNode,
                         (int)mesh->bvh->nodesUsed, mesh->bvh->triIdx), "set_scene(restore)");
    std::cout << "Done calibrating " << std::endl;
}

void RayTracer::ReadDosage(float* out, int first, int count)
{
    check(uvrt_read_dosage(ctx, out, first, count), "read_dosage");
}

void RayTracer::Sync() { check(uvrt_sync(ctx), "sync"); }

void RayTracer::SaveRoute(char fileName[32])                 // raytracer.cpp:233-259
{
    // Same document tinyxml2 prints: 4-space indent, floats as "%.8g".
    std::ostringstream o;
    o << "<route>\n";
    o << "    <aantal_fotonen>" << photonCount << "</aantal_fotonen>\n";
    o << "    <aantal_iteraties>" << maxIterations << "</aantal_iteraties>\n";
    o << "    <lamp_sterkte>" << float_str(lightIntensity) << "</lamp_sterkte>\n";
    o << "    <minimale_dosis>" << float_str(minDosage) << "</minimale_dosis>\n";
    o << "    <minimale_bestralingssterkte>" << float_str(minPower) << "</minimale_bestralingssterkte>\n";
    o << "    <lamp_lengte>" << float_str(lightLength) << "</lamp_lengte>\n";
    o << "    <lamp_hoogte>" << float_str(lightHeight) << "</lamp_hoogte>\n";
    if (lightPositions.empty()) o << "    <route/>\n";
    else {
        o << "    <route>\n";
        for (size_t i = 0; i < lightPositions.size(); i++) {
            const LightPos& lp = lightPositions[i];
            o << "        <lamp_positie_" << i << " positie_x=\"" << float_str(lp.position.x) << "\" positie_y=\""
              << float_str(lp.position.y) << "\" duration=\"" << float_str(lp.duration) << "\"/>\n";
        }
        o << "    </route>\n";
    }
    o << "</route>\n";
    std::ofstream f(routeDir + fileName + ".xml", std::ios::binary);
    if (f) f << o.str();   // a failed save is silent in the reference too (return value dropped)
}

void RayTracer::LoadRoute(char fileName[32])                 // raytracer.cpp:261-300
{
    std::ifstream f(routeDir + fileName + ".xml", std::ios::binary);
    if (!f) return;                                          // :266
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    XmlParser xp(text);
    XmlElem root;
    if (!xp.element(root)) return;
    const XmlElem* e;
    if ((e = root.child("aantal_fotonen"))) to_int(trim(e->text), &photonCount);
    if ((e = root.child("aantal_iteraties"))) to_int(trim(e->text), &maxIterations);
    if ((e = root.child("lamp_sterkte"))) to_float(trim(e->text), &lightIntensity);
    if ((e = root.child("minimale_dosis"))) to_float(trim(e->text), &minDosage);
    if ((e = root.child("minimale_bestralingssterkte"))) to_float(trim(e->text), &minPower);
    if ((e = root.child("lamp_lengte"))) to_float(trim(e->text), &lightLength);
    if ((e = root.child("lamp_hoogte"))) to_float(trim(e->text), &lightHeight);
    if ((e = root.child("route"))) {
        lightPositions.clear();
        for (int i = 0;; i++) {
            const XmlElem* lampElem = e->child("lamp_positie_" + std::to_string(i));
            if (!lampElem) break;
            LightPos lp;
            lp.position = make_float2(0.0f, 0.0f);
            lp.duration = 0.0f;
            const std::string* a;
            if ((a = lampElem->attr("positie_x"))) to_float(*a, &lp.position.x);
            if ((a = lampElem->attr("positie_y"))) to_float(*a, &lp.position.y);
            if ((a = lampElem->attr("duration"))) to_float(*a, &lp.duration);
            lightPositions.push_back(lp);
        }
    }
    UpdatePhotonsPerLight();
}
