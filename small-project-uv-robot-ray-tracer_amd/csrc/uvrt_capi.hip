// uvrt_capi.hip -- context, scene, buffers, knobs, read-backs and test hooks
// (the C ABI of include/uvrt.h over the HIP kernels; the context and its helpers are in uvrt_ctx.h)
#include "uvrt_ctx.h"

using namespace uvrt;
using namespace uvrt_impl;

namespace uvrt_impl {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace uvrt_impl

extern "C" {

const char* uvrt_last_error(void) { return g_err.c_str(); }
const char* uvrt_version(void) { return "uvrt-mi355x 0.1 (gfx950)"; }
int uvrt_clock_probe_start(uvrt_ctx* c, int32_t microseconds)
{
    if (!c || microseconds < 1 || microseconds > 1000000) return fail(UVRT_ERR_INVALID, "uvrt_clock_probe_start: 1 .. 1 000 000 us");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->probe_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->probe_stream, hipStreamNonBlocking, hi));     // a wave slot as soon as one frees
    }
    if (int rc = c->probe_out.ensure(16, true, c->probe_stream)) return rc;
    launch_clock_probe(c->probe_out.as<unsigned long long>(), (unsigned long long)microseconds * 100ull, c->probe_stream);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_clock_probe_read(uvrt_ctx* c, double* shader_mhz)
{
    if (!c || !shader_mhz || !c->probe_stream) return fail(UVRT_ERR_INVALID, "uvrt_clock_probe_read: no probe started");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(v, c->probe_out.p, 16, hipMemcpyDeviceToHost, c->probe_stream));
    HIP_TRY(hipStreamSynchronize(c->probe_stream));
    if (v[1] == 0) return fail(UVRT_ERR_HIP, "uvrt_clock_probe_read: the probe wrote nothing");
    *shader_mhz = (double)v[0] / (double)v[1] * 100.0;          // s_memrealtime ticks at 100 MHz
    return UVRT_OK;
}

int uvrt_device_cus(uvrt_ctx* c) { return c ? c->num_cus : 0; }
int uvrt_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int uvrt_create(int device_id, uvrt_ctx** out)
{
    if (!out) return fail(UVRT_ERR_INVALID, "uvrt_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(UVRT_ERR_NO_DEVICE, "uvrt_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(UVRT_ERR_INVALID, "uvrt_create: device %d out of range [0,%d)", device_id, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(UVRT_ERR_NO_DEVICE, "uvrt_create: device %d is %s, this library is built for gfx950 only",
                    device_id, prop.gcnArchName);
    uvrt_ctx* c = new uvrt_ctx();
    c->device = device_id;
    c->hot.reserve(uvrt_ctx::HOT_MAX);         // entries are referred to by pointer while their set-up is pending
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (c->num_cus > 256) c->num_cus = 256;   // the overflow-stack buffer is sized for 256 CUs x 16 workgroups
    if (const char* e = getenv("UVRT_REPLICAS")) {   // developer knob: deposit replicas of tempPhotonMap
        const int r = atoi(e);
        if (r >= 1 && r <= 64) c->replicas_knob = r;
    }
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) {
        if (l > 0) HIP_TRY(hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_tail[l], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_mapfence, hipEventDisableTiming));
    if (const char* e = getenv("UVRT_LANES")) { const int v = atoi(e); if (v >= 1 && v <= uvrt_ctx::MAXL) c->nlanes = v; }
    if (const char* e = getenv("UVRT_PIPELINE")) c->pipeline = atoi(e) != 0;   // developer knob
    if (const char* e = getenv("UVRT_BATCH_CHUNK_MB")) { const long v = atol(e); if (v > 0) c->batch_chunk_bytes = (size_t)v << 20; }
#ifdef UVRT_DEV_VARIANTS
    if (const char* e = getenv("UVRT_PROBE_SKIP_GENERATE")) c->probe_skip_generate = atoi(e);
#endif
    if (const char* e = getenv("UVRT_DRAIN_MERGE")) c->drain_merge = atoi(e) != 0;      // developer knob
    if (const char* e = getenv("UVRT_BATCH_LANES")) { const int v = atoi(e); if (v >= 1 && v <= uvrt_ctx::MAXL - 1) c->batch_lanes = v; }
    if (const char* e = getenv("UVRT_COMM_RESERVE_CUS")) { const int v = atoi(e); if (v >= 0 && v <= 64 && v % 8 == 0) c->comm_reserve_knob = v; }
    if (const char* e = getenv("UVRT_HOT_SAMPLE")) { const int v = atoi(e); if (v >= 256 && v <= (1 << 20)) c->hot_sample = v; }
    if (const char* e = getenv("UVRT_HOT_DIRECT")) { const int v = atoi(e); if (v >= 0 && v <= 8192) c->hot_direct = v; }
    if (const char* e = getenv("UVRT_HOT_TAIL")) { const int v = atoi(e); if (v >= 0 && v < 64) c->hot_tail = v; }
    int rc = c->error_flag.ensure(256, true, c->stream);      // (a developer build keeps trip statistics behind the flag)
#ifndef UVRT_TRIP_STATS
    // the stack-overflow flag lives in pinned host memory the kernels can write: uvrt_sync reads it after the stream sync
    // instead of copying a device word back (a pageable 4-byte copy cost every sync ~10 us)
    if (!rc) {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            delete c;
            return fail(UVRT_ERR_HIP, "uvrt_create: cannot allocate the pinned error flag");
        }
        memset(hp, 0, 64);
        c->host_flag = (uint32_t*)hp;
        c->host_flag_dev = (uint32_t*)dp;
    }
#endif
    // 256 CUs x 16 workgroups x 256 threads x 16 entries: the largest persistent grid
    if (!rc) rc = c->ovf_stack.ensure((size_t)OVF_MAX_ENTRIES * sizeof(uint32_t), false, c->stream);
    if (rc) { delete c; return rc; }
    *out = c;
    return UVRT_OK;
}

void uvrt_destroy(uvrt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) uvrt_comm_destroy(c);       // first: it restores the lanes' plain streams
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) {
        if (c->side[l]) (void)hipStreamSynchronize(c->side[l]);
        for (DevBuf* b : {&c->xrays[l], &c->xrecs[l], &c->xcounts[l], &c->xovf[l]}) b->release();
        if (c->ev_tail[l]) (void)hipEventDestroy(c->ev_tail[l]);
        if (c->side[l]) (void)hipStreamDestroy(c->side[l]);
    }
    if (c->ev_fence) (void)hipEventDestroy(c->ev_fence);
    if (c->ev_mapfence) (void)hipEventDestroy(c->ev_mapfence);
    c->quads.release();
    for (DevBuf& b : c->recs4) b.release();
    for (DevBuf& b : c->b_recs) b.release();
    (void)hot_reset(c, false);
    for (auto& bset : c->bs) {
        for (DevBuf* b : {&bset.rays, &bset.planes, &bset.folded}) b->release();
        if (bset.free_ev) (void)hipEventDestroy(bset.free_ev);
    }
    for (DevBuf* b : {&c->pairs, &c->recs, &c->perm, &c->ltris, &c->leaf_count, &c->area, &c->photon_map, &c->max_map,
                      &c->counts, &c->dosage, &c->color, &c->rays, &c->keyrank, &c->sorted,
                      &c->order, &c->hits, &c->hist, &c->bin_start, &c->export_buf,
                      &c->ovf_stack, &c->error_flag})
        b->release();
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (c->host_flag) (void)hipHostFree(c->host_flag);
    if (c->probe_stream) { (void)hipStreamSynchronize(c->probe_stream); (void)hipStreamDestroy(c->probe_stream); }
    c->probe_out.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int uvrt_set_stream(uvrt_ctx* c, void* hip_stream)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    c->lane = 0;
    return UVRT_OK;
}

int uvrt_set_scene(uvrt_ctx* c, const void* tris64, int32_t T, const void* nodes32,
                   int32_t node_count, const uint32_t* tri_idx)
{
    if (!c || !tris64 || !nodes32 || !tri_idx) return fail(UVRT_ERR_INVALID, "uvrt_set_scene: null argument");
    if (T <= 0 || T > MAX_TRIS || node_count <= 0)
        return fail(UVRT_ERR_INVALID, "uvrt_set_scene: tri_count %d / node_count %d out of range", T, node_count);
    if (int rc = set_device(c)) return rc;

    struct HostNode { float mn[3]; int32_t leftFirst; float mx[3]; int32_t triCount; };
    static_assert(sizeof(HostNode) == 32, "BVHNode is 32 bytes (bvh.h:11-21)");
    const HostNode* nodes = (const HostNode*)nodes32;

    for (int32_t i = 0; i < T; ++i)
        if (tri_idx[i] >= (uint32_t)T)
            return fail(UVRT_ERR_BVH, "uvrt_set_scene: triIdx[%d] = %u >= tri_count %d", i, tri_idx[i], T);
    // vertex coordinates beyond 1e9 (or non-finite): the shortcuts' range proofs do not cover them
    bool huge_vertex = false;
    {
        const float* tv = (const float*)tris64;
        for (int64_t i = 0; i < (int64_t)T && !huge_vertex; ++i)
            for (int k = 0; k < 12; ++k)
                if ((k & 3) != 3 && !(std::fabs(tv[i * 16 + k]) <= 1e9f)) { huge_vertex = true; break; }
    }

    // Re-layout: walk the tree breadth-first from node 0; every inner node becomes one pair
    // record (numbered in visit order, so the top of the tree is contiguous at the front).
    auto leaf_ref = [&](const HostNode& n, std::vector<uint32_t>& leaf_count, int& err) -> uint32_t {
        int64_t first = n.leftFirst, cnt = n.triCount;
        if (first < 0 || first + cnt > T) { err = 1; return REF_DONE; }
        leaf_count[first] = (uint32_t)cnt;
        uint32_t code = cnt >= 15 ? 15u : (uint32_t)cnt;
        return REF_LEAF_BIT | (code << REF_COUNT_SHIFT) | (uint32_t)first;
    };
    std::vector<uint32_t> leaf_count(T, 0u);
    std::vector<PairRec> pairs;
    std::vector<int32_t> queue;   // inner nodes in BFS order; index in queue == pair index
    std::vector<int32_t> depth;   // tree depth of queue[i]
    uint32_t top_pairs = 0;
    int err = 0;
    bool tiny_bound = false;
    uint32_t root_ref;
    if (nodes[0].triCount > 0) {
        root_ref = leaf_ref(nodes[0], leaf_count, err);
    } else {
        root_ref = 0;
        queue.push_back(0);
        depth.push_back(0);
    }
    for (size_t qi = 0; qi < queue.size() && !err; ++qi) {
        const HostNode& n = nodes[queue[qi]];
        const int64_t l = n.leftFirst;
        if (l < 0 || l + 1 >= node_count) { err = 2; break; }
        if (queue.size() > (size_t)node_count) { err = 3; break; }   // cycle
        uint32_t ref[2];
        for (int k = 0; k < 2; ++k) {
            const HostNode& ch = nodes[l + k];
            if (ch.triCount > 0) ref[k] = leaf_ref(ch, leaf_count, err);
            else { ref[k] = (uint32_t)queue.size(); queue.push_back((int32_t)(l + k)); depth.push_back(depth[qi] + 1); }
        }
        const HostNode& a = nodes[l];
        const HostNode& b = nodes[l + 1];
        for (const HostNode* hn : {&a, &b})
            for (int k = 0; k < 3; ++k)
                for (float v : {hn->mn[k], hn->mx[k]})
                    if ((v != 0.0f && std::fabs(v) < 8.6736174e-19f) || !(std::fabs(v) <= 1e9f))
                        tiny_bound = true;   // below 2^-60, above 1e9, inf or NaN
        PairRec pr;
        pr.c0min_ref0 = make_float4(a.mn[0], a.mn[1], a.mn[2], 0.f);
        pr.c0max_ref1 = make_float4(a.mx[0], a.mx[1], a.mx[2], 0.f);
        memcpy(&pr.c0min_ref0.w, &ref[0], 4);
        memcpy(&pr.c0max_ref1.w, &ref[1], 4);
        pr.c1min = make_float4(b.mn[0], b.mn[1], b.mn[2], 0.f);
        pr.c1max = make_float4(b.mx[0], b.mx[1], b.mx[2], 0.f);
        pairs.push_back(pr);
        if (depth[qi] < 8 && top_pairs < TOP6_MAX) top_pairs = (uint32_t)qi + 1;   // level order: a prefix
    }
    if (err) return fail(UVRT_ERR_BVH, "uvrt_set_scene: malformed BVH (code %d)", err);
    // The 4-wide collapse (one level): a node takes the children of its inner children.  Numbered breadth-first
    // like the pairs; a child reference is a 64-byte unit index (2 x node index) or the BVH2 leaf reference.
    std::vector<QuadRec> quads;
    uint32_t top_quads = 0;
    if (!pairs.empty()) {
        std::vector<int32_t> qpair{0};        // pair index (of the BVH2 inner node) of every 4-wide node
        std::vector<int32_t> qdepth{0};
        const float far_box = 1e30f;          // an empty slot: a box no ray reaches (entry distance >= 1e30)
        for (size_t qi = 0; qi < qpair.size(); ++qi) {
            struct Child { float mn[3], mx[3]; uint32_t ref; };
            Child ch[4];
            int nch = 0;
            auto add = [&](const float4& mn, const float4& mx, uint32_t ref) {
                Child& c4 = ch[nch++];
                c4.mn[0] = mn.x; c4.mn[1] = mn.y; c4.mn[2] = mn.z;
                c4.mx[0] = mx.x; c4.mx[1] = mx.y; c4.mx[2] = mx.z;
                c4.ref = ref;
            };
            auto children_of = [&](const PairRec& pr, float4 mn[2], float4 mx[2], uint32_t ref[2]) {
                mn[0] = pr.c0min_ref0; mx[0] = pr.c0max_ref1; mn[1] = pr.c1min; mx[1] = pr.c1max;
                memcpy(&ref[0], &pr.c0min_ref0.w, 4);
                memcpy(&ref[1], &pr.c0max_ref1.w, 4);
            };
            float4 mn[2], mx[2];
            uint32_t ref[2];
            children_of(pairs[qpair[qi]], mn, mx, ref);
            for (int k = 0; k < 2; ++k) {
                if (ref[k] >= REF_LEAF_BIT) { add(mn[k], mx[k], ref[k]); continue; }
                float4 gmn[2], gmx[2];
                uint32_t gref[2];
                children_of(pairs[ref[k]], gmn, gmx, gref);
                for (int j = 0; j < 2; ++j) add(gmn[j], gmx[j], gref[j]);
            }
            QuadRec q;
            memset(&q, 0, sizeof q);
            float y[4][2];
            for (int k = 0; k < 4; ++k) {
                if (k < nch) {
                    uint32_t r = ch[k].ref;
                    if (r < REF_LEAF_BIT) {           // inner: it becomes a 4-wide node of its own
                        qpair.push_back((int32_t)r);
                        qdepth.push_back(qdepth[qi] + 1);
                        r = 2u * (uint32_t)(qpair.size() - 1);
                    }
                    q.xz[k] = make_float4(ch[k].mn[0], ch[k].mx[0], ch[k].mn[2], ch[k].mx[2]);
                    y[k][0] = ch[k].mn[1]; y[k][1] = ch[k].mx[1];
                    q.ref[k] = r;
                } else {
                    q.xz[k] = make_float4(far_box, far_box, far_box, far_box);
                    y[k][0] = y[k][1] = far_box;
                    q.ref[k] = REF_DONE;
                }
            }
            q.y01 = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]);
            q.y23 = make_float4(y[2][0], y[2][1], y[3][0], y[3][1]);
            quads.push_back(q);
            if (qdepth[qi] < 4 && top_quads < 64) top_quads = (uint32_t)qi + 1;     // levels 0-3: up to 85 nodes, 64 cached
        }
    }
    // a child reference is a record index with 32-bit byte offsets: inner-node records + leaf records < 2^26
    if (pairs.size() + (size_t)T >= ((size_t)1 << 26))
        return fail(UVRT_ERR_INVALID, "uvrt_set_scene: %zu inner nodes + %d triangles exceed 2^26 records", pairs.size(), T);

    if (int rcj = join_all(c)) return rcj;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->lane = 0;
    const bool resized = (T != c->T);
    int rc;
    if ((rc = c->pairs.ensure(std::max<size_t>(pairs.size(), 1) * sizeof(PairRec), false, c->stream))) return rc;
    if ((rc = c->recs.ensure((pairs.size() + (size_t)T + 1) * 64, true, c->stream))) return rc;
    for (int l = 1; l < c->nlanes; ++l)
        if ((rc = c->xrecs[l].ensure((pairs.size() + (size_t)T + 1) * 64, true, c->stream))) return rc;
    // + 16 bytes: the merged record fetch of the traversal reads 64 bytes at every leaf record
    if ((rc = c->ltris.ensure((size_t)T * sizeof(LeafTri) + 16, true, c->stream))) return rc;
    if ((rc = c->leaf_count.ensure((size_t)T * 4, false, c->stream))) return rc;
    if ((rc = c->area.ensure((size_t)T * 4, false, c->stream))) return rc;
    if (resized || !c->photon_map.p) {
        // raytracer.cpp:32-37 (the reference leaves them uninitialised until reset; zero here)
        // The colour buffer stands for the GL vertex buffer of the caller's mesh (raytracer.cpp:37): a scene swap
        // does not touch it.  It only grows, so CalibratePower's detour over a 2-triangle scene
        // (raytracer.cpp:166-224, ClearBuffers(false)) leaves the room's colours as they were.
        for (DevBuf* b : {&c->photon_map, &c->max_map, &c->counts, &c->xcounts[1], &c->xcounts[2], &c->xcounts[3],
                          &c->dosage}) b->release();
        if ((rc = c->photon_map.ensure((size_t)T * 8, true, c->stream))) return rc;
        if ((rc = c->max_map.ensure((size_t)T * 8, true, c->stream))) return rc;
        // 16 deposit replicas (two per XCD: a workgroup deposits into replica blockIdx % 16), at most 64 MiB in
        // total.  Fewer than 8 serialise on the hot triangles' counters; more than ~24 push the planes out of L2 and
        // every wave then waits for its deposits' memory round trips (profiles/r02/r02_experiments.txt: 64 replicas cost
        // the batched step 5 %)
        int R = c->replicas_knob > 0 ? c->replicas_knob : 16;
        while (R > 1 && (size_t)R * (size_t)T * 4 > ((size_t)64 << 20)) R >>= 1;
        c->replicas = R;
        if ((rc = c->counts.ensure((size_t)R * (size_t)T * 4, true, c->stream))) return rc;
        for (int l = 1; l < c->nlanes; ++l)
            if ((rc = c->xcounts[l].ensure((size_t)R * (size_t)T * 4, true, c->stream))) return rc;
        if ((rc = c->dosage.ensure((size_t)T * 4, true, c->stream))) return rc;
        if ((rc = c->color.ensure((size_t)T * 36, true, c->stream))) return rc;
    }
    if (!pairs.empty())
        HIP_TRY(hipMemcpyAsync(c->pairs.p, pairs.data(), pairs.size() * sizeof(PairRec), hipMemcpyHostToDevice, c->stream));
    if ((rc = c->quads.ensure(std::max<size_t>(quads.size(), 1) * sizeof(QuadRec), false, c->stream))) return rc;
    if (!quads.empty())
        HIP_TRY(hipMemcpyAsync(c->quads.p, quads.data(), quads.size() * sizeof(QuadRec), hipMemcpyHostToDevice, c->stream));
    for (DevBuf& b : c->recs4) b.release();              // sized per scene; rebuilt on demand (uvrt_set_wide_bvh)
    HIP_TRY(hipMemcpyAsync(c->leaf_count.p, leaf_count.data(), (size_t)T * 4, hipMemcpyHostToDevice, c->stream));
    // staging copies of the reference-layout arrays for the device-side preparation kernel
    DevBuf d_tris, d_idx;
    if ((rc = d_tris.ensure((size_t)T * 64, false, c->stream))) return rc;
    if ((rc = d_idx.ensure((size_t)T * 4, false, c->stream))) { d_tris.release(); return rc; }
    hipError_t e1 = hipMemcpyAsync(d_tris.p, tris64, (size_t)T * 64, hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(d_idx.p, tri_idx, (size_t)T * 4, hipMemcpyHostToDevice, c->stream);
    if (e1 == hipSuccess && e2 == hipSuccess) {
        launch_prepare_scene(d_tris.as<float4>(), d_idx.as<uint32_t>(), c->ltris.as<LeafTri>(),
                             c->area.as<float>(), T, c->stream);
        launch_prepare_leaves6(c->ltris.as<LeafTri>(), c->recs.p, (int32_t)pairs.size(), T, c->stream);
        for (int l = 1; l < c->nlanes; ++l)
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), c->xrecs[l].p, (int32_t)pairs.size(), T, c->stream);
        e1 = hipGetLastError();
        e2 = hipStreamSynchronize(c->stream);
    }
    d_tris.release();
    d_idx.release();
    if (e1 != hipSuccess) return fail(UVRT_ERR_HIP, "uvrt_set_scene: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return fail(UVRT_ERR_HIP, "uvrt_set_scene: %s", hipGetErrorString(e2));
    c->T = T;
    c->root_ref = root_ref;
    c->top_pairs = top_pairs;
    c->npairs = (int32_t)pairs.size();
    c->nquads = (int32_t)quads.size();
    c->top_quads = top_quads;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->recs4_valid[l] = false;
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->xrecs_valid[l] = false;
    c->have_perm = false;
    c->have_scene = true;
    c->scene_force_exact = tiny_bound || huge_vertex;
    if ((rc = hot_reset(c, true))) return rc;         // statistics of the previous scene; the new scene's first slab
    HIP_TRY(hipStreamSynchronize(c->stream));
    // a batch of the previous scene is void; its buffers are sized per scene
    c->b_count = 0;
    c->b_is_folded = false;
    for (DevBuf& b : c->b_recs) b.release();
    c->b_recs.clear();
    c->b_recs_key.clear();
    for (auto& bset : c->bs) for (DevBuf* b : {&bset.planes, &bset.folded}) b->release();
    return UVRT_OK;
}

int uvrt_resize_rays(uvrt_ctx* c, int64_t photon_count)
{
    if (!c || photon_count < 0) return fail(UVRT_ERR_INVALID, "uvrt_resize_rays: bad argument");
    if (int rc = set_device(c)) return rc;
    if (photon_count == c->capacity && c->rays.p && (!c->record_hits || c->hits.p)) { c->last_n = -1; return UVRT_OK; }
    if (int rcj = join_all(c)) return rcj;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->lane = 0;
    int rc;
    const size_t n = (size_t)photon_count;
    if ((rc = c->rays.ensure(n * 16, false, c->stream))) return rc;
    for (int l = 1; l < c->nlanes; ++l)
        if ((rc = c->xrays[l].ensure(n * 16, false, c->stream))) return rc;
    if ((rc = c->keyrank.ensure(n * 8, false, c->stream))) return rc;
    if ((rc = c->sorted.ensure(n * 16, false, c->stream))) return rc;
    if ((rc = c->order.ensure(n * 4, false, c->stream))) return rc;
    if (c->record_hits && (rc = c->hits.ensure(n * 8, false, c->stream))) return rc;
    c->capacity = photon_count;
    c->last_n = -1;
    return UVRT_OK;
}

int uvrt_reset(uvrt_ctx* c, int32_t reset_color)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_reset: no scene");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    launch_reset(c->photon_map.as<double>(), c->max_map.as<double>(), c->counts.as<int32_t>(),
                 c->replicas, c->T, c->color.as<float>(), reset_color, c->T, c->stream);
    HIP_TRY(hipGetLastError());
    // side-lane count buffers are zero unless an extend was never accumulated; only then must later
    // generate / extend work on the side streams wait for this reset
    bool dirty = false;
    for (int l = 1; l < uvrt_ctx::MAXL; ++l) {
        if (c->counts_dirty[l] && c->xcounts[l].p) {
            HIP_TRY(hipMemsetAsync(c->xcounts[l].p, 0, c->xcounts[l].bytes, c->stream));
            dirty = true;
        }
        c->counts_dirty[l] = false;
    }
    c->counts_dirty[0] = false;
    if (c->b_count > 0) {     // a traced batch that was never replayed: drop its deposits
        if (c->b_is_folded) HIP_TRY(hipMemsetAsync(c->bs[c->b_set].folded.p, 0, c->bs[c->b_set].folded.bytes, c->stream));
        else HIP_TRY(hipMemsetAsync(c->bs[c->b_set].planes.p, 0, c->bs[c->b_set].planes.bytes, c->stream));
        HIP_TRY(hipEventRecord(c->bs[c->b_set].free_ev, c->stream));
        c->b_count = 0;
        c->b_is_folded = false;
    }
    return dirty ? mark_fence(c) : mark_map_fence(c);
}

int uvrt_set_record_perm(uvrt_ctx* c, const uint32_t* perm, int32_t n)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: no scene");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->xrecs_valid[l] = false;
    if (!perm) { c->have_perm = false; return UVRT_OK; }
    if (n != c->npairs) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: %d entries, the scene has %d inner nodes", n, c->npairs);
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int32_t i = 0; i < n; ++i) {
        if (perm[i] >= (uint32_t)n || seen[perm[i]]) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: not a permutation");
        seen[perm[i]] = 1;
    }
    if (n == 0) { c->have_perm = false; return UVRT_OK; }
    if (int rc = c->perm.ensure((size_t)n * 4, false, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(c->perm.p, perm, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_perm = true;
    c->perm_gen = ++c->perm_clock;       // same address, another renumbering: records prepared from the old one are stale
    return UVRT_OK;
}

int uvrt_read_record_perm(uvrt_ctx* c, uint32_t* out, int32_t n)
{
    if (!c || !out || !c->have_scene || n != c->npairs)
        return fail(UVRT_ERR_INVALID, "uvrt_read_record_perm: need a scene and n = its %d inner nodes", c ? c->npairs : 0);
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    const uint32_t* pm = c->have_perm ? c->perm.as<uint32_t>() : c->lane_perm[c->lane];
    if (!pm) { for (int32_t i = 0; i < n; ++i) out[i] = (uint32_t)i; return UVRT_OK; }
    HIP_TRY(hipMemcpyAsync(out, pm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

int uvrt_sync(uvrt_ctx* c)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    uint32_t flag = 0;
#ifdef UVRT_TRIP_STATS
    HIP_TRY(hipMemcpyAsync(&flag, c->error_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
#else
    HIP_TRY(hipStreamSynchronize(c->stream));
    flag = *(volatile uint32_t*)c->host_flag;
#endif
#ifdef UVRT_TRIP_STATS
    if (getenv("UVRT_TRIP_STATS")) {
        unsigned long long st[27];
        HIP_TRY(hipMemcpy(st, (char*)c->error_flag.p + 8, sizeof st, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset((char*)c->error_flag.p + 8, 0, sizeof st));
        if (st[1]) {
            const double t = (double)st[1];
            fprintf(stderr, "trip stats: waves %llu trips %llu (%.1f per wave)  lanes per trip: inner %.2f leaf-visit %.2f leaf-wait %.2f "
                    "idle %.2f  cached %.2f  leaf trips %.3f  drain trips %.3f  slow trips %.4f (sp>=9 %.4f >=10 %.4f >=11 %.4f >=12 %.4f)  "
                    "refills/wave %.1f\n",
                    st[0], st[1], t / (double)st[0], st[2] / t, st[3] / t, st[4] / t, st[5] / t, st[10] / t, st[6] / t, st[7] / t,
                    st[8] / t, st[11] / t, st[12] / t, st[13] / t, st[14] / t, (double)st[9] / (double)st[0]);
            const double w = (double)st[0];
            fprintf(stderr, "trip clocks (s_memtime ticks per wave): life %.0f  record fetch %.0f  leaf tests %.0f  box tests + descend %.0f  "
                    "refill %.0f  general step %.0f  rest (loop overhead) %.0f;  per trip: fetch %.0f leaf %.0f box %.0f; per refill %.0f; per general step %.0f\n",
                    st[19] / w, st[15] / w, st[16] / w, st[17] / w, st[18] / w, st[20] / w,
                    (st[19] - st[15] - st[16] - st[17] - st[18] - st[20]) / w,
                    st[15] / t, st[16] / t, st[17] / t, (double)st[18] / (double)(st[9] ? st[9] : 1), (double)st[20] / (double)(st[8] ? st[8] : 1));
            // totals since the last sync, for tests/tools/stream_census.py
            fprintf(stderr, "trip census: waves %llu stream_in %llu stream_leaf %llu stream_both %llu general_exact %llu general_other %llu "
                    "refills %llu leaf_lane_tests %llu trips %llu\n", st[0], st[21], st[22], st[23], st[24], st[25], st[9], st[26], st[1]);
        }
    }
#endif
    if (flag) {
#ifdef UVRT_TRIP_STATS
        HIP_TRY(hipMemsetAsync(c->error_flag.p, 0, 4, c->stream));
#else
        *(volatile uint32_t*)c->host_flag = 0u;
#endif
        return fail(UVRT_ERR_STACK, "extend: BVH traversal needed more than 32 stack entries (extend.cl:43)");
    }
    return UVRT_OK;
}

static int read_back(uvrt_ctx* c, const DevBuf& b, size_t elem, void* out, int64_t first, int64_t count,
                     int64_t limit, const char* what)
{
    if (!c || !out || first < 0 || count < 0 || first + count > limit)
        return fail(UVRT_ERR_INVALID, "%s: range [%lld,+%lld) outside [0,%lld)", what, (long long)first,
                    (long long)count, (long long)limit);
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    if (count == 0) return UVRT_OK;
    HIP_TRY(hipMemcpyAsync(out, (const char*)b.p + (size_t)first * elem, (size_t)count * elem,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

int uvrt_read_dosage(uvrt_ctx* c, float* out, int32_t first, int32_t count)
{
    return read_back(c, c ? c->dosage : DevBuf(), 4, out, first, count, c ? c->T : 0, "uvrt_read_dosage");
}
int uvrt_read_color(uvrt_ctx* c, float* out9, int32_t first, int32_t count)
{
    return read_back(c, c ? c->color : DevBuf(), 36, out9, first, count, c ? c->T : 0, "uvrt_read_color");
}
int uvrt_read_counts(uvrt_ctx* c, int32_t* out, int32_t first, int32_t count)
{
    if (c && c->have_scene) {
        if (int rc = set_device(c)) return rc;
        if (int rc = join_all(c)) return rc;
        launch_fold_counts(lane_counts(c).as<int32_t>(), c->replicas, c->T, c->T, c->stream);
        if (int rc = mark_fence(c)) return rc;
    }
    return read_back(c, c ? lane_counts(c) : DevBuf(), 4, out, first, count, c ? c->T : 0, "uvrt_read_counts");
}
int uvrt_read_photon_map(uvrt_ctx* c, int32_t which, double* out, int32_t first, int32_t count)
{
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_read_photon_map: which_map must be 0 or 1");
    return read_back(c, c ? (which == UVRT_MAP_SUM ? c->photon_map : c->max_map) : DevBuf(), 8, out, first,
                     count, c ? c->T : 0, "uvrt_read_photon_map");
}

int uvrt_get_seed(uvrt_ctx* c, uint32_t* seed)
{
    if (!c || !seed) return fail(UVRT_ERR_INVALID, "null argument");
    *seed = c->seed;
    return UVRT_OK;
}
int uvrt_set_seed(uvrt_ctx* c, uint32_t seed)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->seed = seed;
    return UVRT_OK;
}

int uvrt_set_wide_bvh(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->wide = on != 0;
    return UVRT_OK;
}

int uvrt_set_hot_records(uvrt_ctx* c, int32_t mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(UVRT_ERR_INVALID, "uvrt_set_hot_records: mode must be 0 or 1");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->hot_mode = mode;
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) { c->xrecs_valid[l] = false; c->lane_perm[l] = nullptr; }
    return UVRT_OK;
}

int uvrt_set_seed_mode(uvrt_ctx* c, int32_t mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(UVRT_ERR_INVALID, "uvrt_set_seed_mode: mode must be 0 or 1");
    c->seed_mode = mode;
    return UVRT_OK;
}

int uvrt_set_sort_bits(uvrt_ctx* c, int32_t bits)
{
    if (!c || bits < -1 || bits > 20) return fail(UVRT_ERR_INVALID, "uvrt_set_sort_bits: bits must be in [-1,20]");
    c->sort_bits = bits;
    return UVRT_OK;
}
int uvrt_set_record_hits(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->record_hits = on != 0;
    return UVRT_OK;
}
int uvrt_set_flavour(uvrt_ctx* c, int32_t flavour)
{
    if (!c || flavour < 0 || flavour > 2) return fail(UVRT_ERR_INVALID, "uvrt_set_flavour: flavour must be 0, 1 or 2");
    c->flavour = flavour;
    return UVRT_OK;
}
int uvrt_set_variant(uvrt_ctx* c, int32_t variant)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (!variant_ok(variant))
        return fail(UVRT_ERR_INVALID, "uvrt_set_variant: %d is not a variant of this build (0, 400-1299; codes other than 1 in the "
                    "last digit need the developer build libuvrt_hip_dev.so)", variant);
    c->variant = variant;
    return UVRT_OK;
}
int uvrt_set_pipeline(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->pipeline = on != 0;
    return UVRT_OK;
}
int uvrt_set_timing(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->timing = on != 0;
    return UVRT_OK;
}

int uvrt_read_rays(uvrt_ctx* c, void* rays32, int64_t first, int64_t count)
{
    if (!c || !rays32 || c->last_n < 0 || first < 0 || count < 0 || first + count > c->last_n)
        return fail(UVRT_ERR_INVALID, "uvrt_read_rays: range outside the last generate");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    if (count == 0) return UVRT_OK;
    if (int rc = c->export_buf.ensure((size_t)count * 32, false, c->stream)) return rc;
    const uint2* hits = (c->last_extended && c->hits.p) ? c->hits.as<uint2>() : nullptr;
    launch_export_rays(lane_rays(c).as<float4>(), hits, c->export_buf.p, c->ox, c->oz, first, count, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rays32, c->export_buf.p, (size_t)count * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

int uvrt_write_rays(uvrt_ctx* c, const void* rays32, int64_t n)
{
    if (!c || !rays32 || n <= 0 || n > c->capacity)
        return fail(UVRT_ERR_INVALID, "uvrt_write_rays: n must be in (0, capacity]");
    if (int rc = set_device(c)) return rc;
    struct HostRay { float d[3], o[3], dist; uint32_t tri; };
    const HostRay* hr = (const HostRay*)rays32;
    std::vector<float> packed((size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) {
        if (memcmp(&hr[i].o[0], &hr[0].o[0], 4) != 0 || memcmp(&hr[i].o[2], &hr[0].o[2], 4) != 0)
            return fail(UVRT_ERR_INVALID, "uvrt_write_rays: record %lld has a different orig.x/orig.z", (long long)i);
        packed[4 * i + 0] = hr[i].d[0]; packed[4 * i + 1] = hr[i].d[1];
        packed[4 * i + 2] = hr[i].d[2]; packed[4 * i + 3] = hr[i].o[1];
    }
    if (int rc = join_all(c)) return rc;
    c->lane = 0;
    c->cur_pipelined = false;
    HIP_TRY(hipMemcpyAsync(c->rays.p, packed.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->recs_valid = false;
    c->lane_perm[0] = nullptr;
    c->last_n = n;
    c->last_first = 0;
    c->last_sorted = false;
    c->last_extended = false;
    c->ox = hr[0].o[0];
    c->oz = hr[0].o[2];
    return UVRT_OK;
}

int uvrt_device_ptr(uvrt_ctx* c, int32_t which, void** ptr, int64_t* bytes)
{
    if (!c || !ptr || !bytes || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: bad argument");
    // the caller is about to touch the buffers on the context's stream: order it after the side lane,
    // and the side lane's next work after whatever the caller enqueues up to the next call
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    const DevBuf* b = nullptr;
    size_t elem = 0;
    switch (which) {
        case 0: b = &c->photon_map; elem = 8; break;
        case 1: b = &c->max_map; elem = 8; break;
        case 2:
            b = &lane_counts(c); elem = 4;
            launch_fold_counts(lane_counts(c).as<int32_t>(), c->replicas, c->T, c->T, c->stream);
            break;
        case 3: b = &c->dosage; elem = 4; break;
        case 4: b = &c->color; elem = 36; break;
        case 5:
            if (c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: no traced batch");
            if (int rc = uvrt_fold_batch(c)) return rc;
            b = &c->bs[c->b_set].folded; elem = 4 * (size_t)c->b_count;
            break;
        default: return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: which must be 0..5");
    }
    *ptr = b->p;
    *bytes = (int64_t)((size_t)c->T * elem);
    // see lane_stream(): the next side-lane work (on the maps: the next accumulate / Shade) waits for
    // what the caller enqueues on the main stream up to the next call
    // (the folded planes of a batch are only touched on the context's stream until their replay: no fence)
    if (which == 2) c->ext_touch = true; else if (which != 5) c->ext_touch_maps = true;
    return UVRT_OK;
}

int uvrt_copy_device(uvrt_ctx* c, int32_t which, void* ext, int32_t to_ctx)
{
    void* p = nullptr;
    int64_t bytes = 0;
    if (!ext) return fail(UVRT_ERR_INVALID, "uvrt_copy_device: null pointer");
    if (int rc = uvrt_device_ptr(c, which, &p, &bytes)) return rc;
    if (int rc = set_device(c)) return rc;
    if (to_ctx) HIP_TRY(hipMemcpyAsync(p, ext, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(hipMemcpyAsync(ext, p, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    return mark_fence(c);
}

int uvrt_extend_time_ms(uvrt_ctx* c, double* ms, int64_t* launches)
{
    if (!c || !ms || !launches) return fail(UVRT_ERR_INVALID, "null argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, c->ev_pool[i].first, c->ev_pool[i].second));
        total += t;
    }
    *ms = total;
    *launches = (int64_t)c->ev_used;
    c->ev_used = 0;
    return UVRT_OK;
}

}  // extern "C"
