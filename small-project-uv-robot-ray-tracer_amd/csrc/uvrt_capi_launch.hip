// uvrt_capi_launch.hip -- the per-launch entry points: generate, extend, accumulate, Shade (raytracer.cpp:74-120)
// (the C ABI of include/uvrt.h over the HIP kernels; the context and its helpers are in uvrt_ctx.h)
#include "uvrt_ctx.h"

using namespace uvrt;
using namespace uvrt_impl;

namespace uvrt_impl {

int hot_reset(uvrt_ctx* c, bool slab)
{
    for (auto& h : c->hot) (void)hipEventDestroy(h.ready);
    c->hot.clear();
    for (DevBuf& b : c->hot_slabs) b.release();
    c->hot_slabs.clear();
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) { c->hot_hist[l].release(); c->hot_list[l].release(); c->lane_perm[l] = nullptr; }
    if (!slab || c->npairs <= 128) return UVRT_OK;
    // the first slab and the lanes' scratch come with the scene: a new lamp position then costs no allocation
    c->hot_slabs.emplace_back();
    if (int rc = c->hot_slabs.back().ensure((size_t)uvrt_ctx::HOT_SLAB * (size_t)c->npairs * 4, false, c->stream)) return rc;
    for (int l = 0; l < c->nlanes || l < 3; ++l) {
        // lane 0 (uvrt_trace_batch) sets up all new lamps of a batch in one launch: room for HS_GROUPS of them where that is small
        const size_t groups = (l == 0 && (size_t)c->npairs * 4 * HS_GROUPS <= ((size_t)64 << 20)) ? (size_t)HS_GROUPS : 1;
        if (int rc = c->hot_hist[l].ensure(groups * (size_t)c->npairs * 4, true, c->stream)) return rc;
        if (int rc = c->hot_list[l].ensure(groups * ((size_t)TOP6_MAX + 1) * 4, false, c->stream)) return rc;
    }
    return UVRT_OK;
}

int hot_lookup(uvrt_ctx* c, const float lamp[3], hipStream_t s, const uint32_t** out, uvrt_ctx::HotEntry** fresh)
{
    *out = nullptr;
    *fresh = nullptr;
    if (c->have_perm) { *out = c->perm.as<uint32_t>(); return UVRT_OK; }
    if (c->hot_mode == 0 || c->npairs <= (int32_t)128 || c->root_ref >= REF_LEAF_BIT) return UVRT_OK;
    ++c->hot_clock;
    for (auto& h : c->hot)
        if (memcmp(h.lamp, lamp, 12) == 0) {
            h.stamp = c->hot_clock;
            HIP_TRY(hipStreamWaitEvent(s, h.ready, 0));      // it may have been built on another lane's stream
            *out = h.perm;
            return UVRT_OK;
        }
    uvrt_ctx::HotEntry* e = nullptr;
    if ((int)c->hot.size() < uvrt_ctx::HOT_MAX) {
        const size_t idx = c->hot.size();
        if (idx / uvrt_ctx::HOT_SLAB >= c->hot_slabs.size()) {
            c->hot_slabs.emplace_back();
            if (int rc = c->hot_slabs.back().ensure((size_t)uvrt_ctx::HOT_SLAB * (size_t)c->npairs * 4, false, s)) {
                c->hot_slabs.pop_back();
                return rc;
            }
        }
        uvrt_ctx::HotEntry ne;
        memset(&ne, 0, sizeof ne);
        ne.perm = c->hot_slabs[idx / uvrt_ctx::HOT_SLAB].as<uint32_t>() + (idx % uvrt_ctx::HOT_SLAB) * (size_t)c->npairs;
        HIP_TRY(hipEventCreateWithFlags(&ne.ready, hipEventDisableTiming));
        c->hot.push_back(ne);                 // (the vector's capacity is HOT_MAX from the start: entries never move)
        e = &c->hot.back();
    } else {          // recycle the least recently used entry: nothing in flight may still read its renumbering
        if (int rc = join_all(c)) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        e = &c->hot[0];
        for (auto& h : c->hot) if (h.stamp < e->stamp) e = &h;
    }
    memcpy(e->lamp, lamp, 12);
    e->stamp = c->hot_clock;
    e->gen = ++c->perm_clock;            // records prepared from the entry's previous renumbering are stale (uvrt_trace_batch's keys)
    *fresh = e;
    return UVRT_OK;
}

int hot_build(uvrt_ctx* c, uvrt_ctx::HotEntry* const* entries, const uint32_t* seed_prev, const uint32_t* seed_next, int count,
              float light_length, hipStream_t s, int lane)
{
    for (int k0 = 0; k0 < count; k0 += HS_GROUPS) {
        const int kc = std::min(HS_GROUPS, count - k0);
        // scratch of the set-up kernels, per launch lane: visit counters (zero between uses) and the hot lists
        if (int rc = c->hot_hist[lane].ensure((size_t)kc * (size_t)c->npairs * 4, true, s)) return rc;
        if (int rc = c->hot_list[lane].ensure((size_t)kc * ((size_t)TOP6_MAX + 1) * 4, false, s)) return rc;
        HotSetupParams p;
        memset(&p, 0, sizeof p);
        p.pairs = c->pairs.as<PairRec>();
        p.ltris = c->ltris.as<LeafTri>();
        p.leaf_count = c->leaf_count.as<uint32_t>();
        p.root_ref = c->root_ref;
        p.light_length = light_length;
        p.seed_mode = c->seed_mode;
        // the statistics always sample global ids [0, hot_sample) of the lamp (the kernel makes its own rays), whichever
        // range of the launch this context traces
        p.n = c->hot_sample;
        p.tail_lanes = c->hot_tail;
        p.direct_bins = c->hot_direct;
        p.npairs = c->npairs;
        p.keep = (int32_t)TOP6_MAX;
        p.count = kc;
        p.hist = c->hot_hist[lane].as<uint32_t>();
        p.hot_list = c->hot_list[lane].as<uint32_t>();
        for (int k = 0; k < kc; ++k) {
            const uvrt_ctx::HotEntry* e = entries[k0 + k];
            p.perm[k] = e->perm;
            p.lx[k] = e->lamp[0]; p.ly[k] = e->lamp[1]; p.lz[k] = e->lamp[2];
            p.seed_prev[k] = seed_prev[k0 + k];
            p.seed_next[k] = seed_next[k0 + k];
        }
        launch_hot_setup(p, s);
        HIP_TRY(hipGetLastError());
        for (int k = 0; k < kc; ++k) HIP_TRY(hipEventRecord(entries[k0 + k]->ready, s));
    }
    return UVRT_OK;
}

int launch_perm(uvrt_ctx* c, const float lamp[3], float light_length, uint32_t seed_prev, uint32_t seed_next,
                hipStream_t s, int lane, const uint32_t** out)
{
    uvrt_ctx::HotEntry* fresh = nullptr;
    if (int rc = hot_lookup(c, lamp, s, out, &fresh)) return rc;
    if (!fresh) return UVRT_OK;
    if (int rc = hot_build(c, &fresh, &seed_prev, &seed_next, 1, light_length, s, lane)) return rc;
    *out = fresh->perm;
    return UVRT_OK;
}

}  // namespace uvrt_impl

extern "C" {

uint32_t uvrt_seed_next(const float lp[3], float light_length, uint32_t seed_prev)
{
    return uvrt_seed_next_mode(lp, light_length, seed_prev, 0);
}

uint32_t uvrt_seed_next_mode(const float lp[3], float light_length, uint32_t seed_prev, int32_t seed_mode)
{
    // work-item 0 of cl/generate.cl:13-39; the ray itself is not needed, only the RNG state
    float acc = (float)(0 * 17 + 1);
    acc = acc + lp[0] * 13.0f;
    acc = acc + lp[1] * 7.0f;
    acc = acc + lp[2] * 11.0f;
    acc = acc + (float)(seed_prev >> 15);
    uint32_t seed = host_wang_hash((seed_mode == 1 && acc < 0.0f) ? 0u : (uint32_t)(int64_t)acc);
    (void)light_length;
    (void)host_random_float(seed);   // origin.y
    (void)host_random_float(seed);   // diry
    double x = (double)(host_random_float(seed) * 2.0f - 1.0f);
    double y = (double)(host_random_float(seed) * 2.0f - 1.0f);
    while (x * x + y * y > 1.0) {
        x = (double)(host_random_float(seed) * 2.0f - 1.0f);
        y = (double)(host_random_float(seed) * 2.0f - 1.0f);
    }
    return seed;
}

int uvrt_generate(uvrt_ctx* c, const float lp[3], float light_length, int64_t first_gid, int64_t n)
{
    if (!c || !lp) return fail(UVRT_ERR_INVALID, "uvrt_generate: null argument");
    if (n < 0 || first_gid < 0 || n > c->capacity)
        return fail(UVRT_ERR_INVALID, "uvrt_generate: n = %lld exceeds the ray capacity %lld (uvrt_resize_rays)",
                    (long long)n, (long long)c->capacity);
    if (first_gid + n > (int64_t)INT32_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_generate: global id beyond int range (generate.cl:11)");
    if (int rc = set_device(c)) return rc;

    const uint32_t seed_prev = c->seed;
    const uint32_t seed_next = uvrt_seed_next_mode(lp, light_length, seed_prev, c->seed_mode);

    int bits = c->sort_bits < 0 ? auto_sort_bits(n) : c->sort_bits;
    if (bits > 20) bits = 20;
    // launch lane: alternate between the two streams / buffer sets when nothing stands against it
    {
        const bool pipe_ok = c->pipeline && c->nlanes > 1 && !c->record_hits && bits == 0 &&
                             c->xrays[1].p;
        c->prev_lane = c->lane;
        c->cur_pipelined = pipe_ok;
        if (pipe_ok) c->lane = (c->lane + 1) % c->nlanes;
        else { if (int rc = join_all(c)) return rc; c->lane = 0; }
        if (c->lane != 0) {
            // 8 workgroups per CU x 256 threads x 24 overflow entries (the largest grid a side lane runs)
            if (int rc = c->xovf[c->lane].ensure((size_t)c->num_cus * 8 * 256 * 24 * sizeof(uint32_t), false, c->stream)) return rc;
        }
    }
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls)) return rc;
    GenParams p;
    memset(&p, 0, sizeof p);
    p.rays = lane_rays(c).as<float4>();
    p.lx = lp[0]; p.ly = lp[1]; p.lz = lp[2];
    p.light_length = light_length;
    p.first_gid = first_gid;
    p.n = n;
    p.seed_prev = seed_prev;
    p.seed_next = seed_next;
    p.seed_mode = c->seed_mode;
    if (bits > 0 && n > 0) {
        const int32_t nbins = 1 << bits;
        if (c->hist_bins < nbins) {
            int rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->hist.release();
            c->bin_start.release();
            if ((rc = c->hist.ensure((size_t)nbins * 4, true, c->stream))) return rc;
            if ((rc = c->bin_start.ensure((size_t)nbins * 4, true, c->stream))) return rc;
            c->hist_bins = nbins;
        }
        p.keyrank = c->keyrank.as<uint2>();
        p.hist = c->hist.as<uint32_t>();
        split_bits(bits, p.bits_phi, p.bits_y, p.bits_o);
    }
    if (c->npairs > 0) {   // extend's per-launch records ride along in the same launch
        p.prep_pairs = c->pairs.as<PairRec>();
        p.prep_recs = lane_recs(c).as<float4>();
        // launches too small to repay the statistics keep the breadth-first order
        const uint32_t* pm = c->have_perm ? c->perm.as<uint32_t>() : nullptr;
        if (!pm && n >= 16384)
            if (int rc = launch_perm(c, lp, light_length, seed_prev, seed_next, ls, c->lane, &pm)) return rc;
        c->lane_perm[c->lane] = pm;
        p.prep_perm = pm;
        p.prep_npairs = c->npairs;
    }
    launch_generate(p, ls);
    HIP_TRY(hipGetLastError());
    (c->lane ? c->xrecs_valid[c->lane] : c->recs_valid) = p.prep_recs != nullptr;
    (c->lane ? c->xrecs_ox[c->lane] : c->recs_ox) = lp[0];
    (c->lane ? c->xrecs_oz[c->lane] : c->recs_oz) = lp[2];
    if (p.keyrank) {
        launch_scan_bins(c->hist.as<uint32_t>(), c->bin_start.as<uint32_t>(), 1 << bits, c->stream);
        launch_scatter(c->rays.as<float4>(), c->keyrank.as<uint2>(), c->bin_start.as<uint32_t>(),
                       c->sorted.as<float4>(), c->order.as<uint32_t>(), n, c->stream);
        HIP_TRY(hipGetLastError());
    }
    c->seed = seed_next;
    c->last_n = n;
    c->last_first = first_gid;
    c->last_sorted = p.keyrank != nullptr;
    c->last_extended = false;
    c->ox = lp[0];
    c->oz = lp[2];
    return UVRT_OK;
}

int uvrt_extend(uvrt_ctx* c, int64_t n)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_extend: no scene");
    if (c->last_n < 0 || n != c->last_n)
        return fail(UVRT_ERR_INVALID, "uvrt_extend: n = %lld does not match the last generate (%lld)",
                    (long long)n, (long long)c->last_n);
    if (int rc = set_device(c)) return rc;
    if (c->record_hits) {
        if (int rc = c->hits.ensure((size_t)c->capacity * 8, false, c->stream)) return rc;
    }
    ExtendParams p;
    memset(&p, 0, sizeof p);
    p.scene.pairs = c->pairs.as<PairRec>();
    p.scene.ltris = c->ltris.as<LeafTri>();
    p.scene.leaf_count = c->leaf_count.as<uint32_t>();
    p.scene.root_ref = c->root_ref;
    p.scene.tri_count = c->T;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls)) return rc;
    p.rays = c->last_sorted ? c->sorted.as<float4>() : lane_rays(c).as<float4>();
    {
        // conditions of the reciprocal shortcut that are uniform over the launch (slab<>())
        const float ax = std::fabs(c->ox), az = std::fabs(c->oz);
        const float tiny = 7.888609e-31f;   // 2^-100
        p.force_exact = (c->scene_force_exact || (ax != 0.0f && ax < tiny) || (az != 0.0f && az < tiny) ||
                         !(ax <= 1e9f) || !(az <= 1e9f)) ? 1 : 0;
    }
    p.order = c->last_sorted ? c->order.as<uint32_t>() : nullptr;
    p.hits = c->record_hits ? c->hits.as<uint2>() : nullptr;
    p.ovf_stack = lane_ovf(c).as<uint32_t>();
    p.ovf_capacity = lane_ovf(c).bytes / sizeof(uint32_t);
    p.num_cus = lane_cus(c);
    p.flavour = c->flavour;
    p.top_pairs = c->top_pairs;
    p.counts = lane_counts(c).as<int32_t>();
    p.count_replicas = c->replicas;
    p.count_stride = c->T;
    p.error_flag = c->host_flag_dev ? c->host_flag_dev : c->error_flag.as<uint32_t>();
    p.ox = c->ox;
    p.oz = c->oz;
    p.n = n;
    p.npairs = c->npairs;
    p.recs = lane_recs(c).p;
    p.perm = c->have_perm ? c->perm.as<uint32_t>() : c->lane_perm[c->lane];
    {
        const bool valid = c->lane ? c->xrecs_valid[c->lane] : c->recs_valid;
        const float rox = c->lane ? c->xrecs_ox[c->lane] : c->recs_ox, roz = c->lane ? c->xrecs_oz[c->lane] : c->recs_oz;
        p.recs_prepared = (valid && memcmp(&rox, &c->ox, 4) == 0 && memcmp(&roz, &c->oz, 4) == 0) ? 1 : 0;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->ev_pool.emplace_back(a, b);
        }
        e0 = c->ev_pool[c->ev_used].first;
        e1 = c->ev_pool[c->ev_used].second;
        ++c->ev_used;
        HIP_TRY(hipEventRecord(e0, ls));
    }
    if (c->wide && c->nquads > 0) {
        // the opt-in 4-wide walk: its per-launch records are (re)made here when the lane's are for another lamp
        DevBuf& r4 = c->recs4[c->lane];
        if (!r4.p) {
            if (int rc = r4.ensure(((size_t)2 * c->nquads + (size_t)c->T + 1) * 64, true, ls)) return rc;
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), r4.p, 2 * c->nquads, c->T, ls);
            c->recs4_valid[c->lane] = false;
        }
        if (!c->recs4_valid[c->lane] || memcmp(&c->recs4_ox[c->lane], &c->ox, 4) != 0 || memcmp(&c->recs4_oz[c->lane], &c->oz, 4) != 0) {
            launch_prepare_launch4(c->quads.as<QuadRec>(), r4.p, c->ox, c->oz, c->nquads, ls);
            c->recs4_valid[c->lane] = true;
            c->recs4_ox[c->lane] = c->ox;
            c->recs4_oz[c->lane] = c->oz;
        }
        p.recs4 = r4.p;
        p.nquads = c->nquads;
        p.top_quads = c->top_quads;
        p.refill_min = 8;
        if (c->variant >= 500 && c->variant < 600) p.force_exact = 1;
        if (!launch_extend4(p, 7, ls)) return fail(UVRT_ERR_INVALID, "uvrt_extend: overflow-stack buffer too small for the 4-wide kernel");
        HIP_TRY(hipGetLastError());
        if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
        c->counts_dirty[c->lane] = true;
        c->last_extended = c->record_hits;
        return UVRT_OK;
    }
    if (c->variant >= 500 && c->variant < 600) p.force_exact = 1;
    p.refill_min = variant_refill_min(c->variant, (size_t)c->npairs + (size_t)c->T);
    // default grid: 8 workgroups per CU on one stream (20 KB of LDS each: eight fit a CU); 7 when launches are
    // pipelined over several streams -- the free slot per CU lets the first workgroups of the next launch and the
    // small kernels around it (generate, accumulate, replay) run at once instead of queueing behind persistent waves
    // (profiles/r02/r02_experiments.txt); with four launch lanes 4 per CU
    const int per_cu_default = (c->cur_pipelined && c->nlanes >= 4) ? 4 : c->cur_pipelined ? 7 : 8;
    p.drain_merge = c->drain_merge;
    if (!launch_extend6(p, variant_code6(c->variant), variant_per_cu(c->variant, per_cu_default), ls))
        return fail(UVRT_ERR_INVALID, "uvrt_extend: variant %d needs a larger overflow-stack buffer than the context holds", c->variant);
    HIP_TRY(hipGetLastError());
    if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
    c->counts_dirty[c->lane] = true;
    c->last_extended = c->record_hits;
    return UVRT_OK;
}

int uvrt_accumulate(uvrt_ctx* c, float time_step, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_accumulate: bad tri_count");
    if (int rc = set_device(c)) return rc;       // (an earlier deferred accumulate goes first)
    // the maps are updated in launch order: wait for whatever the other lane has enqueued so far
    // (its accumulate and shade), not for this lane's successor
    if (int rc = order_after_previous(c)) return rc;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    if (tri_count == c->T && c->T > 0) {
        // deferred: the ordering is enqueued, the launch waits for the next call -- a uvrt_shade takes it along in one
        // kernel (uvrt_ctx.h PendingAcc), anything else launches it first
        c->pend.valid = true;
        c->pend.lane = c->lane;
        c->pend.time_step = time_step;
        c->counts_dirty[c->lane] = false;
        return UVRT_OK;
    }
    launch_accumulate(c->photon_map.as<double>(), c->max_map.as<double>(), lane_counts(c).as<int32_t>(),
                      c->replicas, c->T, time_step, tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_compute_dosage(uvrt_ctx* c, int32_t which, int32_t photons_per_light, float scaled_power,
                        int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_compute_dosage: bad tri_count");
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_compute_dosage: which_map must be 0 or 1");
    if (int rc = set_device(c)) return rc;
    const double* map = which == UVRT_MAP_SUM ? c->photon_map.as<double>() : c->max_map.as<double>();
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_compute_dosage(map, c->dosage.as<float>(), c->area.as<float>(), photons_per_light,
                          scaled_power, tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_dosage_to_color(uvrt_ctx* c, float min_value, int32_t threshold_view, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_dosage_to_color: bad tri_count");
    if (int rc = set_device(c)) return rc;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_dosage_to_color(c->dosage.as<float>(), c->color.as<float>(), min_value, threshold_view,
                           tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_shade(uvrt_ctx* c, int32_t which, int32_t photons_per_light, float scaled_power, float min_value,
               int32_t threshold_view, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_shade: bad tri_count");
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_shade: which_map must be 0 or 1");
    if (c->pend.valid && c->pend.lane == c->lane && tri_count == c->T) {
        // the launch's accumulate is still pending on this lane: accumulate + computeDosage + dosageToColor in one kernel
        if (int rc = set_device_only(c)) return rc;
        hipStream_t fs;
        if (int rc = lane_stream(c, &fs, true)) return rc;
        c->pend.valid = false;
        launch_accumulate_shade(c->photon_map.as<double>(), c->max_map.as<double>(), lane_counts(c).as<int32_t>(), c->replicas,
                                c->T, c->pend.time_step, c->dosage.as<float>(), c->area.as<float>(), c->color.as<float>(),
                                which == UVRT_MAP_SUM ? 0 : 1, photons_per_light, scaled_power, min_value, threshold_view,
                                tri_count, fs);
        HIP_TRY(hipGetLastError());
        return UVRT_OK;
    }
    if (int rc = set_device(c)) return rc;
    const double* map = which == UVRT_MAP_SUM ? c->photon_map.as<double>() : c->max_map.as<double>();
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_shade(map, c->dosage.as<float>(), c->area.as<float>(), c->color.as<float>(), photons_per_light,
                 scaled_power, min_value, threshold_view, tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_advance_seed(uvrt_ctx* c, const float lp[3], float light_length)
{
    if (!c || !lp) return fail(UVRT_ERR_INVALID, "uvrt_advance_seed: null argument");
    c->seed = uvrt_seed_next_mode(lp, light_length, c->seed, c->seed_mode);
    return UVRT_OK;
}

}  // extern "C"
