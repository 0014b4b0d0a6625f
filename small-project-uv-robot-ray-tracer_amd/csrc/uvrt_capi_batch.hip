// uvrt_capi_batch.hip -- batched tracing: several launches in one go, one count plane per launch
// (the C ABI of include/uvrt.h over the HIP kernels; the context and its helpers are in uvrt_ctx.h)
#include "uvrt_ctx.h"

using namespace uvrt;
using namespace uvrt_impl;

extern "C" {

int uvrt_trace_batch(uvrt_ctx* c, const float* lamps, float light_length, int32_t count, int64_t first_gid, int64_t n)
{
    if (!c || !lamps || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: null argument or no scene");
    if (count <= 0 || count > MAX_BATCH) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: count must be in [1,%d]", MAX_BATCH);
    if (n <= 0 || first_gid < 0 || first_gid + n > (int64_t)INT32_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: bad global-id range");
    if (c->b_count > 0) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: the previous batch has not been replayed (uvrt_replay_batch)");
    if (c->record_hits || c->sort_bits != 0)
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: per-ray hit records and ray ordering are per-launch features");
    if (int rc = set_device(c)) return rc;
    const int64_t n_pad = (n + 63) / 64 * 64;
    // deposit replicas per plane: the contention on a hot triangle's counter grows with the rays per plane
    // (16 replicas for 2 M rays), and every replica is read and zeroed again by the replay -- a shard of a launch
    // gets by with 8 (one per XCD)
    // (a fused batch is happiest with 8-12 replicas even at 2 M rays per plane: +0.4 % over 16, profiles/r03/r03_knobs_room.txt;
    // the per-launch path keeps the context's 16)
    int R = std::min(c->replicas, 12);
    if ((int64_t)R * 131072 > 2 * n) R = std::min(c->replicas, 8);
    if ((uint64_t)count * (uint64_t)n_pad >= ((uint64_t)1 << 30) || (uint64_t)count * (uint64_t)R * (uint64_t)c->T >= ((uint64_t)1 << 32))
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: %d launches x %lld rays exceed one batch (2^30 ray slots, 2^32 counters)", count, (long long)n);

    // group the launches by lamp column (x, z): the per-launch node-pair records depend on it only
    int group_of[MAX_BATCH], ngroups = 0, gfirst[MAX_BATCH], gsize[MAX_BATCH] = {};
    float gx[MAX_BATCH], gz[MAX_BATCH];
    for (int k = 0; k < count; ++k) {
        int g = 0;
        for (; g < ngroups; ++g)
            if (memcmp(&gx[g], &lamps[3 * k], 4) == 0 && memcmp(&gz[g], &lamps[3 * k + 2], 4) == 0) break;
        if (g == ngroups) { gx[g] = lamps[3 * k]; gz[g] = lamps[3 * k + 2]; ++ngroups; }
        group_of[k] = g;
        ++gsize[g];
    }
    for (int g = 0, acc = 0; g < ngroups; ++g) { gfirst[g] = acc; acc += gsize[g]; }
    GenBatchParams gp;
    memset(&gp, 0, sizeof gp);
    uint32_t seed_after = c->seed;
    {
        int fill[MAX_BATCH] = {};
        uint32_t seed = c->seed;
        for (int k = 0; k < count; ++k) {                    // logical order: the SEED chain
            const int g = group_of[k], ph = gfirst[g] + fill[g]++;
            c->b_phys[k] = ph;
            gp.lx[ph] = lamps[3 * k]; gp.ly[ph] = lamps[3 * k + 1]; gp.lz[ph] = lamps[3 * k + 2];
            gp.seed_prev[ph] = seed;
            seed = uvrt_seed_next_mode(&lamps[3 * k], light_length, seed, c->seed_mode);
            gp.seed_next[ph] = seed;
        }
        seed_after = seed;              // committed with the batch: a failed call leaves the SEED chain where it was
    }
    // The batch goes into the buffer set the previous batch did NOT use: its lanes start at once -- in the drain of
    // the previous batch, while that one is still being folded / reduced / replayed on the context's stream -- and
    // only wait for the set's last replay (free_ev), which is two batches back.  Anything that has to touch memory
    // the lanes may still read (growing a buffer, new per-launch records) first waits for everything.
    const int set = c->b_set ^ 1;
    uvrt_ctx::BatchSet& S = c->bs[set];
    int rc;
    const size_t plane_ints = (size_t)R * (size_t)c->T;
    // full planes are allocated for the context's replica count: R only shrinks the part of it that is used
    const size_t plane_alloc = (size_t)c->replicas * (size_t)c->T;
    // growing a buffer needs the device idle (hipFree / hipMalloc); new per-launch records only need the context's stream
    // ordered after the lanes' earlier work -- no host synchronisation, so a new lamp position costs its kernels only
    const bool need_alloc = S.rays.bytes < (size_t)count * (size_t)n_pad * 16 || S.planes.bytes < (size_t)count * plane_alloc * 4 ||
                            S.folded.bytes < (size_t)count * (size_t)c->T * 4 || (int)c->b_recs.size() < ngroups || !S.free_ev;
    bool recs_stale = false;
    const uint32_t* gperm[MAX_BATCH] = {};
    uint64_t ggen[MAX_BATCH] = {};
    uvrt_ctx::HotEntry* fresh[MAX_BATCH];               // lamps the context has not seen: their set-ups go in ONE launch
    uint32_t fresh_prev[MAX_BATCH], fresh_next[MAX_BATCH];
    int nfresh = 0;
    for (int g = 0; g < ngroups; ++g) {
        gperm[g] = c->have_perm ? c->perm.as<uint32_t>() : nullptr;
        bool is_fresh = false;
        if (!gperm[g] && (int64_t)gsize[g] * n >= 16384) {
            const int ph = gfirst[g];      // the group's first launch lends its lamp and seeds to the statistics
            const float gl[3] = {gp.lx[ph], gp.ly[ph], gp.lz[ph]};
            uvrt_ctx::HotEntry* e = nullptr;
            if (int rcp = hot_lookup(c, gl, c->stream, &gperm[g], &e)) return rcp;
            if (e) {
                gperm[g] = e->perm;
                fresh[nfresh] = e; fresh_prev[nfresh] = gp.seed_prev[ph]; fresh_next[nfresh] = gp.seed_next[ph];
                ++nfresh;
                is_fresh = true;
            }
        }
        ggen[g] = perm_generation(c, gperm[g]);
        if (g >= (int)c->b_recs_key.size() || c->b_recs_key[g].perm != gperm[g] || c->b_recs_key[g].gen != ggen[g] || memcmp(&c->b_recs_key[g].ox, &gx[g], 4) != 0 ||
            memcmp(&c->b_recs_key[g].oz, &gz[g], 4) != 0 || is_fresh) {
            recs_stale = true;
            if (g < (int)c->b_recs_key.size()) c->b_recs_key[g].valid = false;
        }
    }
    if (nfresh > 0)
        if (int rcb = hot_build(c, fresh, fresh_prev, fresh_next, nfresh, light_length, c->stream, 0)) return rcb;
    if (need_alloc || recs_stale) {
        if (int rcj = join_all(c)) return rcj;
    }
    if (need_alloc) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!S.free_ev) HIP_TRY(hipEventCreateWithFlags(&S.free_ev, hipEventDisableTiming));
        const bool grown = S.planes.bytes < (size_t)count * plane_alloc * 4 || S.folded.bytes < (size_t)count * (size_t)c->T * 4;
        if ((rc = S.rays.ensure((size_t)count * (size_t)n_pad * 16, false, c->stream))) return rc;
        if ((rc = S.planes.ensure((size_t)count * plane_alloc * 4, true, c->stream))) return rc;
        if ((rc = S.folded.ensure((size_t)count * (size_t)c->T * 4, true, c->stream))) return rc;
        if (grown) HIP_TRY(hipEventRecord(S.free_ev, c->stream));      // the zero fill is the set's "last replay"
        while ((int)c->b_recs.size() < ngroups) {
            DevBuf b;
            if ((rc = b.ensure(((size_t)c->npairs + (size_t)c->T + 1) * 64, true, c->stream))) return rc;
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), b.p, c->npairs, c->T, c->stream);
            c->b_recs.push_back(b);
        }
        c->b_recs_key.resize(c->b_recs.size());
        for (int l = 1; l <= c->batch_lanes; ++l)
            if ((rc = c->xovf[l].ensure((size_t)c->num_cus * 8 * 256 * 24 * sizeof(uint32_t), false, c->stream))) return rc;
    }
    if (need_alloc || recs_stale) {
        // per-launch records of the lamp columns whose array holds something else
        for (int g = 0; g < ngroups; ++g) {
            uvrt_ctx::RecsKey& key = c->b_recs_key[g];
            if (key.perm == gperm[g] && key.gen == ggen[g] && key.valid && memcmp(&key.ox, &gx[g], 4) == 0 && memcmp(&key.oz, &gz[g], 4) == 0) continue;
            launch_prepare_launch6(c->pairs.as<PairRec>(), c->b_recs[g].p, gx[g], gz[g], c->npairs, gperm[g], c->stream);
            key.ox = gx[g]; key.oz = gz[g]; key.perm = gperm[g]; key.gen = ggen[g]; key.valid = true;
        }
        HIP_TRY(hipGetLastError());
        if (int rcf = mark_fence(c)) return rcf;         // the lanes' next work waits for the records
    }
    // Launches in CHUNKS of a few planes: generate + fused extend of a chunk on one launch lane, chunks alternating
    // over the lanes.  A chunk's rays (16 B each) are sized to stay in the Infinity Cache between the generate
    // that writes them and the extend that reads them (a refill that has to go to HBM stalls its wave for
    // microseconds), and the next chunk's generate and first waves run in the drain of the previous one.
    bool lane_waited[uvrt_ctx::MAXL] = {};
    const int per_chunk = (int)std::max<size_t>(1, c->batch_chunk_bytes / ((size_t)n_pad * 16));
    // every error return below leaves the launch-lane rotation as it found it
    struct LaneGuard {
        uvrt_ctx* c; int lane; uint64_t chunks; bool armed;
        ~LaneGuard() { if (armed) { c->lane = lane; c->b_chunks = chunks; } }
    } guard{c, c->lane, c->b_chunks, true};
    int chunk_index = 0;
    for (int g = 0; g < ngroups; ++g) {
        for (int k0 = 0; k0 < gsize[g]; k0 += per_chunk, ++chunk_index) {
            const int kc = std::min(per_chunk, gsize[g] - k0), ph0 = gfirst[g] + k0;
            // two SIDE lanes in turn: the context's own stream carries the fold / reduce / replay of the previous batch,
            // which a chunk enqueued there would have to wait for
            c->lane = c->pipeline ? 1 + (int)(c->b_chunks++ % (uint64_t)c->batch_lanes) : 0;
            // the chunk's rays depend on nothing but their buffer: generate goes to the lane BEFORE the lane waits for the
            // context's stream (new records, a hot-record set-up), so it runs beside them
            hipStream_t ls = stream_of(c, c->lane);
            if (c->lane != 0) c->side_used[c->lane] = true;
            if (!lane_waited[c->lane]) {      // the set's previous occupant has been replayed (two batches back)
                HIP_TRY(hipStreamWaitEvent(ls, S.free_ev, 0));
                lane_waited[c->lane] = true;
            }
            GenBatchParams gq;
            memset(&gq, 0, sizeof gq);
            gq.rays = S.rays.as<float4>() + (size_t)ph0 * (size_t)n_pad;
            gq.n_pad = n_pad;
            gq.first_gid = first_gid;
            gq.n = n;
            gq.light_length = light_length;
            gq.seed_mode = c->seed_mode;
            gq.count = kc;
            for (int j = 0; j < kc; ++j) {
                gq.lx[j] = gp.lx[ph0 + j]; gq.ly[j] = gp.ly[ph0 + j]; gq.lz[j] = gp.lz[ph0 + j];
                gq.seed_prev[j] = gp.seed_prev[ph0 + j]; gq.seed_next[j] = gp.seed_next[ph0 + j];
            }
#ifdef UVRT_DEV_VARIANTS
            if (!(c->probe_skip_generate > 0 && c->probe_batches >= c->probe_skip_generate))
#endif
            launch_generate_batch(gq, ls);
            if (int rcl = lane_stream(c, &ls)) return rcl;      // extend: after the fence
            ExtendParams p;
            memset(&p, 0, sizeof p);
            p.scene.pairs = c->pairs.as<PairRec>();
            p.scene.ltris = c->ltris.as<LeafTri>();
            p.scene.leaf_count = c->leaf_count.as<uint32_t>();
            p.scene.root_ref = c->root_ref;
            p.scene.tri_count = c->T;
            p.rays = gq.rays;
            {
                const float ax = std::fabs(gx[g]), az = std::fabs(gz[g]);
                const float tiny = 7.888609e-31f;   // 2^-100
                p.force_exact = (c->scene_force_exact || (ax != 0.0f && ax < tiny) || (az != 0.0f && az < tiny) ||
                                 !(ax <= 1e9f) || !(az <= 1e9f) || (c->variant >= 500 && c->variant < 600)) ? 1 : 0;
            }
            p.ovf_stack = lane_ovf(c).as<uint32_t>();
            p.ovf_capacity = lane_ovf(c).bytes / sizeof(uint32_t);
            p.num_cus = lane_cus(c);
            p.flavour = c->flavour;
            p.top_pairs = c->top_pairs;
            p.counts = S.planes.as<int32_t>() + (size_t)ph0 * plane_ints;
            p.count_replicas = R;
            p.count_stride = c->T;
            p.error_flag = c->host_flag_dev ? c->host_flag_dev : c->error_flag.as<uint32_t>();
            p.ox = gx[g];
            p.oz = gz[g];
            p.n = (int64_t)kc * n_pad;
            p.npairs = c->npairs;
            p.recs = c->b_recs[g].p;
            p.perm = gperm[g];
            p.recs_prepared = 1;
            p.drain_merge = c->drain_merge;
            p.refill_min = variant_refill_min(c->variant, (size_t)c->npairs + (size_t)c->T);
            p.plane_batches = (uint32_t)(n_pad / 64);
            p.plane_n = (uint32_t)n;
            p.plane_stride = (uint32_t)plane_ints;
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
            if (!launch_extend6(p, variant_code6(c->variant), variant_per_cu(c->variant, c->pipeline ? 7 : 8), ls)) {
                return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: variant %d needs a larger overflow-stack buffer", c->variant);
            }
            HIP_TRY(hipGetLastError());
            if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
        }
    }
#ifdef UVRT_DEV_VARIANTS
    ++c->probe_batches;
#endif
    guard.armed = false;
    c->seed = seed_after;
    c->lane = 0;
    c->cur_pipelined = false;
    c->last_n = -1;                      // the per-launch generate/extend pairing starts afresh
    c->b_set = set;
    c->b_repl = R;
    c->b_count = count;
    c->b_n = n;
    c->b_npad = n_pad;
    c->b_is_folded = false;
    return UVRT_OK;
}

int uvrt_fold_batch(uvrt_ctx* c)
{
    if (!c || c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_fold_batch: no traced batch");
    if (c->b_is_folded) return UVRT_OK;
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    launch_fold_planes(c->bs[c->b_set].planes.as<int32_t>(), c->bs[c->b_set].folded.as<int32_t>(), c->b_count, c->b_repl, c->T, c->stream);
    HIP_TRY(hipGetLastError());
    c->b_is_folded = true;
    return UVRT_OK;          // on the context's stream like everything else that touches the set until its replay
}

int uvrt_replay_batch(uvrt_ctx* c, const uvrt_replay_op* ops, int32_t count, int32_t tri_count)
{
    if (!c || !ops || c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: no traced batch");
    if (count != c->b_count) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: %d operations for a batch of %d launches", count, c->b_count);
    if (tri_count < 0 || tri_count > c->T) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: bad tri_count");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    ReplayParams p;
    memset(&p, 0, sizeof p);
    p.photon_map = c->photon_map.as<double>();
    p.max_map = c->max_map.as<double>();
    p.planes = c->bs[c->b_set].planes.as<int32_t>();
    p.folded = c->bs[c->b_set].folded.as<int32_t>();
    p.dosage = c->dosage.as<float>();
    p.color = c->color.as<float>();
    p.area = c->area.as<float>();
    p.plane_stride = (int64_t)c->b_repl * c->T;
    p.replicas = c->b_repl;
    p.T = tri_count;
    p.count = count;
    p.is_folded = c->b_is_folded ? 1 : 0;
    for (int k = 0; k < count; ++k) {
        if (ops[k].which_map != UVRT_MAP_SUM && ops[k].which_map != UVRT_MAP_MAX)
            return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: which_map must be 0 or 1");
        p.ops[k].plane = c->b_phys[k];
        p.ops[k].duration = ops[k].duration;
        p.ops[k].shade = ops[k].shade;
        p.ops[k].which_map = ops[k].which_map;
        p.ops[k].photons_per_light = ops[k].photons_per_light;
        p.ops[k].scaled_power = ops[k].scaled_power;
        p.ops[k].min_value = ops[k].min_value;
        p.ops[k].threshold_view = ops[k].threshold_view;
    }
    launch_replay_batch(p, c->stream);
    HIP_TRY(hipGetLastError());
    if (tri_count < c->T) {     // a partial replay (calibration's 2-triangle scene never does this): clear the rest
        if (c->b_is_folded) HIP_TRY(hipMemsetAsync(c->bs[c->b_set].folded.p, 0, c->bs[c->b_set].folded.bytes, c->stream));
        else HIP_TRY(hipMemsetAsync(c->bs[c->b_set].planes.p, 0, c->bs[c->b_set].planes.bytes, c->stream));
    }
    HIP_TRY(hipEventRecord(c->bs[c->b_set].free_ev, c->stream));    // the set may be traced into again
    c->b_count = 0;
    c->b_is_folded = false;
    // later accumulate / Shade work waits for this replay; the next batch's generate / extend do not
    return mark_map_fence(c);
}

int uvrt_read_batch_counts(uvrt_ctx* c, int32_t launch, int32_t* out, int32_t first, int32_t count)
{
    if (!c || c->b_count <= 0 || launch < 0 || launch >= c->b_count)
        return fail(UVRT_ERR_INVALID, "uvrt_read_batch_counts: no such launch in the traced batch");
    if (int rc = uvrt_fold_batch(c)) return rc;
    if (!out || first < 0 || count < 0 || first + count > c->T) return fail(UVRT_ERR_INVALID, "uvrt_read_batch_counts: bad range");
    if (count == 0) return UVRT_OK;
    HIP_TRY(hipMemcpyAsync(out, c->bs[c->b_set].folded.as<int32_t>() + (size_t)c->b_phys[launch] * c->T + first, (size_t)count * 4,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

}  // extern "C"
