/*
 * uvrt.h -- C ABI of the MI355X (gfx950) UV-dose hot path.
 *
 * This is the drop-in boundary: the entry points below are what the reference's
 * `RayTracer` (raytracer.h:13-59, raytracer.cpp) needs in place of its OpenCL wrapper
 * (`Kernel`/`Buffer`, template/precomp.h:1239-1325).  Plain pointers and sizes only; every
 * call returns UVRT_OK (0) or a negative error code, and uvrt_last_error() returns the text.
 * A context is bound to one HIP device and one in-order HIP stream (the reference's single
 * in-order cl_command_queue, template/template.cpp:1446); calls are asynchronous unless they
 * read back to the host.  Not thread-safe per context (the reference is single-threaded).
 *
 * Record layouts handed over by the host are the reference's own:
 *   Tri      64 B  v0.xyz,pad,v1.xyz,pad,v2.xyz,pad,centroid.xyz,pad   (mesh.h:6-13, cl/tools.cl:31-37)
 *   BVHNode  32 B  min.xyz,leftFirst,max.xyz,triCount                  (bvh.h:11-21, cl/tools.cl:39-45)
 *   triIdx   u32[T]                                                    (bvh.h:45)
 *   Ray      32 B  dir.xyz,orig.xyz,dist,triID                         (cl/tools.cl:8-14)
 * Inside the context they are re-laid-out for the GPU.
 */
#ifndef UVRT_H
#define UVRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvrt_ctx uvrt_ctx;

enum {
    UVRT_OK = 0,
    UVRT_ERR_INVALID = -1,   /* bad argument / call order */
    UVRT_ERR_HIP = -2,       /* a HIP runtime call failed */
    UVRT_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    UVRT_ERR_BVH = -4,       /* malformed BVH (cycle, out-of-range child, leaf beyond triIdx) */
    UVRT_ERR_STACK = -5      /* traversal needed more than 32 stack entries (extend.cl:43) */
};

/* which per-triangle map computeDosage reads (raytracer.cpp:96-116) */
enum { UVRT_MAP_SUM = 0, UVRT_MAP_MAX = 1 };

const char* uvrt_last_error(void);
const char* uvrt_version(void);

/* Kernel::InitCL + the six `new Kernel(...)` of RayTracer::Init (myapp.cpp:21,
 * raytracer.cpp:17-22).  A new context starts with SEED = 0, like a freshly built
 * generate.cl program (generate.cl:6). */
int uvrt_create(int device_id, uvrt_ctx** out);
void uvrt_destroy(uvrt_ctx* ctx);

/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the context's
 * own.  NULL restores the context's own stream. */
int uvrt_set_stream(uvrt_ctx* ctx, void* hip_stream);

/* verticesBuffer / bvhNodesBuffer / triIdxBuffer upload (raytracer.cpp:24-30); callable again
 * at any time (CalibratePower swaps scenes, raytracer.cpp:166-187,212-224).  The arrays are
 * copied; the caller keeps ownership.  (Re)allocates and zeroes the per-triangle maps
 * (raytracer.cpp:32-35) when tri_count changes. */
int uvrt_set_scene(uvrt_ctx* ctx, const void* tris64, int32_t tri_count,
                   const void* nodes32, int32_t node_count, const uint32_t* tri_idx);

/* rayBuffer = new Buffer(32 * photonCount) (raytracer.cpp:136-139); size_t arithmetic, so the
 * reference's int overflow at 2^26 photons does not exist here. */
int uvrt_resize_rays(uvrt_ctx* ctx, int64_t photon_count);

/* cl/reset.cl:4-26 over the current scene's triangles (raytracer.cpp:141-142) */
int uvrt_reset(uvrt_ctx* ctx, int32_t reset_color);

/* cl/generate.cl:8-40 for global ids [first_gid, first_gid+n) of a launch of any size, under
 * the pinned SEED semantics of SURVEY.md 8c: work-item 0 reads SEED_{k-1}, every other
 * work-item reads SEED_k; the context's SEED advances to SEED_k on every call (SEED_k is a
 * function of light_pos and SEED_{k-1} only, so ranks that generate disjoint gid ranges of the
 * same launch stay in step).  n <= the capacity set by uvrt_resize_rays. */
int uvrt_generate(uvrt_ctx* ctx, const float light_pos[3], float light_length,
                  int64_t first_gid, int64_t n);

/* cl/extend.cl:85-99 over the n rays of the last uvrt_generate: closest hit through the BVH,
 * then one increment of tempPhotonMap[triID] per hit. */
int uvrt_extend(uvrt_ctx* ctx, int64_t n);

/* cl/accumulate.cl:4-14 over tri_count triangles.  (A full-range accumulate is enqueued with the next call: a uvrt_shade
 * right behind it -- the host loop's order, myapp.cpp:159-160 -- runs both in one kernel; same arithmetic, same order.) */
int uvrt_accumulate(uvrt_ctx* ctx, float time_step, int32_t tri_count);

/* cl/shade.cl:23-41 (computeDosage) over tri_count triangles */
int uvrt_compute_dosage(uvrt_ctx* ctx, int32_t which_map, int32_t photons_per_light,
                        float scaled_power, int32_t tri_count);

/* cl/shade.cl:43-71 (dosageToColor) into the context's 9-float/triangle colour buffer (the
 * reference writes a GL VBO, raytracer.cpp:37,119) */
int uvrt_dosage_to_color(uvrt_ctx* ctx, float min_value, int32_t threshold_view,
                         int32_t tri_count);

/* RayTracer::Shade's kernel pair in one launch (raytracer.cpp:96-118): computeDosage followed by
 * dosageToColor over tri_count triangles, same arithmetic, the dose buffer is written as well */
int uvrt_shade(uvrt_ctx* ctx, int32_t which_map, int32_t photons_per_light, float scaled_power,
               float min_value, int32_t threshold_view, int32_t tri_count);

/* clFinish(Kernel::GetQueue()) (myapp.cpp:165, raytracer.cpp:202); also reports a traversal
 * stack overflow raised by any extend since the last sync. */
int uvrt_sync(uvrt_ctx* ctx);

/* dosageBuffer->CopyFromDevice() (raytracer.cpp:204-207), any range; synchronises. */
int uvrt_read_dosage(uvrt_ctx* ctx, float* out, int32_t first, int32_t count);
int uvrt_read_color(uvrt_ctx* ctx, float* out9, int32_t first, int32_t count);

/* ---- program-scope SEED of generate.cl (generate.cl:6,39) ---- */
int uvrt_get_seed(uvrt_ctx* ctx, uint32_t* seed);
int uvrt_set_seed(uvrt_ctx* ctx, uint32_t seed);
/* SEED_k from SEED_{k-1} and the lamp position, without launching (host-side RNG walk of
 * work-item 0); used to give every rank of a sharded job its place in the global launch
 * order. */
uint32_t uvrt_seed_next(const float light_pos[3], float light_length, uint32_t seed_prev);

/* Advance the context's SEED as a uvrt_generate at this lamp would, without launching (a rank of a
 * sharded job skipping a launch that another rank traces). */
int uvrt_advance_seed(uvrt_ctx* ctx, const float light_pos[3], float light_length);

/* Which outcome of generate.cl's SEED race (generate.cl:6,13,39: every work-item reads the
 * program-scope SEED, work-item 0 overwrites it at its end) and of its float -> uint conversion of a
 * negative seed sum (undefined in OpenCL C) the context reproduces:
 *   0 (default) = the canonical semantics of SURVEY.md 8c: serial work-item order -- work-item 0 reads
 *     SEED_{k-1}, every other work-item reads SEED_k -- and the conversion through int64 (x86-64);
 *   1 = "gfx950-ocl": what the reference's generate.cl, compiled unmodified by ROCm's OpenCL compiler,
 *     does on this GPU (measured, tests/test_gpu_reference_kernels.py): SEED is fetched through the
 *     scalar cache, so EVERY work-item of launch k reads SEED_{k-1}; v_cvt_u32_f32 turns a negative sum
 *     into 0.  With uvrt_set_flavour(ctx, 1) the whole pipeline then reproduces the reference's own
 *     kernel chain running live on the MI355X: counts bit for bit, dose within 1e-4. */
int uvrt_set_seed_mode(uvrt_ctx* ctx, int32_t mode);
uint32_t uvrt_seed_next_mode(const float light_pos[3], float light_length, uint32_t seed_prev,
                             int32_t seed_mode);

/* ---- batched tracing: several launches in one go, one count "plane" per launch ----
 *
 * A computation is iterations x lamps launches of generate -> extend -> accumulate (raytracer.cpp:66-88).
 * accumulate.cl:9-13 needs every launch's TOTAL per-triangle count (max), so a job that shards launches by
 * global-id range over GPUs keeps one int32[T] plane per launch, sums all planes over the ranks ONCE per
 * batch and then replays accumulate (and the host loop's Shade calls) plane by plane: the f64 maps, the dose
 * and the colours come out bit-identical to the per-launch sequence.  On one GPU the same calls trace the
 * launches of a batch in as few kernel launches as their lamps allow (launches of one lamp column share their
 * per-launch records), which keeps the persistent wavefronts supplied with rays across launch boundaries. */
typedef struct {
    float duration;              /* accumulate.cl timeStep of this launch (raytracer.cpp:84) */
    int32_t shade;               /* != 0: the host loop runs Shade after this launch (myapp.cpp:160) ... */
    int32_t which_map;           /* ... with these arguments (raytracer.cpp:96-118; UVRT_MAP_SUM / UVRT_MAP_MAX) */
    int32_t photons_per_light;
    float scaled_power, min_value;
    int32_t threshold_view;
} uvrt_replay_op;

/* generate + extend for `count` launches (<= 64), launch k from lamp lamps[3k..3k+2], all over the global
 * ids [first_gid, first_gid + n).  The SEED chain advances launch by launch exactly as `count` calls of
 * uvrt_generate would.  The deposits stay in the batch's planes until uvrt_replay_batch. */
int uvrt_trace_batch(uvrt_ctx* ctx, const float* lamps, float light_length, int32_t count,
                     int64_t first_gid, int64_t n);
/* per launch (logical order): accumulate.cl, then Shade where ops[k].shade is set; ends the batch */
int uvrt_replay_batch(uvrt_ctx* ctx, const uvrt_replay_op* ops, int32_t count, int32_t tri_count);
/* sum the deposit replicas of every plane into the int32[count][T] array the reduction works on */
int uvrt_fold_batch(uvrt_ctx* ctx);
/* test hook: tempPhotonMap of launch `launch` of the traced batch (folds; after a reduce: the global counts) */
int uvrt_read_batch_counts(uvrt_ctx* ctx, int32_t launch, int32_t* out, int32_t first, int32_t count);

/* ---- the one collective of a sharded computation (RCCL over xGMI; librccl is opened on first use) ----
 * One process per GPU: rank 0 calls uvrt_comm_unique_id and hands the 128 bytes to every rank (any
 * out-of-band channel), every rank calls uvrt_comm_init_rank; then per batch uvrt_trace_batch (its own
 * global-id range) -> uvrt_reduce_batch (int32 SUM all-reduce of the planes on the context's stream) ->
 * uvrt_replay_batch.  One process driving several GPUs: uvrt_comm_init_all over one context per device and
 * uvrt_reduce_batch_group.  uvrt_reduce_batch_group without communicators sums contexts that share ONE
 * device (rehearsals and tests of the sharded path on a single-GPU box). */
/* 1 when librccl opens and every entry point used here resolves, else 0 (text in uvrt_last_error): a LOCAL
 * precondition.  uvrt_comm_init_rank is itself a collective (ncclCommInitRank): ranks agree on this flag over their
 * out-of-band channel first, so that a rank without RCCL cannot leave the others waiting inside the init. */
int uvrt_comm_available(void);
int uvrt_comm_unique_id(void* id128);
int uvrt_comm_init_rank(uvrt_ctx* ctx, const void* id128, int32_t rank, int32_t world);
int uvrt_comm_init_all(uvrt_ctx** ctxs, int32_t n);
int uvrt_comm_destroy(uvrt_ctx* ctx);
/* what the context's communicator is: out4 = { world given at init (0: none), rank, ncclCommCount of the native
 * communicator (what RCCL itself spans; 0: none), compute units the launch lanes leave to the collective } */
int uvrt_comm_info(uvrt_ctx* ctx, int32_t out4[4]);
int uvrt_reduce_batch(uvrt_ctx* ctx);
int uvrt_reduce_batch_group(uvrt_ctx** ctxs, int32_t n);

/* ---- tuning knobs (results never depend on them) ---- */
/* bits of the ray-coherence key used to order rays before extend; 0 = trace in gid order
 * (default), -1 = choose from n (about one wavefront of rays per key). */
int uvrt_set_sort_bits(uvrt_ctx* ctx, int32_t bits);
/* record (dist, triID) per ray in gid order during extend (the reference updates rays in
 * place, extend.cl:90-92); off by default, needed by uvrt_read_rays. */
int uvrt_set_record_hits(uvrt_ctx* ctx, int32_t on);
/* arithmetic flavour of extend (cl/extend.cl:6-38):
 *   0 (default) = the canonical strict flavour of SURVEY.md 8c: every operator one IEEE rounding in source order,
 *     correctly rounded divisions -- bit-exact against a CPU restatement unconditionally;
 *   1 = "ocl-amd": IntersectTri's cross()/dot() in the fused multiply-add forms ROCm's OpenCL device library gives the
 *     reference's extend.cl on gfx950 (strict build) -- results equal that kernel's, run
 *     live on the same GPU, bit for bit; slab test, traversal and deposit as flavour 0;
 *   2 = "shipped flags" (OPT-IN): the arithmetic the reference's OWN clBuildProgram options (-cl-fast-relaxed-math
 *     -cl-mad-enable -cl-single-precision-constant, template/template.cpp:1192) make of extend.cl on gfx950, read
 *     off the disassembly of that build: IntersectAABB's t = (b - o) * v_rcp_f32(d) (one multiply by the hardware's
 *     approximate reciprocal instead of a division, cl/extend.cl:31-35), IntersectTri as flavour 1 with
 *     f = v_rcp_f32(a) (extend.cl:17).  (dist bits, triID, counts) equal that kernel's bit for bit; against
 *     flavours 0 / 1 a few rays per million land on a neighbouring triangle (the dose stays within 1e-4).  v_rcp_f32
 *     is specific to the GPU generation: a CPU restatement reproduces this flavour only with the instruction's table
 *     read from the device.  Rays whose direction
 *     has all three components zero or NaN are outside this flavour's parity domain. */
int uvrt_set_flavour(uvrt_ctx* ctx, int32_t flavour);
/* OPT-IN 4-wide traversal: uvrt_extend walks a one-level collapse of the caller's BVH
 * (a node holds its grandchildren's boxes) -- about half the loop trips per ray, the same box and triangle
 * arithmetic, but NOT the reference's visit order: `dist` and `triID` equal the default kernel's except on
 * rays where two accepted hits tie exactly in t or a box is culled by an almost equally distant earlier hit
 * (extend.cl:25,66-76 make those order-dependent; none in 25 M rays on the test room).  Off by default: the
 * default walk is bit-exact unconditionally.  uvrt_trace_batch always uses the default walk. */
int uvrt_set_wide_bvh(uvrt_ctx* ctx, int32_t on);
/* Which node-pair records the traversal serves from LDS: 1 (default) = the 175 records the lamp's photons
 * visit most, found on the device from a sample of the launch's own rays the first time a lamp position is
 * seen (62-70 % of all inner-node visits on the test room); 0 = the first levels of the tree in breadth-first order
 * (31-40 %).  Only the order of records in memory changes; results never depend on it. */
int uvrt_set_hot_records(uvrt_ctx* ctx, int32_t mode);
/* Renumber the node-pair records of the default extend kernel: record i (breadth-first index of the
 * inner node, as uvrt_set_scene lays them out) moves to perm[i]; the first 175 records of the new
 * numbering are served from LDS.  Results do not depend on it.  NULL restores the breadth-first
 * order.  Reset by uvrt_set_scene. */
int uvrt_set_record_perm(uvrt_ctx* ctx, const uint32_t* perm, int32_t n);
/* test hook: the renumbering the launch of the last uvrt_generate uses (the caller's own, the automatic hot-record
 * one, or the identity): out[i] = index of breadth-first record i, n = number of inner nodes; synchronises. */
int uvrt_read_record_perm(uvrt_ctx* ctx, uint32_t* out, int32_t n);

/* Launch pipelining (on by default): consecutive launches (generate, extend, accumulate and the Shade
 * that follows) alternate between the context's stream and an internal second stream with their own
 * ray / count buffers, so the next launch starts in the wave slots the previous one frees while its
 * last rays finish.  The per-triangle maps are still updated in launch order and every other entry
 * point first orders the context's stream after all outstanding work, so callers see the in-order
 * behaviour of the reference's single command queue.  0 = everything on the one stream.  (The
 * tempPhotonMap pointer of uvrt_device_ptr(ctx, 2, ...) alternates with the launch lane: callers
 * that cache it across launches must switch the pipelining off.) */
int uvrt_set_pipeline(uvrt_ctx* ctx, int32_t on);

/* extend kernel knobs (developer / A-B; every setting is bit-exact): 0 = default; 400-499 = leaf period /
 * LDS top cache code + 10 * grid code with refill at 16 idle lanes; 500-599 = the same with IEEE divisions
 * everywhere; 600-1299 = like 400-499 with the refill threshold 8 / 24 / 4 / 32 / 40 / 48 / 56 idle lanes.  See DESIGN.md 4. */
int uvrt_set_variant(uvrt_ctx* ctx, int32_t variant);

/* ---- test / interop hooks ---- */
/* the rays of the last generate (+ extend, if hits were recorded) in the reference's 32-byte
 * Ray layout and gid order; synchronises. */
int uvrt_read_rays(uvrt_ctx* ctx, void* rays32, int64_t first, int64_t count);
/* test hook: replace the rays of the "last generate" by n host records in the reference's 32-byte
 * Ray layout (dir, orig; dist/triID ignored).  All records must share orig.x and orig.z (rays of
 * one lamp, generate.cl:16).  SEED is untouched.  Lets tests feed adversarial rays (zero
 * direction components, origins on box planes) straight into uvrt_extend. */
int uvrt_write_rays(uvrt_ctx* ctx, const void* rays32, int64_t n);
int uvrt_read_counts(uvrt_ctx* ctx, int32_t* out, int32_t first, int32_t count);
int uvrt_read_photon_map(uvrt_ctx* ctx, int32_t which_map, double* out, int32_t first,
                         int32_t count);
/* raw device pointers of the per-triangle arrays, for zero-copy wrapping (e.g. as torch
 * tensors handed to an RCCL collective).  which: 0 photonMap f64[T], 1 maxPhotonMap f64[T],
 * 2 tempPhotonMap i32[T], 3 dosageMap f32[T], 4 colour f32[9T], 5 the folded planes i32[launches][T] of the
 * traced batch (for a caller that brings its own collective; uvrt_reduce_batch is the native one).
 * The call is also the ordering point for external work on these arrays: it orders the context's
 * stream after every outstanding launch (with launch pipelining some sit on the library's second
 * stream) and makes the library's next work on the arrays wait for whatever the caller enqueues on
 * the context's stream before its next uvrt call.  Call it before EVERY external use, not once. */
int uvrt_device_ptr(uvrt_ctx* ctx, int32_t which, void** ptr, int64_t* bytes);
/* copy one of those arrays to (to_ctx = 0) or from (to_ctx = 1) an external device buffer of the
 * same size, on the context's stream (staging for collectives when zero-copy wrapping is not
 * available). */
int uvrt_copy_device(uvrt_ctx* ctx, int32_t which, void* ext_dev_ptr, int32_t to_ctx);
/* time in ms the device spent in the extend kernels since the last call (HIP events on the
 * context's stream), and the number of extend launches; synchronises. */
int uvrt_extend_time_ms(uvrt_ctx* ctx, double* ms, int64_t* launches);
int uvrt_set_timing(uvrt_ctx* ctx, int32_t on);
/* measurement hook: the shader clock UNDER LOAD.  _start enqueues a one-wave kernel on a stream of its own that reads the
 * shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) `microseconds` apart, beside whatever the
 * context's streams are doing; _read waits for it and returns ticks ratio x 100 MHz.  bench.py prices the issue-rate peaks of its
 * roofline with the clock measured during its own steps instead of the nominal 2.4 GHz. */
int uvrt_clock_probe_start(uvrt_ctx* ctx, int32_t microseconds);
int uvrt_clock_probe_read(uvrt_ctx* ctx, double* shader_mhz);
/* compute units of the context's device (the persistent extend grid is 8 workgroups per CU) */
int uvrt_device_cus(uvrt_ctx* ctx);
/* HIP devices visible to the process (0 when there is none) */
int uvrt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
