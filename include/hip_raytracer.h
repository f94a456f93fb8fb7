/*
 * hip_raytracer.h - C ABI of the MI355X (gfx950) backend for the reference's IRaytracer hot path.
 *
 * This is the drop-in boundary: a `HIPRaytracer : IRaytracer` (C++: opencl-raytracer_amd/host/HIPRaytracer.hpp,
 * Python: opencl-raytracer_amd/hip_raytracer.py) replaces the reference's `OpenCLRaytracer` and calls only
 * these entry points. Plain pointers and sizes; no C++/torch/HIP types in any signature.
 *
 * Reference interfaces replaced (citations into the reference tree):
 *   rt_create          <- OpenCLRaytracer::OpenCLRaytracer(objects, lights, rays, MAX_BOUNCES)
 *                         OpenCLRaytracer.cpp:13-74 (AoS conversion :16-33, buffers :47-50, program build :53-59,
 *                         kernel args :62-68, uploads :70-73)
 *   rt_render          <- OpenCLRaytracer::Render()  OpenCLRaytracer.cpp:80-105 (enqueue_1d_range_kernel :89-91,
 *                         blocking enqueue_read_buffer :94, returns the object-owned host buffer :104);
 *                         declared by IRaytracer::Render()  IRaytracer.hpp:13
 *   rt_render_device   <- the same launch without the read-back (device-resident framebuffer; used for
 *                         multi-GPU gathers and for timing with inputs/outputs resident in HBM)
 *   rt_destroy         <- OpenCLRaytracer::~OpenCLRaytracer()  OpenCLRaytracer.cpp:76-78
 *   rt_set_camera      <- main()'s primary-ray loop  OpenCL-Raytracer.cpp:18-26,68-72 (rays regenerated in-kernel)
 *   kernel selector    <- `__kernel hittest`  hittest_kernel.cl:54, `__kernel shade`  shade_kernel.cl:180,
 *                         `__kernel shade_and_reflect`  shade_and_reflect_kernel.cl:244
 *
 * Record layouts are the reference's device structs, byte for byte (shade_and_reflect_kernel.cl:1-29,
 * OpenCLRaytracer.hpp:25-58): ObjectData 320 B, Light 64 B, Ray 32 B, pixel float4 16 B. See rt_records.h.
 *
 * Error behaviour: the reference has no error codes (Boost.Compute throws). Every function here returns
 * RT_OK (0) or a negative rt_status and records a message retrievable with rt_last_error(); nothing throws
 * across the boundary. There is no CPU fallback: without a usable HIP device rt_create fails with
 * RT_ERR_NO_DEVICE.
 *
 * Threading: like the reference (single in-order queue, OpenCLRaytracer.cpp:44) a context is not
 * re-entrant; use one context per thread / per GPU.
 */
#ifndef HIP_RAYTRACER_H
#define HIP_RAYTRACER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version history:
 *   1  round 1-2.
 *   2  rt_render_device(ctx, out, NULL) means the LEGACY DEFAULT stream (it was the context's private stream in early
 *      builds of version 1); rt_get_setup_times; the multi-device entry points rt_create_multi / rt_render_multi /
 *      rt_render_multi_device / rt_multi_context / rt_destroy_multi. A caller built against version 1 keeps working:
 *      no existing signature or struct changed.
 *   3  rt_get_stats_multi / rt_count_rays_multi (the counters of every shard, summed); rt_render_multi places every device's
 *      tiles straight in the pinned host frame (no hop through devices[0]); rt_create / rt_set_camera switch a frame whose
 *      primary directions leave the default path's domain (|d|^2 == 0, < 1e-30, > 1e30) to RT_FLAG_LITERAL by themselves.
 *      Additions only: a caller built against version 2 keeps working. */
#define RT_ABI_VERSION 3

typedef struct rt_context rt_context;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,
    RT_ERR_OUT_OF_MEMORY = -3,
    RT_ERR_HIP = -4,
    RT_ERR_STATE = -5
} rt_status;

typedef enum rt_kernel {
    RT_KERNEL_HITTEST = 0,           /* nearest t per ray          (hittest_kernel.cl:54)            */
    RT_KERNEL_SHADE = 1,             /* direct lighting, summed    (shade_kernel.cl:180)             */
    RT_KERNEL_SHADE_AND_REFLECT = 2  /* + iterative reflection     (shade_and_reflect_kernel.cl:244) */
} rt_kernel;

/* rt_create flags */
#define RT_FLAG_UNFUSED   0x1u /* arithmetic of an OpenCL device WITHOUT fma contraction (x86 baseline); default
                                  is the contraction the OpenCL front-end marks (llvm.fmuladd -> fma)          */
#define RT_FLAG_LITERAL   0x2u /* trace every ray the reference traces (no exact eliminations: any-hit shadow
                                  early-out, backward light scan, dead reflection ray). Results are identical.
                                  rt_create sets it by itself for a scene with a DEGENERATE instance (non-finite or
                                  singular mvInverse): its NaN hit times make the reference's result depend on the
                                  order of the object loop, which only the literal loops reproduce - and for a frame
                                  whose PRIMARY rays leave the domain of the exact eliminations: any direction with
                                  |d|^2 == 0, < 1e-30 or > 1e30 (or not finite) in the uploaded buffer, or a pinhole
                                  camera (rt_set_camera) that produces one. The reference's tests give such a ray a
                                  NaN time on EVERY object (shade_and_reflect_kernel.cl:85-108). Domain of the default
                                  (grid) path therefore: finite rays with start.w = 1, direction.w = 0,
                                  1e-30 < |d|^2 < 1e30, affine instances; everything else is rendered by the
                                  brute-force or literal loops, never approximated                               */
#define RT_FLAG_NO_RAYGEN 0x4u /* never replace an uploaded pinhole ray grid by in-kernel generation            */
#define RT_FLAG_WAVEFRONT  0x8u  /* force the large-scene path (separate traversal / shading kernels)          */
#define RT_FLAG_NO_GRID    0x20u /* large-scene path: test every object for every ray (no conservative grid culling);
                                    results are identical, this is the brute-force baseline                  */
#define RT_FLAG_MONOLITHIC 0x10u /* force the small-scene path (one fused kernel per frame); default: chosen by
                                    object count. Both paths produce identical bits.                          */

#define RT_FLAG_FAST_PHONG 0x40u /* opt-in: values that feed COLOUR only (the shading normal, the view vector, the reflected
                                    light vector, the specular power) on the hardware's fast reciprocal-square-root / log2 /
                                    exp2 paths instead of IEEE sqrt + divisions and the library powf. Every ray - hit index, t,
                                    shadow and reflection rays, the reference-equivalent ray count - stays bit-exact; colours
                                    stay within the 1e-5 the reference comparison allows (north_star), not bit-identical to
                                    the default arithmetic.                                                              */

typedef struct rt_stats_t {
    uint64_t rays_traced;     /* rays this backend actually issued in the last counted render (R_act)          */
    uint64_t rays_reference;  /* rays the reference semantics trace for the same frame (R_ref)                 */
    uint64_t hit_pixels;      /* work-items whose primary ray hit something                                    */
    float    last_kernel_ms;  /* device time of the last render's kernel(s), HIP events on the render stream   */
    uint32_t pinhole;         /* 1 if primary rays are generated in-kernel                                     */
    uint32_t width, height;   /* pinhole grid (0 when rays come from the uploaded buffer)                      */
    uint64_t local_rays;      /* work-items this context renders (after rt_set_shard)                          */
    uint32_t wavefront;       /* 1 if the last render used the large-scene (wavefront) path                    */
    uint32_t rounds;          /* trace/resume rounds of the last wavefront render                              */
    uint64_t object_tests;    /* ray-object tests executed by the traversal kernels in the last counted render
                                 (large-scene path only; 0 otherwise)                                           */
} rt_stats_t;

/* Build a raytracer for one GPU.
 *   objs   : n_objs  x 320-byte ObjectData records (may be NULL when n_objs == 0)
 *   lights : n_lights x 64-byte Light records      (may be NULL when n_lights == 0)
 *   rays   : n_rays  x 32-byte Ray records in work-item order, or NULL when rt_set_camera() will
 *            describe an n_rays = width*height pinhole grid
 *   max_bounces : MAX_BOUNCES of shade_and_reflect (ignored by the other kernels, as in the reference)
 *   device : HIP device ordinal
 * Host buffers are copied; the caller may free them after the call returns. */
int rt_create(rt_context** ctx, const void* objs, uint32_t n_objs, const void* lights, uint32_t n_lights,
              const void* rays, uint64_t n_rays, uint32_t max_bounces, int kernel, int device, uint32_t flags);

/* Primary rays = the reference's pinhole grid, generated in-kernel (bit-exact):
 * start (0,0,0,1), direction (i - W/2, (H - j) - H/2, z, 0) for work-item j*W + i. width*height must equal n_rays. */
int rt_set_camera(rt_context* ctx, uint32_t width, uint32_t height, float z);

/* Multi-GPU partition: work-items are cut into tiles of `tile_rays` consecutive rays (a row-tile is
 * tile_rows*width rays); tile j belongs to rank j % world. After this call the context renders only its
 * own tiles, packed back to back in its output buffer (rt_local_rays() work-items). */
int rt_set_shard(rt_context* ctx, uint64_t tile_rays, uint32_t rank, uint32_t world);
uint64_t rt_local_rays(const rt_context* ctx);

/* Synchronous render into a context-owned host buffer, like IRaytracer::Render():
 *   kernels 1,2: rt_local_rays() x float4; RGB in [0..2]; misses are (0,0,0,1) (the reference's upload-time
 *                background, OpenCLRaytracer.cpp:32), hit pixels have w = 1
 *   kernel 0   : rt_local_rays() x float; misses are 3.402823466e+38f
 * The buffer is overwritten by the next call and freed by rt_destroy.
 * A large frame of the large-scene path (>= 4 M rays, not sharded by the caller) is rendered in two passes over interleaved
 * 16-row tiles - three tiles of every four, then the fourth - the first pass's read-back running while the second renders: same
 * pixels, three quarters of the blocking copy (OpenCLRaytracer.cpp:94) hidden. RT_RENDER_PASSES=1 in the environment keeps it
 * to one pass, RT_RENDER_SPLIT="a,b[,c[,d]]" chooses another split (tiles per pass out of every a + b + ..). */
int rt_render(rt_context* ctx, const float** out);

/* Render into caller-provided DEVICE memory (same element layout) on a caller-provided HIP stream (hipStream_t passed
 * as void*). NULL is the LEGACY DEFAULT stream - what a host framework's "current stream" handle is when it is on its
 * default stream (torch: cuda_stream == 0) - so the frame is always ordered with the caller's own work on that stream;
 * it is never a private stream of the context. The small-scene path does not synchronise the host. The large-scene
 * path keeps its round loop on the device (queue lengths never travel to the host inside a round) and synchronises
 * the stream once per batch of rounds - once per frame in the normal case. The calling thread's current device is
 * left as it was. */
int rt_render_device(rt_context* ctx, void* d_out, void* hip_stream);

/* Optional per-work-item primary-hit record of the NEXT render: t (float) and winning object index
 * (int32, -1 on a miss) into caller-provided DEVICE buffers of rt_local_rays() elements (either may be NULL). */
int rt_set_aux_device(rt_context* ctx, void* d_hit_t, void* d_hit_index);
/* Host-side convenience: render once and copy t / index to host arrays (either may be NULL). */
int rt_render_aux(rt_context* ctx, float* hit_t, int32_t* hit_index);

/* Counted render (untimed instrumentation pass): fills rays_traced / rays_reference / hit_pixels. */
int rt_count_rays(rt_context* ctx);
int rt_get_stats(rt_context* ctx, rt_stats_t* stats);

/* Device-time bookkeeping: every render records a HIP event pair around its kernel(s) on the stream it was
 * launched on (up to 256 launches are kept). rt_timing_summary waits for them and returns the summed device
 * time and the number of launches since rt_timing_reset. */
int rt_timing_reset(rt_context* ctx);
int rt_timing_summary(rt_context* ctx, double* sum_ms, uint32_t* launches);

/* One-time host-side work that sits OUTSIDE every render timer (the reference's equivalent is its constructor,
 * OpenCLRaytracer.cpp:13-74): milliseconds of wall clock spent in rt_create and, for the per-camera screen tiles and the
 * large-scene path's state buffers, in the first render. */
typedef struct rt_setup_times_t {
    double create_ms;        /* rt_create as a whole (incl. the HIP runtime's start-up when rt_create is the process's
                                first HIP call: ~140 ms that belong to no phase below)                          */
    double upload_ms;        /* record re-pack + uploads (objects, lights, rays)                               */
    double grid_ms;          /* conservative grid + the record table of the unified walk (0 for small scenes)  */
    double blocks_ms;        /* the closest-hit walk's coarse grid of 32-byte blocks                           */
    double light_tiles_ms;   /* light tiles of the last positional light                                       */
    double screen_tiles_ms;  /* per-camera screen tiles (first render after rt_create / rt_set_camera)         */
    double buffers_ms;       /* large-scene path: pixel state + queues allocation (first render)               */
} rt_setup_times_t;
int rt_get_setup_times(rt_context* ctx, rt_setup_times_t* times);

void rt_destroy(rt_context* ctx);

/* ---- several GPUs from one process ------------------------------------------------------------------------------------
 * `new HIPRaytracer(objects, lights, rays, MAX_BOUNCES)` on a multi-GPU node (north_star: "row-tiles across the 8 GPUs of
 * one node"; the reference drives exactly one device, OpenCLRaytracer.cpp:36-44). rt_create_multi builds one context per
 * entry of `devices` (the same ordinal may appear more than once: a rehearsal on fewer GPUs), each with
 * rt_set_shard(tile_rays, r, n_devices): interleaved tiles of `tile_rays` consecutive rays (a row-tile = tile_rows * width).
 * A render runs every context on a host thread of its own, on its own device and stream; each context's packed tiles are
 * then copied device-to-device (peer access where the runtime grants it) straight to their final offsets in the frame held
 * on devices[0] - the gather of SURVEY.md 8e without a second process or a collective library.
 *   rt_render_multi         synchronous, like rt_render: the whole frame (n_rays elements) in a host buffer owned by `m`
 *   rt_render_multi_device  the whole frame into caller-provided memory ON devices[0] (n_rays elements, padded up to whole
 *                           tiles: rt_multi_frame_elems()); returns when the frame is complete. The shards write d_frame
 *                           from their own streams: the buffer must be IDLE on entry (no pending work of the caller on it -
 *                           synchronise the stream that last touched it first)
 *   rt_multi_context        the r-th context (rt_count_rays / rt_get_stats / rt_timing_* per shard)
 *   rt_count_rays_multi     the untimed counted render on every shard
 *   rt_get_stats_multi      rays_traced / rays_reference / hit_pixels / object_tests / local_rays summed over the shards,
 *                           last_kernel_ms and rounds of the slowest one (the whole frame's figures, like Render()'s frame)
 * rt_render_multi: every device copies ITS tiles from its own memory into the pinned (portable) host frame, on its own
 * stream and PCIe link (one strided device-to-host copy per shard); nothing is gathered on devices[0]. */
typedef struct rt_multi rt_multi;
int rt_create_multi(rt_multi** m, const void* objs, uint32_t n_objs, const void* lights, uint32_t n_lights,
                    const void* rays, uint64_t n_rays, uint32_t max_bounces, int kernel,
                    const int* devices, uint32_t n_devices, uint64_t tile_rays, uint32_t flags);
int rt_set_camera_multi(rt_multi* m, uint32_t width, uint32_t height, float z);
uint64_t rt_multi_frame_elems(const rt_multi* m);
int rt_render_multi(rt_multi* m, const float** out);
int rt_render_multi_device(rt_multi* m, void* d_frame);
int rt_count_rays_multi(rt_multi* m);
int rt_get_stats_multi(rt_multi* m, rt_stats_t* stats);
rt_context* rt_multi_context(rt_multi* m, uint32_t r);
const char* rt_multi_last_error(const rt_multi* m);
void rt_destroy_multi(rt_multi* m);
/* Message of the last failure on this context (ctx may be NULL for a failed rt_create). */
const char* rt_last_error(const rt_context* ctx);
int rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HIP_RAYTRACER_H */
