/*
 * raytracing_amd.h — C ABI of the MI355X (gfx950) hot path of the Neuroglial/RayTracing renderer:
 * KD-tree traversal + watertight ray/triangle intersection + the Path / Whitted bounce loop.
 *
 * Boundary: the reference has no FFI; its extension seam is the Renderer object
 * (reference src/core/Render/Render.h:20-21, registered through AURORA_REGISTER_CLASS,
 * src/core/Object/Object.h:290-300). A Renderer subclass flattens the host scene into
 * rt_scene_desc and calls the entry points below from Renderer::render().
 *
 * Conventions: every entry point returns 0 on success or a negative rt_status, never throws or
 * aborts. Unless a name ends in _dev, pointers are HOST memory owned by the caller; the library
 * copies what it needs before returning. *_dev entry points take DEVICE pointers on the context's
 * GPU (for callers that keep rays / film buffers resident in HBM). One context per GPU (rt_create) or per
 * group of GPUs (rt_create_multi); a context is single-caller (not thread-safe), like the reference's render loop
 * (src/RayTracing.cpp:32-33).
 */
#ifndef RAYTRACING_AMD_H
#define RAYTRACING_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_NO_DEVICE = -2,
    RT_ERR_HIP = -3,
    RT_ERR_NO_SCENE = -4,
    RT_ERR_UNSUPPORTED = -5,
    RT_ERR_NO_CAMERA = -6
} rt_status;

/* Shape kinds (reference: ATriangleShape "Triangle" TriangleShape.cpp:120, ASphereShape "Sphere" SphereShape.cpp:10). */
enum { RT_SHAPE_TRIANGLE = 0, RT_SHAPE_SPHERE = 1 };

/* Material kinds. LAMBERTIAN = LambertianMaterial (Material/LambertianMaterial.cpp:19-28),
 * MIRROR = MirrorMaterial (Material/MirrorMaterial.cpp:19-29). GLASS has no reference material: it is
 * the pair {ASpecularReflection(R, AFresnelDielectric(1, eta)), ASpecularTransmission(T, 1, eta, Radiance)}
 * built from the reference's BxDFs (Render/BSDF.h:134-176). */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_MIRROR = 1, RT_MAT_GLASS = 2 };

typedef struct rt_material {
    int32_t type;      /* RT_MAT_* */
    float rgb[3];      /* Kr (Lambertian / Mirror), R = T (Glass) */
    float eta;         /* Glass only */
} rt_material;

/* One DiffuseAreaLight per emissive primitive (reference Object/Entity.cpp:169-176, Render/Light.h:95-133). */
typedef struct rt_light {
    int32_t prim_id;
    float emit[3];     /* m_Lemit */
    int32_t two_sided; /* m_twoSided */
} rt_light;

/* ASphereShape (Shape/SphereShape.h:11-37): radius in object space plus the owning Entity's transforms
 * (m_objectToWorld and its inverse, Object/Entity.cpp:69-70), column-major as glm stores them. */
typedef struct rt_sphere {
    int32_t prim_id;             /* index into the primitive list; shape_type[prim_id] must be RT_SHAPE_SPHERE */
    float radius;
    float object_to_world[16];
    float world_to_object[16];
} rt_sphere;

/*
 * Flattened scene. The KD-tree arrays are the reference's own, verbatim
 * (KdTree::m_nodes / m_hitableIndices / m_bounds, Accelerators/KDTree.h:40-45):
 *   nodes[2*i+0] = float split | int oneHitable | int indicesOffset   (KDTree.cpp:29-34)
 *   nodes[2*i+1] = low 2 bits: split axis 0,1,2 or 3 = leaf; high 30 bits: aboveChild (interior)
 *                  or nHitables (leaf)                                (KDTree.cpp:38-43)
 *   the below child of interior node i is node i+1.
 * Primitive i is the reference's m_hitables[i]; triangles carry world-space vertices
 * (TriangleMesh bakes objectToWorld at load, Shape/TriangleShape.cpp:100-111).
 */
typedef struct rt_scene_desc {
    uint32_t n_nodes;
    const uint32_t *nodes;       /* 2 * n_nodes words */
    uint32_t n_leaf_idx;
    const int32_t *leaf_idx;     /* may be NULL when n_leaf_idx == 0 */
    float bounds[6];             /* pMin.xyz, pMax.xyz */
    uint32_t n_prims;
    const uint8_t *shape_type;   /* n_prims, RT_SHAPE_*; NULL = all triangles */
    const float *tri_p;          /* n_prims * 9: p0, p1, p2 (ignored for spheres) */
    const float *tri_n;          /* n_prims * 9: n0, n1, n2 or NULL when no primitive has normals */
    const uint8_t *tri_has_n;    /* n_prims or NULL (= all have normals iff tri_n != NULL) */
    const float *tri_uv;         /* n_prims * 6 or NULL: default (0,0),(1,0),(1,1) (TriangleShape.cpp:396-401) */
    const int32_t *material_id;  /* n_prims */
    const int32_t *light_id;     /* n_prims, -1 = not emissive */
    uint32_t n_materials;
    const rt_material *materials;
    uint32_t n_lights;
    const rt_light *lights;      /* order = Scene::m_lights (Scene/SceneParser.cpp:166-172) */
    uint32_t n_spheres;          /* one entry per primitive whose shape_type is RT_SHAPE_SPHERE */
    const rt_sphere *spheres;
} rt_scene_desc;

enum { RT_INTEGRATOR_PATH = 0, RT_INTEGRATOR_WHITTED = 1 };

/* RNG layout. REFERENCE_TILE_STREAM reproduces the reference's sampler exactly: one PCG32 stream per
 * 16x16 tile seeded with the tile index, consumed serially pixel by pixel, sample by sample
 * (Render/Render.cpp:32-92, Render/Sampler.cpp:174-179). COUNTER gives every camera sample its own
 * PCG32 stream (sequence = (pixel index * spp_total + sample index) ^ seed scrambling, see rt_render).
 * The tile stream is serial inside a tile and cannot be cut by sample index (spp_begin = 0, spp_end = spp_total); for the
 * Path integrator it runs one tile per wavefront and is meant for reproducing the CPU renderer's image, COUNTER is the
 * wavefront path tracer. */
enum { RT_RNG_COUNTER = 0, RT_RNG_REFERENCE_TILE_STREAM = 1 };

typedef struct rt_render_params {
    int32_t integrator;    /* RT_INTEGRATOR_* */
    int32_t spp_total;     /* Sampler "SPP": samples per pixel of the whole job */
    int32_t spp_begin;     /* this call renders sample indices [spp_begin, spp_end) — the multi-GPU shard */
    int32_t spp_end;
    int32_t max_depth;     /* Renderer "Depth" (Render.cpp:321,458) */
    float rr_threshold;    /* PathRenderer::m_rrThreshold, 1 in the reference (Render.cpp:322) */
    int32_t rng_mode;      /* RT_RNG_* */
    uint64_t seed;         /* COUNTER mode only */
    int32_t spp_per_pass;  /* wavefront batch: samples per pixel in flight at once; 0 = library default (the job cut into
                            * equal passes of at most 2^27 paths, their number a multiple of the pipelines; 272 B of queue
                            * state per path and pipeline; fewer paths if that much device memory is not free) */
    int32_t collect_stats; /* nonzero: count nodes / leaves / triangle tests per ray (slow path) */
} rt_render_params;

typedef struct rt_stats {
    /* rays traced since rt_reset_stats: one ray = one KdTree::hit call of either flavour */
    uint64_t rays_closest;     /* extend + MIS-probe rays */
    uint64_t rays_any;         /* shadow rays */
    uint64_t camera_samples;
    /* traversal counters (only advanced when collect_stats != 0) */
    uint64_t n_interior;
    uint64_t n_leaf;
    uint64_t n_leaf_idx;
    uint64_t n_tri_tests;
    uint64_t counted_rays;
    /* HIP-event time of the closest-hit (extend) kernel launches since rt_reset_stats (RT_OPTION_KERNEL_TIMES = 1) */
    uint64_t extend_launches;
    double extend_ms;
    uint64_t extend_rays;      /* rays processed by those launches */
    double render_ms;          /* wall time of rt_render* calls (host clock, device-synchronised) */
    /* HIP-event time of the shade kernel launches (one per bounce; they process the rays of extend_rays) */
    uint64_t shade_launches;
    double shade_ms;
    /* state of the context (not counters; rt_reset_stats leaves them alone) */
    uint64_t queue_bytes_per_pipeline;  /* device memory of one wavefront pipeline's queues and per-path state; 0 before the first Path render */
    uint64_t queue_paths;               /* paths in flight those queues are sized for */
    int32_t pipelines;                  /* RT_OPTION_PIPELINES */
    int32_t scene_in_lds;               /* 1 when the traversal tables of the uploaded scene are staged in LDS by every tracing workgroup */
} rt_stats;

/* Library / context ------------------------------------------------------------------------- */

/* Creates a context on HIP device `device_id`. Fails with RT_ERR_NO_DEVICE when no GPU is visible:
 * there is no CPU fallback. */
int rt_create(int device_id, rt_context **out_ctx);
/* One context over several GPUs of this node: the C++ host's whole-node drop-in for Renderer::render (reference
 * src/core/Render/Render.h:20-21 is called once from main, src/RayTracing.cpp:32-33; the reference itself has no
 * multi-device code). The context owns one sub-context per listed device and the RCCL communicator between them.
 * rt_upload_scene, rt_set_camera, rt_set_film and rt_set_option address every device (the scene is replicated).
 * rt_render / rt_render_dev split [spp_begin, spp_end) by sample index - device k of G renders
 * [b + k (e - b) / G, b + (k + 1) (e - b) / G) of every pixel - and sum the per-device float films into device 0's with
 * ONE ncclReduce (sum, fp32) over a packed buffer of 4 * width * height floats before adding them to the caller's
 * buffers (device pointers of the *_dev variants live on device_ids[0]). Jobs that cannot be cut by sample index (the
 * reference's tile stream) and the batch tracing, film development and no-allocation entry points run on device 0.
 * RCCL (librccl.so) is opened at run time and only when n_devices > 1; n_devices == 1 is rt_create. Listing one device
 * n_devices times is a one-GPU rehearsal of the same code path: RCCL refuses duplicate devices, so the films are then
 * summed by a device-local kernel in member order. Any other list with a repeated device is RT_ERR_INVALID_ARG.
 * rt_get_stats sums the counters over the devices; its kernel times (extend_ms, shade_ms) are the slowest device's, since
 * the devices run side by side, and render_ms is the host clock of the call. */
int rt_create_multi(const int *device_ids, int n_devices, rt_context **out_ctx);
int rt_device_count(const rt_context *ctx);   /* devices of the context (1 for rt_create) */
/* The sample-index range member `member` of `n_members` renders of a job [spp_begin, spp_end): contiguous, balanced to
 * within one sample, the union over the members is the job. Host arithmetic only (no GPU needed); rt_render uses it, and
 * a caller that runs one process per GPU (one rt_create each) uses it to pick each process's spp_begin / spp_end.
 * n_members <= 0, a member outside [0, n_members) or spp_end < spp_begin give the empty range [spp_begin, spp_begin);
 * a NULL output pointer is skipped. */
void rt_shard_range(int spp_begin, int spp_end, int member, int n_members, int *out_begin, int *out_end);
void rt_destroy(rt_context *ctx);
const char *rt_last_error(const rt_context *ctx);   /* ctx may be NULL: last creation error */
const char *rt_version(void);

/* Scene / camera ------------------------------------------------------------------------------ */

/* Replaces what the reference does implicitly by holding Scene / KdTree objects
 * (Scene/Scene.h:33-37, Accelerators/KDTree.h:36-45). Copies everything to the device. */
int rt_upload_scene(rt_context *ctx, const rt_scene_desc *scene);

/* Camera::m_cameraToWorld and AProjectiveCamera::m_rasterToCamera (Camera/Camera.h:42,59), column-major
 * as glm stores them; film resolution and the sample bounds [x0,x1) x [y0,y1) of Film::getSampleBounds
 * (Object/Film.cpp:79-85). Replaces Camera::castingRay (Camera/Camera.cpp:87-95) on the device. */
int rt_set_camera(rt_context *ctx, const float raster_to_camera[16], const float camera_to_world[16],
                  int width, int height, const int sample_bounds[4]);

/* Film parameters that differ from what rt_set_camera installs (0.5 box filter, cropped pixel bounds = sample bounds
 * clipped to the film, no clamp): the Box filter's "Radius" (Render/Filter.cpp:4-24; the reference's only filter, weight 1
 * inside the radius), Film::m_croppedPixelBounds (Object/Film.cpp:21-24) - only these pixels receive samples
 * (Film::getFilmTile, :87-96; FilmTile::addSample, Object/Film.h:111-155) - and "MaxLum" (Film.cpp:30, Film.h:113-115;
 * pass INFINITY for none). The sample bounds given to rt_set_camera must be Film::getSampleBounds for this radius and
 * crop (Film.cpp:79-85). Call after rt_set_camera; a later rt_set_camera resets the defaults. */
int rt_set_film(rt_context *ctx, float radius_x, float radius_y, const int cropped_bounds[4], float max_sample_luminance);

/* Batch tracing --------------------------------------------------------------------------------- */

/* Closest hit: KdTree::hit(ray, isect) (Accelerators/KDTree.cpp:385-490) over n rays. d is used as
 * given (callers pass the normalised direction a reference Ray holds, Utils/Math.h:326-327).
 * Outputs: prim_id (-1 = miss), t (= ray.m_tMax after the call), bary (b0,b1,b2 of
 * Shape/TriangleShape.cpp:361-363; for a sphere hit (t, 0, 0): ASphereShape::hit, Shape/SphereShape.cpp:167-235, has
 * no barycentrics). Any output pointer may be NULL. */
int rt_trace_closest(rt_context *ctx, uint64_t n, const float *o_xyz, const float *d_xyz,
                     const float *tmax, int32_t *prim_id, float *t, float *bary3);
/* Any hit: KdTree::hit(ray) (Accelerators/KDTree.cpp:284-383). occluded[i] = 1 when something is hit. */
int rt_trace_any(rt_context *ctx, uint64_t n, const float *o_xyz, const float *d_xyz,
                 const float *tmax, uint8_t *occluded);
/* How traversal maps rays to lanes. No counterpart in the reference (its CPU loop has no such choice); results
 * are identical under all settings, only the speed differs.
 *   RT_SCHEDULE_AUTO (default): LOCKSTEP for scenes whose traversal tables fit in LDS, REFILL otherwise.
 *   RT_SCHEDULE_LOCKSTEP:       one ray per lane, a wave ends with its slowest ray. Fastest for coherent batches and
 *                               small scenes (cornellBox camera rays: 37 vs 23 G rays/s).
 *   RT_SCHEDULE_STREAM:         lanes pick up the next ray as soon as theirs ends. Fastest for incoherent rays and for
 *                               scenes in HBM (1 M-triangle soup, incoherent rays: 0.63 vs 0.36 G rays/s).
 *   RT_SCHEDULE_REFILL:         lanes fetch and prepare their next ray themselves once enough of them are idle; no pool
 *                               of prepared rays in LDS, so more waves stay resident than under STREAM (6 against 4 per
 *                               SIMD on scenes in HBM). Fastest for the bounce rays of a scene in LDS (cornellBox: 9.6 vs
 *                               6.4 G rays/s in lockstep) and on scenes in HBM (4 M-triangle mesh, incoherent rays: 1.55 vs
 *                               1.31 G rays/s streaming).
 * Applies to rt_trace_* and, for scenes in HBM, to the extend and shadow stages of rt_render. On scenes in LDS rt_render's
 * extend stage runs LOCKSTEP for camera rays and REFILL from the first bounce on, whatever the option says. */
enum { RT_OPTION_TRACE_SCHEDULE = 1, RT_OPTION_PIPELINES = 2, RT_OPTION_KERNEL_TIMES = 3 };
enum { RT_SCHEDULE_AUTO = 0, RT_SCHEDULE_LOCKSTEP = 1, RT_SCHEDULE_STREAM = 2, RT_SCHEDULE_REFILL = 3 };
/* RT_OPTION_PIPELINES: how many passes of the wavefront path tracer overlap on separate streams (1..RT_MAX_PIPELINES,
 * default RT_MAX_PIPELINES). 1 runs the passes one after the other, which is how a single kernel's launch time is
 * measured without other passes sharing the CUs. Results do not depend on it. */
enum { RT_MAX_PIPELINES = 3 };
/* RT_OPTION_KERNEL_TIMES: 1 = rt_render / rt_render_dev record a HIP event before and after every extend and shade launch
 * of a Path render, which rt_get_stats turns into extend_ms / shade_ms (and their launch counts); 0 (default) = no events
 * in the render's streams, and those fields stay 0. A measurement times the job with 0 and reads launch times from a
 * separate run with 1. */
int rt_set_option(rt_context *ctx, int option, int value);

/* Same, all pointers in device memory, asynchronous on the context's stream (call rt_synchronize). */
int rt_trace_closest_dev(rt_context *ctx, uint64_t n, const float *o_xyz, const float *d_xyz,
                         const float *tmax, int32_t *prim_id, float *t, float *bary3);
int rt_trace_any_dev(rt_context *ctx, uint64_t n, const float *o_xyz, const float *d_xyz,
                     const float *tmax, uint8_t *occluded);

/* Rendering --------------------------------------------------------------------------------------- */

/* Replaces SamplerRenderer::render's sample loop (Render/Render.cpp:29-98) with PathRenderer::Li
 * (Render.cpp:348-449) or WhittedRenderer::Li (Render.cpp:473-532). Adds, for every film pixel
 * (row-major, width*height), the box-filtered sums FilmTile::addSample keeps
 * (Object/Film.h:111-155): rgb_sum += L * weight, weight_sum += weight. Output buffers are
 * ACCUMULATED INTO (zero them for a fresh render). Host variant blocks and copies back. */
int rt_render(rt_context *ctx, const rt_render_params *params, float *rgb_sum, float *weight_sum);
/* Device variant: rgb_sum_dev / weight_sum_dev live on the context's GPU; returns after the work is
 * queued on the context's stream (rt_synchronize to wait). */
int rt_render_dev(rt_context *ctx, const rt_render_params *params, float *rgb_sum_dev, float *weight_sum_dev);

/* rt_render_dev for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc). No counterpart in the reference. The call allocates nothing, on the device or through any library: the
 * wavefront queues, queue counters, film spill buffers and statistics counters that rt_render_dev keeps in memory the
 * context allocates on demand live here in a fixed array inside the library's code object, RT_NOALLOC_SCRATCH_MB per GPU,
 * held from the moment the library is loaded on a GPU until the process ends. Same kernels, same results as rt_render_dev.
 * Limits (RT_ERR_UNSUPPORTED beyond them): film and sample bounds of at most RT_NOALLOC_MAX_PATHS pixels (one sample per
 * pixel in flight at that size, more for smaller films), max_depth <= RT_NOALLOC_MAX_DEPTH, rng_mode RT_RNG_COUNTER,
 * collect_stats 0; one pipeline (passes run one after the other). Rays it traces are not added to rt_get_stats.
 * All contexts on one GPU share that GPU's array: calls that overlap on one GPU race - run them one at a time
 * (rt_synchronize between calls on different contexts of the same GPU; calls on one context are ordered by its stream). */
enum { RT_NOALLOC_MAX_PATHS = 1 << 18, RT_NOALLOC_MAX_DEPTH = 64, RT_NOALLOC_SCRATCH_MB = 73 };
int rt_render_noalloc_dev(rt_context *ctx, const rt_render_params *params, float *rgb_sum_dev, float *weight_sum_dev);

/* Film development on the device: what Film::mergeFilmTile + Film::writeImageToFile do to the sums before the PNG is
 * written (Object/Film.cpp:98-167: RGB -> XYZ -> RGB, divide by weight_sum, clamp at 0, multiply by the film's
 * "Scale", sRGB curve, 8 bits). rgb8 receives width*height*3 bytes, row-major; the film size is the one given to
 * rt_set_camera. Encoding the PNG stays with the caller. rt_develop takes host buffers and blocks.
 * The one libm call on this path is the sRGB curve's powf (Utils/Base.h:50-55), whose last bit is implementation-defined:
 * the device evaluates pow in double and rounds once, so against a host libm (rt_host_develop, or the reference itself) a
 * byte may differ by ONE code where 255 * g + 0.5 sits on an integer (measured: fewer than 1 byte in 10 000). */
int rt_develop(rt_context *ctx, const float *rgb_sum, const float *weight_sum, float film_scale, uint8_t *rgb8);
int rt_develop_dev(rt_context *ctx, const float *rgb_sum_dev, const float *weight_sum_dev, float film_scale, uint8_t *rgb8_dev);

int rt_synchronize(rt_context *ctx);
/* The hipStream_t the context launches on (as void*), for callers that record their own events. */
void *rt_stream(rt_context *ctx);

int rt_get_stats(rt_context *ctx, rt_stats *out);
int rt_reset_stats(rt_context *ctx);

/* Diagnostics (no counterpart in the reference; host-side, no GPU needed for the first two) ----------------------- */

/* The device code's correctly rounded sine / cosine, evaluated on the host: what stands in for libm's sinf / cosf in
 * concentricSampleDisk (theta in [-pi/4, 3pi/4], Render/Sampler.cpp) and in the sphere code (any theta,
 * Shape/SphereShape.cpp:217-222, 262-266). Tests compare them with a float64 evaluation rounded once. */
void rt_debug_sincos_disk(float theta, float *sin_out, float *cos_out);
void rt_debug_sincos_cr(float theta, float *sin_out, float *cos_out);
/* Count passes the last Whitted render with RT_RNG_REFERENCE_TILE_STREAM needed to place every pixel in its tile's random
 * stream (-1: the serial fallback ran; 0: no such render yet). */
int rt_debug_whitted_passes(const rt_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACING_AMD_H */
