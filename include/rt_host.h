/*
 * rt_host.h — C ABI of the host-side scene front-end that sits in front of raytracing_amd.h.
 *
 * The reference keeps scene loading, camera set-up, the KD-tree build and image output on the host;
 * a reference-side adapter hands its own objects to raytracing_amd.h and needs nothing from here.
 * This library exists so the device path can be driven without the reference: it restates
 *   - the scene JSON front-end       (reference src/core/Scene/SceneParser.cpp:20-179,
 *                                      src/core/Object/Entity.cpp:93-178, Object/Object.cpp:26-136),
 *   - the OBJ import as assimp performs it for these flags (Shape/TriangleShape.cpp:76-78),
 *   - the SAH KD-tree build           (src/core/Accelerators/KDTree.cpp:86-280),
 *   - the perspective camera matrices (src/core/Camera/Camera.cpp:14-66, Utils/Transform.cpp:52-74),
 *   - film bounds and image output    (src/core/Object/Film.cpp:10-40,79-85,98-167).
 * Pure CPU code; no GPU is needed to call it.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>
#include "raytracing_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene rt_host_scene;

/* What Renderer / Sampler / Camera / Film hold after Parser::parser (SceneParser.cpp:122-131). */
typedef struct rt_host_config {
    int32_t integrator;        /* RT_INTEGRATOR_* from Renderer "Type": "Path" | "Whitted" */
    int32_t max_depth;         /* "Depth", default 2 */
    int32_t spp;               /* Sampler "SPP", default 1 */
    int32_t width, height;     /* Film "Resolution", default 800x600 */
    int32_t cropped_bounds[4]; /* x0,y0,x1,y1 of Film::m_croppedPixelBounds */
    int32_t sample_bounds[4];  /* x0,y0,x1,y1 of Film::getSampleBounds */
    float filter_radius[2];    /* Box filter "Radius", default 0.5 */
    float film_scale;          /* "Scale", default 1 */
    float raster_to_camera[16];
    float camera_to_world[16];
    float fov, eye[3], focus[3], up[3];
    char filename[256];        /* Film "Filename" */
    float max_sample_luminance; /* Film "MaxLum", default +inf (Object/Film.cpp:30) */
} rt_host_config;

/* KD-tree shape numbers, for checking a build against known answers. */
typedef struct rt_host_tree_stats {
    uint32_t n_nodes, n_leaves, n_empty_leaves, max_leaf_prims, n_prim_refs, n_leaf_idx;
    int32_t max_depth_param;   /* round(8 + 1.3 log2 N), KDTree.cpp:97 */
} rt_host_tree_stats;

const char *rt_host_last_error(void);

/* Parse a scene JSON (and the OBJ files it names), build the KD-tree. */
int rt_host_load_json(const char *path, rt_host_scene **out);

/* Build a scene from caller-supplied world-space triangles (synthetic scenes). Arrays as in
 * rt_scene_desc; tri_n / tri_uv may be NULL. The KD-tree is built with the reference's defaults
 * (isectCost 80, travCost 1, emptyBonus 0.5, maxPrims 1, maxDepth -1; KDTree.h:16-17). */
int rt_host_build(uint32_t n_prims, const float *tri_p, const float *tri_n, const float *tri_uv,
                  const int32_t *material_id, const int32_t *light_id,
                  uint32_t n_materials, const rt_material *materials,
                  uint32_t n_lights, const rt_light *lights, rt_host_scene **out);

void rt_host_free(rt_host_scene *s);

/* Threads the KD-tree build of the next rt_host_load_json / rt_host_build may use (the reference builds on one,
 * Accelerators/KDTree.cpp:86-129): 0 = as many as the machine has (default), 1 = one. The tree is the same either way. */
void rt_host_set_build_threads(int threads);

/* Pointers stay valid until rt_host_free. */
const rt_scene_desc *rt_host_scene_desc(const rt_host_scene *s);
int rt_host_get_config(const rt_host_scene *s, rt_host_config *out);
int rt_host_tree_stats_get(const rt_host_scene *s, rt_host_tree_stats *out);

/* Camera matrices for a perspective camera (APerspectiveCamera JSON constructor + initialize). */
int rt_host_camera(float fov_deg, const float eye[3], const float focus[3], const float up[3],
                   int width, int height, float raster_to_camera[16], float camera_to_world[16]);

/* Film::mergeFilmTile + writeImageToFile arithmetic: rgb_sum/weight_sum (width*height, row-major) ->
 * 8-bit sRGB, 3 bytes per pixel. film_scale = Film "Scale". */
int rt_host_develop(const float *rgb_sum, const float *weight_sum, int width, int height,
                    float film_scale, uint8_t *rgb8);

/* Camera ray for a film position, on the host: Camera::castingRay. */
int rt_host_camera_ray(const float raster_to_camera[16], const float camera_to_world[16],
                       float film_x, float film_y, float o[3], float d[3]);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
