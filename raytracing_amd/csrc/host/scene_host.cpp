// Host scene front-end: scene JSON -> flat rt_scene_desc (+ render configuration), camera matrices,
// film development. See include/rt_host.h for the reference files each entry point restates.
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "hostmath.h"
#include "json_min.h"
#include "kd_build.h"
#include "mesh_import.h"

using namespace rth;

namespace {

thread_local std::string g_err;
int g_buildThreads = 0;   // rt_host_set_build_threads

// ---- property values ---------------------------------------------------------------------------
// The reference turns every JSON leaf into a string and parses it again on use
// (reference src/core/Scene/SceneParser.cpp:44-69, src/core/Object/Object.h:64-131): floats go through
// operator<< of a float (6 significant digits), ints through operator<< of an int.
std::string leaf_to_string(const JsonValue &v) {
    char buf[64];
    switch (v.kind) {
    case JsonValue::String: return v.s;
    case JsonValue::Bool: return v.b ? "true" : "false";
    case JsonValue::Float: snprintf(buf, sizeof buf, "%g", (double)(float)v.f); return buf;
    default: snprintf(buf, sizeof buf, "%d", (int)v.i); return buf;
    }
}

struct Props {
    const JsonValue *node;
    explicit Props(const JsonValue *n) : node(n) {}
    std::vector<std::string> values(const std::string &k) const {
        std::vector<std::string> out;
        const JsonValue *v = node ? node->find(k) : nullptr;
        if (!v || v->kind == JsonValue::Object) return out;
        if (v->kind == JsonValue::Array) for (auto &e : v->arr) out.push_back(leaf_to_string(e));
        else out.push_back(leaf_to_string(*v));
        return out;
    }
    bool has(const std::string &k) const { return !values(k).empty(); }
    static float tof(const std::string &s) { return strtof(s.c_str(), nullptr); }
    float getFloat(const std::string &k, float def) const { auto v = values(k); return v.empty() ? def : tof(v[0]); }
    float getFloat(const std::string &k) const {
        auto v = values(k);
        if (v.empty()) throw std::runtime_error("property \"" + k + "\" is missing");
        return tof(v[0]);
    }
    int getInteger(const std::string &k, int def) const { auto v = values(k); return v.empty() ? def : (int)tof(v[0]); }
    std::string getString(const std::string &k, const std::string &def) const { auto v = values(k); return v.empty() ? def : v[0]; }
    std::string getString(const std::string &k) const {
        auto v = values(k);
        if (v.empty()) throw std::runtime_error("property \"" + k + "\" is missing");
        return v[0];
    }
    bool getBoolean(const std::string &k, bool def) const { auto v = values(k); return v.empty() ? def : v[0] == "true"; }
    std::vector<float> getVectorNf(const std::string &k) const {
        std::vector<float> out;
        for (auto &s : values(k)) out.push_back(tof(s));
        return out;
    }
    V3 getVector3f(const std::string &k) const {
        auto v = getVectorNf(k);
        if (v.size() < 3) throw std::runtime_error("property \"" + k + "\" needs 3 components");
        return {v[0], v[1], v[2]};
    }
    V3 getVector3f(const std::string &k, V3 def) const { return has(k) ? getVector3f(k) : def; }
    bool getVector2f(const std::string &k, float out[2]) const {
        auto v = getVectorNf(k);
        if (v.size() < 2) return false;
        out[0] = v[0]; out[1] = v[1];
        return true;
    }
    const JsonValue *child(const std::string &k) const {
        const JsonValue *v = node ? node->find(k) : nullptr;
        return (v && v->kind == JsonValue::Object) ? v : nullptr;
    }
};

// Shape "Transform" token stream (reference src/core/Object/Entity.cpp:104-150).
Xf parse_transform(const Props &shape) {
    Xf o2w = xf_identity();
    if (!shape.has("Transform")) return o2w;
    std::vector<float> seq = shape.getVectorNf("Transform");
    std::vector<Xf> stack;
    size_t it = 0;
    while (it < seq.size()) {
        int token = static_cast<int>(seq[it]);
        if (token == 0 || token == 1) {
            if (it + 3 >= seq.size()) throw std::runtime_error("Transform: truncated token stream");
            if (token == 0) stack.push_back(xf_translate({seq[it + 1], seq[it + 2], seq[it + 3]}));
            else stack.push_back(xf_scale(seq[it + 1], seq[it + 2], seq[it + 3]));
            it += 4;
        } else if (token == 2) {
            if (it + 4 >= seq.size()) throw std::runtime_error("Transform: truncated token stream");
            stack.push_back(xf_rotate(seq[it + 4], {seq[it + 1], seq[it + 2], seq[it + 3]}));
            it += 5;
        } else {
            return xf_identity();   // undefined action: the reference logs and keeps the identity
        }
    }
    for (auto r = stack.rbegin(); r != stack.rend(); ++r) o2w = xf_mul(o2w, *r);
    return o2w;
}

void perspective_camera(float fov, V3 eye, V3 focus, V3 up, int resx, int resy, M4 &rasterToCamera, M4 &cameraToWorld) {
    // APerspectiveCamera(const PropertyTreeNode&) (reference src/core/Camera/Camera.cpp:48-66)
    M4 worldToCamera = m4_lookat_lh(eye, focus, up);
    Xf look{worldToCamera, m4_inverse(worldToCamera)};           // lookAt() (Utils/Transform.cpp:52-56)
    Xf camToWorld = xf_inverse(look);
    // perspective(fov, 1e-2, 1000) (Utils/Transform.cpp:63-74)
    const float n = 1e-2f, f = 1000.f;
    M4 persp;
    persp.c[0] = {1, 0, 0, 0};
    persp.c[1] = {0, 1, 0, 0};
    persp.c[2] = {0, 0, f / (f - n), 1};
    persp.c[3] = {0, 0, -f * n / (f - n), 0};
    const float invTanAng = 1 / std::tan(radians(fov) / 2);
    Xf camToScreen = xf_mul(xf_scale(invTanAng, invTanAng, 1), Xf{persp, m4_inverse(persp)});
    // AProjectiveCamera::initialize (Camera/Camera.cpp:14-41)
    float sx0, sx1, sy0, sy1;
    const float frame = (float)(resx) / resy;
    if (frame > 1.f) { sx0 = -frame; sx1 = frame; sy0 = -1.f; sy1 = 1.f; }
    else { sx0 = -1.f; sx1 = 1.f; sy0 = -1.f / frame; sy1 = 1.f / frame; }
    Xf screenToRaster = xf_mul(xf_mul(xf_scale((float)resx, (float)resy, 1), xf_scale(1 / (sx1 - sx0), 1 / (sy0 - sy1), 1)),
                               xf_translate({-sx0, -sy1, 0}));
    Xf rasterToScreen = xf_inverse(screenToRaster);
    Xf r2c = xf_mul(xf_inverse(camToScreen), rasterToScreen);
    rasterToCamera = r2c.m;
    cameraToWorld = camToWorld.m;
}

void store_m4(const M4 &m, float out[16]) {
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out[c * 4 + r] = m[c][r];
}
M4 load_m4(const float in[16]) {
    M4 m;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m[c][r] = in[c * 4 + r];
    return m;
}

}  // namespace

struct rt_host_scene {
    // flat storage
    std::vector<float> tri_p, tri_n, tri_uv;
    std::vector<uint8_t> shape_type, tri_has_n;
    std::vector<int32_t> material_id, light_id;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_sphere> spheres;
    KdTreeFlat tree;
    rt_scene_desc desc{};
    rt_host_config cfg{};
    bool has_cfg = false;

    void finish() {
        const size_t n = material_id.size();
        std::vector<Box3> pb(n);
        for (size_t i = 0; i < n; ++i) {   // ATriangleShape::worldBound (Shape/TriangleShape.cpp:144-151)
            const float *p = &tri_p[9 * i];
            pb[i] = box_union(box_of({p[0], p[1], p[2]}, {p[3], p[4], p[5]}), V3{p[6], p[7], p[8]});
        }
        shape_type.assign(n, RT_SHAPE_TRIANGLE);
        for (const rt_sphere &sp : spheres) {
            // Shape::worldBound = objectToWorld(objectBound) (Shape/Shape.cpp:26, SphereShape.cpp:20-24, Utils/Transform.cpp:82-95)
            shape_type[sp.prim_id] = RT_SHAPE_SPHERE;
            const M4 m = load_m4(sp.object_to_world);
            const float r = sp.radius;
            const V3 lo{-r, -r, -r}, hi{r, r, r};
            const V3 corner[8] = {{lo.x, lo.y, lo.z}, {hi.x, lo.y, lo.z}, {lo.x, hi.y, lo.z}, {lo.x, lo.y, hi.z},
                                  {lo.x, hi.y, hi.z}, {hi.x, hi.y, lo.z}, {hi.x, lo.y, hi.z}, {hi.x, hi.y, hi.z}};
            const V3 first = xf_apply(m, corner[0], 1.0f);
            Box3 b;
            b.pMin = first;   // BBox3(p): both corners are the point
            b.pMax = first;
            for (int k = 1; k < 8; ++k) b = box_union(b, xf_apply(m, corner[k], 1.0f));
            pb[sp.prim_id] = b;
        }
        kd_build(pb, tree, -1, g_buildThreads);
        desc.n_nodes = (uint32_t)(tree.nodes.size() / 2);
        desc.nodes = tree.nodes.data();
        desc.n_leaf_idx = (uint32_t)tree.leafIdx.size();
        desc.leaf_idx = tree.leafIdx.empty() ? nullptr : tree.leafIdx.data();
        desc.bounds[0] = tree.bounds.pMin.x; desc.bounds[1] = tree.bounds.pMin.y; desc.bounds[2] = tree.bounds.pMin.z;
        desc.bounds[3] = tree.bounds.pMax.x; desc.bounds[4] = tree.bounds.pMax.y; desc.bounds[5] = tree.bounds.pMax.z;
        desc.n_prims = (uint32_t)n;
        desc.n_spheres = (uint32_t)spheres.size();
        desc.spheres = spheres.empty() ? nullptr : spheres.data();
        desc.shape_type = shape_type.data();
        desc.tri_p = tri_p.data();
        desc.tri_n = tri_n.empty() ? nullptr : tri_n.data();
        desc.tri_has_n = tri_has_n.empty() ? nullptr : tri_has_n.data();
        desc.tri_uv = tri_uv.empty() ? nullptr : tri_uv.data();
        desc.material_id = material_id.data();
        desc.light_id = light_id.data();
        desc.n_materials = (uint32_t)materials.size();
        desc.materials = materials.data();
        desc.n_lights = (uint32_t)lights.size();
        desc.lights = lights.empty() ? nullptr : lights.data();
    }
};

static void load_json_impl(const std::string &path, rt_host_scene &S) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("could not open the json file: " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    JsonValue root = JsonParser(text).parse();

    std::string dir;
    {
        size_t k = path.rfind('\\');
        if (k == std::string::npos) k = path.rfind('/');
        if (k != std::string::npos) dir = path.substr(0, k + 1);
    }

    // ---- Renderer / Sampler / Camera / Film / Filter -----------------------------------------
    const JsonValue *rj = root.find("Renderer");
    if (!rj) throw std::runtime_error("there is no Renderer in " + path);
    Props rp(rj);
    rt_host_config &c = S.cfg;
    std::string rtype = rp.getString("Type");
    if (rtype == "Path") c.integrator = RT_INTEGRATOR_PATH;
    else if (rtype == "Whitted") c.integrator = RT_INTEGRATOR_WHITTED;
    else throw std::runtime_error("unknown Renderer type \"" + rtype + "\"");
    c.max_depth = rp.getInteger("Depth", 2);
    Props sp(rp.child("Sampler"));
    c.spp = sp.getInteger("SPP", 1);
    Props cp(rp.child("Camera"));
    if (cp.getString("Type", "") != "Perspective") throw std::runtime_error("only the Perspective camera exists");
    c.fov = cp.getFloat("Fov");
    V3 eye = cp.getVector3f("Eye"), focus = cp.getVector3f("Focus"), up = cp.getVector3f("WorldUp", {0.f, 1.f, 0.f});
    Props fp(cp.child("Film"));
    float res[2] = {800, 600};
    fp.getVector2f("Resolution", res);
    c.width = static_cast<int>(res[0]);
    c.height = static_cast<int>(res[1]);
    float cmin[2] = {0.0f, 0.0f}, cmax[2] = {1.0f, 1.0f};
    fp.getVector2f("CropMin", cmin);
    fp.getVector2f("CropMax", cmax);
    // Film::Film (reference src/core/Object/Film.cpp:21-24); BBox2i(p1,p2) orders the corners
    int ax = (int)std::ceil(c.width * cmin[0]), ay = (int)std::ceil(c.height * cmin[1]);
    int bx = (int)std::ceil(c.width * cmax[0]), by = (int)std::ceil(c.height * cmax[1]);
    c.cropped_bounds[0] = std::min(ax, bx); c.cropped_bounds[1] = std::min(ay, by);
    c.cropped_bounds[2] = std::max(ax, bx); c.cropped_bounds[3] = std::max(ay, by);
    c.film_scale = fp.getFloat("Scale", 1.0f);
    c.max_sample_luminance = fp.getFloat("MaxLum", std::numeric_limits<float>::infinity());
    std::string fname = fp.getString("Filename", "rendered.png");
    snprintf(c.filename, sizeof c.filename, "%s", fname.c_str());
    Props flt(fp.child("Filter"));
    c.filter_radius[0] = c.filter_radius[1] = 0.5f;
    flt.getVector2f("Radius", c.filter_radius);
    // Film::getSampleBounds (Film.cpp:79-85)
    {
        float x0 = std::floor((float)c.cropped_bounds[0] + 0.5f - c.filter_radius[0]);
        float y0 = std::floor((float)c.cropped_bounds[1] + 0.5f - c.filter_radius[1]);
        float x1 = std::ceil((float)c.cropped_bounds[2] - 0.5f + c.filter_radius[0]);
        float y1 = std::ceil((float)c.cropped_bounds[3] - 0.5f + c.filter_radius[1]);
        c.sample_bounds[0] = (int)std::min(x0, x1); c.sample_bounds[1] = (int)std::min(y0, y1);
        c.sample_bounds[2] = (int)std::max(x0, x1); c.sample_bounds[3] = (int)std::max(y0, y1);
    }
    M4 r2c, c2w;
    perspective_camera(c.fov, eye, focus, up, c.width, c.height, r2c, c2w);
    store_m4(r2c, c.raster_to_camera);
    store_m4(c2w, c.camera_to_world);
    for (int i = 0; i < 3; ++i) { c.eye[i] = eye[i]; c.focus[i] = focus[i]; c.up[i] = up[i]; }
    S.has_cfg = true;

    // ---- Entities --------------------------------------------------------------------------------
    static const float kDefaultUV[6] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};
    bool anyUV = false;
    const JsonValue *ej = root.find("Entity");
    if (!ej || ej->kind != JsonValue::Array) throw std::runtime_error("there is no Entity in " + path);
    for (const JsonValue &e : ej->arr) {
        Props ep(&e);
        std::string etype = ep.getString("Type");
        if (etype != "MeshEntity" && etype != "Entity") throw std::runtime_error("unknown entity type \"" + etype + "\"");
        Props shape(ep.child("Shape"));
        Xf o2w = parse_transform(shape);
        Props mp(ep.child("Material"));
        rt_material mat{};
        std::string mtype = mp.getString("Type");
        V3 kr = mp.getVector3f("R");
        mat.rgb[0] = kr.x; mat.rgb[1] = kr.y; mat.rgb[2] = kr.z;
        mat.eta = 1.0f;
        if (mtype == "Lambertian") mat.type = RT_MAT_LAMBERTIAN;
        else if (mtype == "Mirror") mat.type = RT_MAT_MIRROR;
        else if (mtype == "Glass") { mat.type = RT_MAT_GLASS; mat.eta = mp.getFloat("Eta", 1.5f); }
        else throw std::runtime_error("unknown material type \"" + mtype + "\"");
        const int matId = (int)S.materials.size();
        S.materials.push_back(mat);

        const JsonValue *lj = ep.child("Light");
        rt_light light{};
        if (lj) {
            Props lp(lj);
            V3 le = lp.getVector3f("Radiance");
            light.emit[0] = le.x; light.emit[1] = le.y; light.emit[2] = le.z;
            light.two_sided = lp.getBoolean("TwoSided", false) ? 1 : 0;
        }

        if (etype == "Entity") {
            // Entity (reference src/core/Object/Entity.cpp:11-87): one analytic shape, here ASphereShape (SphereShape.cpp:12-15)
            const std::string stype = shape.getString("Type");
            if (stype != "Sphere") throw std::runtime_error("Entity shape \"" + stype + "\" is not a shape of the reference");
            rt_sphere sp{};
            sp.prim_id = (int)S.material_id.size();
            sp.radius = shape.getFloat("Radius", 1.0f);
            store_m4(o2w.m, sp.object_to_world);
            store_m4(o2w.inv, sp.world_to_object);   // inverse(Transform) swaps the pair (Utils/Transform.h)
            S.spheres.push_back(sp);
            for (int k = 0; k < 9; ++k) { S.tri_p.push_back(0.f); S.tri_n.push_back(0.f); }
            for (float v : kDefaultUV) S.tri_uv.push_back(v);
            S.tri_has_n.push_back(0);
            S.material_id.push_back(matId);
            if (lj) {
                light.prim_id = sp.prim_id;
                S.light_id.push_back((int)S.lights.size());
                S.lights.push_back(light);
            } else {
                S.light_id.push_back(-1);
            }
            continue;
        }
        ImportedMesh mesh;
        import_obj(dir + ep.getString("Filename"), mesh);
        // TriangleMesh keeps mTextureCoords[0] when the file has them (TriangleShape.cpp:28-37, 97-110); a triangle of a
        // mesh without them gets (0,0), (1,0), (1,1) (:396-401)
        const bool meshHasUV = !mesh.uv.empty();
        anyUV = anyUV || meshHasUV;
        // TriangleMesh: bake objectToWorld into positions (w=1) and normals (w=0)
        // (reference src/core/Shape/TriangleShape.cpp:100-111)
        std::vector<V3> wp(mesh.position.size()), wn(mesh.position.size());
        for (size_t i = 0; i < mesh.position.size(); ++i) {
            wp[i] = xf_apply(o2w.m, mesh.position[i], 1.0f);
            wn[i] = xf_apply(o2w.m, mesh.normal[i], 0.0f);
        }
        for (size_t t = 0; t + 2 < mesh.indices.size(); t += 3) {
            for (int k = 0; k < 3; ++k) {
                V3 p = wp[mesh.indices[t + k]];
                S.tri_p.push_back(p.x); S.tri_p.push_back(p.y); S.tri_p.push_back(p.z);
            }
            for (int k = 0; k < 3; ++k) {
                V3 n = wn[mesh.indices[t + k]];
                S.tri_n.push_back(n.x); S.tri_n.push_back(n.y); S.tri_n.push_back(n.z);
            }
            for (int k = 0; k < 3; ++k) {
                const size_t vi = (size_t)mesh.indices[t + k];
                S.tri_uv.push_back(meshHasUV ? mesh.uv[2 * vi] : kDefaultUV[2 * k]);
                S.tri_uv.push_back(meshHasUV ? mesh.uv[2 * vi + 1] : kDefaultUV[2 * k + 1]);
            }
            S.tri_has_n.push_back(1);
            S.material_id.push_back(matId);
            if (lj) {
                light.prim_id = (int)S.material_id.size() - 1;
                S.light_id.push_back((int)S.lights.size());
                S.lights.push_back(light);
            } else {
                S.light_id.push_back(-1);
            }
        }
    }
    if (S.material_id.empty()) throw std::runtime_error("scene has no primitives");
    if (!anyUV) S.tri_uv.clear();   // all defaults: the consumer's own default is the same
    S.finish();
}

extern "C" {

const char *rt_host_last_error(void) { return g_err.c_str(); }

int rt_host_load_json(const char *path, rt_host_scene **out) {
    if (!path || !out) { g_err = "null argument"; return RT_ERR_INVALID_ARG; }
    *out = nullptr;
    auto *S = new rt_host_scene();
    try {
        load_json_impl(path, *S);
    } catch (const std::exception &e) {
        g_err = e.what();
        delete S;
        return RT_ERR_INVALID_ARG;
    }
    *out = S;
    return RT_OK;
}

int rt_host_build(uint32_t n_prims, const float *tri_p, const float *tri_n, const float *tri_uv,
                  const int32_t *material_id, const int32_t *light_id, uint32_t n_materials,
                  const rt_material *materials, uint32_t n_lights, const rt_light *lights, rt_host_scene **out) {
    if (!out || !tri_p || !material_id || !materials || n_prims == 0 || n_materials == 0) {
        g_err = "null or empty argument";
        return RT_ERR_INVALID_ARG;
    }
    *out = nullptr;
    for (uint32_t i = 0; i < n_prims; ++i) {
        if (material_id[i] < 0 || (uint32_t)material_id[i] >= n_materials) { g_err = "material_id out of range"; return RT_ERR_INVALID_ARG; }
        if (light_id && light_id[i] >= (int32_t)n_lights) { g_err = "light_id out of range"; return RT_ERR_INVALID_ARG; }
    }
    for (uint32_t i = 0; i < n_lights; ++i)
        if (!lights || lights[i].prim_id < 0 || (uint32_t)lights[i].prim_id >= n_prims) { g_err = "light prim_id out of range"; return RT_ERR_INVALID_ARG; }
    auto *S = new rt_host_scene();
    try {
        S->tri_p.assign(tri_p, tri_p + (size_t)n_prims * 9);
        if (tri_n) { S->tri_n.assign(tri_n, tri_n + (size_t)n_prims * 9); S->tri_has_n.assign(n_prims, 1); }
        if (tri_uv) S->tri_uv.assign(tri_uv, tri_uv + (size_t)n_prims * 6);
        S->material_id.assign(material_id, material_id + n_prims);
        if (light_id) S->light_id.assign(light_id, light_id + n_prims);
        else S->light_id.assign(n_prims, -1);
        S->materials.assign(materials, materials + n_materials);
        if (n_lights) S->lights.assign(lights, lights + n_lights);
        S->finish();
    } catch (const std::exception &e) {
        g_err = e.what();
        delete S;
        return RT_ERR_INVALID_ARG;
    }
    *out = S;
    return RT_OK;
}

void rt_host_free(rt_host_scene *s) { delete s; }

void rt_host_set_build_threads(int threads) { g_buildThreads = threads < 0 ? 0 : threads; }

const rt_scene_desc *rt_host_scene_desc(const rt_host_scene *s) { return s ? &s->desc : nullptr; }

int rt_host_get_config(const rt_host_scene *s, rt_host_config *out) {
    if (!s || !out) { g_err = "null argument"; return RT_ERR_INVALID_ARG; }
    if (!s->has_cfg) { g_err = "scene was not loaded from a scene file"; return RT_ERR_NO_CAMERA; }
    *out = s->cfg;
    return RT_OK;
}

int rt_host_tree_stats_get(const rt_host_scene *s, rt_host_tree_stats *out) {
    if (!s || !out) { g_err = "null argument"; return RT_ERR_INVALID_ARG; }
    rt_host_tree_stats st{};
    st.n_nodes = s->desc.n_nodes;
    st.n_leaf_idx = s->desc.n_leaf_idx;
    st.max_depth_param = s->tree.maxDepthParam;
    for (uint32_t i = 0; i < st.n_nodes; ++i) {
        uint32_t w1 = s->tree.nodes[2 * i + 1];
        if ((w1 & 3u) == 3u) {
            uint32_t n = w1 >> 2;
            ++st.n_leaves;
            if (n == 0) ++st.n_empty_leaves;
            st.n_prim_refs += n;
            if (n > st.max_leaf_prims) st.max_leaf_prims = n;
        }
    }
    *out = st;
    return RT_OK;
}

int rt_host_camera(float fov_deg, const float eye[3], const float focus[3], const float up[3], int width, int height,
                   float raster_to_camera[16], float camera_to_world[16]) {
    if (!eye || !focus || !up || !raster_to_camera || !camera_to_world || width <= 0 || height <= 0) {
        g_err = "bad argument";
        return RT_ERR_INVALID_ARG;
    }
    M4 r2c, c2w;
    perspective_camera(fov_deg, {eye[0], eye[1], eye[2]}, {focus[0], focus[1], focus[2]}, {up[0], up[1], up[2]}, width, height, r2c, c2w);
    store_m4(r2c, raster_to_camera);
    store_m4(c2w, camera_to_world);
    return RT_OK;
}

int rt_host_camera_ray(const float raster_to_camera[16], const float camera_to_world[16], float film_x, float film_y,
                       float o[3], float d[3]) {
    if (!raster_to_camera || !camera_to_world || !o || !d) { g_err = "null argument"; return RT_ERR_INVALID_ARG; }
    // APerspectiveCamera::castingRay (reference src/core/Camera/Camera.cpp:87-95): the direction is
    // normalised three times (explicitly, by the Ray constructor, and again after the transform).
    M4 r2c = load_m4(raster_to_camera), c2w = load_m4(camera_to_world);
    V3 pCamera = xf_apply(r2c, {film_x, film_y, 0}, 1.0f);
    V3 dir = normalize(normalize(pCamera));
    V3 ow = xf_apply(c2w, {0, 0, 0}, 1.0f);
    V3 dw = normalize(xf_apply(c2w, dir, 0.0f));
    o[0] = ow.x; o[1] = ow.y; o[2] = ow.z;
    d[0] = dw.x; d[1] = dw.y; d[2] = dw.z;
    return RT_OK;
}

int rt_host_develop(const float *rgb_sum, const float *weight_sum, int width, int height, float film_scale, uint8_t *rgb8) {
    if (!rgb_sum || !weight_sum || !rgb8 || width <= 0 || height <= 0) { g_err = "bad argument"; return RT_ERR_INVALID_ARG; }
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
        const float *c = &rgb_sum[3 * i];
        // FilmTile contribution -> XYZ at merge (reference src/core/Object/Film.cpp:98-114, Utils/Color.h:15-20)
        float xyz[3];
        xyz[0] = 0.412453f * c[0] + 0.357580f * c[1] + 0.180423f * c[2];
        xyz[1] = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];
        xyz[2] = 0.019334f * c[0] + 0.119193f * c[1] + 0.950227f * c[2];
        // writeImageToFile (Film.cpp:122-157, Color.h:8-13, Base.h:50-55)
        float rgb[3];
        rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
        rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
        rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
        const float w = weight_sum[i];
        if (w != 0) {
            const float invWt = 1.0f / w;
            for (int k = 0; k < 3; ++k) rgb[k] = fmax_glm(0.0f, rgb[k] * invWt);
        }
        for (int k = 0; k < 3; ++k) {
            float v = rgb[k] + 1.0f * 0.0f;   // no splats on this path
            v *= film_scale;
            float g = (v <= 0.0031308f) ? 12.92f * v : 1.055f * std::pow(v, (float)(1.f / 2.4f)) - 0.055f;
            float b = 255.f * g + 0.5f;
            b = b < 0.f ? 0.f : (b > 255.f ? 255.f : b);
            rgb8[3 * i + k] = (uint8_t)b;
        }
    }
    return RT_OK;
}

}  // extern "C"
