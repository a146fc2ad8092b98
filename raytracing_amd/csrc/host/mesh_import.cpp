// OBJ import that reproduces what the reference obtains from assimp for
//   aiProcess_Triangulate | GenSmoothNormals | FlipUVs | FixInfacingNormals | OptimizeMeshes
// (reference src/core/Shape/TriangleShape.cpp:76-78) on single-object OBJ files:
//   * number parsing of assimp's fast_atoreal_move (src/vendor/assimp/include/assimp/fast_atof.h:267-351):
//     integer part as float, fraction accumulated in double, rounded to float, added in float;
//   * one fresh vertex per face corner, in face order (code/AssetLib/Obj/ObjFileImporter.cpp:443-527);
//   * quads fanned from corner 0 unless a concave corner exists (code/PostProcessing/TriangulateProcess.cpp:317-369);
//     larger polygons are fanned from corner 0 (assimp ear-clips them; only convex input is exact here);
//   * FixInfacingNormals for meshes that bring their own normals (code/PostProcessing/FixNormalsStep.cpp:103-184);
//   * smooth normals: per-face normal written to the face's vertices (later faces overwrite), then
//     averaged over vertices closer than 1e-4 * bbox diagonal (code/PostProcessing/GenVertexNormalsProcess.cpp:111-233).
#include "mesh_import.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace rth {

namespace {

// assimp strtoul10_64 with a digit cap; returns the value and how many digits were consumed.
uint64_t parse_digits(const char *&c, unsigned maxDigits, unsigned &used) {
    uint64_t v = 0;
    used = 0;
    while (*c >= '0' && *c <= '9') {
        if (used < maxDigits) {
            v = v * 10 + (uint64_t)(*c - '0');
            ++used;
        }
        ++c;   // digits past the cap are skipped, as assimp does
    }
    return v;
}

float assimp_atof(const char *c) {
    static const double tbl[16] = {0.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001,
                                   0.000000001, 0.0000000001, 0.00000000001, 0.000000000001, 0.0000000000001,
                                   0.00000000000001, 0.000000000000001};
    float f = 0;
    bool neg = (*c == '-');
    if (neg || *c == '+') ++c;
    unsigned used;
    if (*c != '.') f = static_cast<float>(parse_digits(c, 32, used));
    if (*c == '.' && c[1] >= '0' && c[1] <= '9') {
        ++c;
        double pl = static_cast<double>(parse_digits(c, 15, used));
        pl *= tbl[used];
        f += static_cast<float>(pl);
    } else if (*c == '.') {
        ++c;
    }
    if (*c == 'e' || *c == 'E') {
        ++c;
        bool eneg = (*c == '-');
        if (eneg || *c == '+') ++c;
        float e = static_cast<float>(parse_digits(c, 32, used));
        if (eneg) e = -e;
        f *= std::pow(10.0f, e);
    }
    return neg ? -f : f;
}

struct Corner { int v = -1, t = -1, n = -1; };

// aiVector3 helpers (include/assimp/vector3.inl:94-164,296-298)
inline float ai_len(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 ai_cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 ai_div(V3 a, float f) {
    if (f == 0.0f) return a;
    const float inv = 1.0f / f;
    return {a.x * inv, a.y * inv, a.z * inv};
}
inline V3 ai_normalize_safe(V3 a) {
    float l = ai_len(a);
    return l > 0.0f ? ai_div(a, l) : a;
}
inline float ai_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

}  // namespace

void import_obj(const std::string &path, ImportedMesh &out) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open OBJ file: " + path);
    std::vector<V3> V, N;
    std::vector<V3> T;
    std::vector<std::vector<Corner>> faces;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v" || tag == "vn" || tag == "vt") {
            std::string a, b, c;
            ls >> a >> b >> c;
            V3 p{a.empty() ? 0 : assimp_atof(a.c_str()), b.empty() ? 0 : assimp_atof(b.c_str()), c.empty() ? 0 : assimp_atof(c.c_str())};
            (tag == "v" ? V : tag == "vn" ? N : T).push_back(p);
        } else if (tag == "f") {
            std::vector<Corner> f;
            std::string tok;
            while (ls >> tok) {
                Corner cr;
                int part = 0;
                size_t s = 0;
                while (s <= tok.size() && part < 3) {
                    size_t e = tok.find('/', s);
                    if (e == std::string::npos) e = tok.size();
                    if (e > s) {
                        int idx = atoi(tok.substr(s, e - s).c_str());
                        const int count = (int)(part == 0 ? V.size() : part == 1 ? T.size() : N.size());
                        idx = idx > 0 ? idx - 1 : count + idx;
                        (part == 0 ? cr.v : part == 1 ? cr.t : cr.n) = idx;
                    }
                    s = e + 1;
                    ++part;
                }
                if (cr.v < 0 || cr.v >= (int)V.size()) throw std::runtime_error("OBJ: vertex index out of range in " + path);
                f.push_back(cr);
            }
            if (f.size() >= 3) faces.push_back(f);
        }
    }
    if (faces.empty()) throw std::runtime_error("OBJ: no faces in " + path);

    // One vertex per face corner.
    bool hasN = !N.empty(), hasT = !T.empty();
    for (auto &f : faces)
        for (auto &c : f) {
            if (c.n < 0 || c.n >= (int)N.size()) hasN = false;
            if (c.t < 0 || c.t >= (int)T.size()) hasT = false;
        }
    out.position.clear(); out.normal.clear(); out.uv.clear(); out.indices.clear();
    std::vector<std::vector<unsigned>> polys;
    for (auto &f : faces) {
        std::vector<unsigned> poly;
        for (auto &c : f) {
            poly.push_back((unsigned)out.position.size());
            out.position.push_back(V[c.v]);
            if (hasN) out.normal.push_back(N[c.n]);
            if (hasT) { out.uv.push_back(T[c.t].x); out.uv.push_back(1.0f - T[c.t].y); }   // aiProcess_FlipUVs
        }
        polys.push_back(poly);
    }
    const std::vector<V3> &P = out.position;

    // Triangulate.
    std::vector<std::array<unsigned, 3>> tris;
    for (auto &poly : polys) {
        if (poly.size() == 3) {
            tris.push_back({poly[0], poly[1], poly[2]});
        } else if (poly.size() == 4) {
            unsigned start = 0;
            for (unsigned i = 0; i < 4; ++i) {
                V3 v0 = P[poly[(i + 3) % 4]], v1 = P[poly[(i + 2) % 4]], v2 = P[poly[(i + 1) % 4]], v = P[poly[i]];
                V3 left = v0 - v, diag = v1 - v, right = v2 - v;
                left = ai_div(left, ai_len(left));
                diag = ai_div(diag, ai_len(diag));
                right = ai_div(right, ai_len(right));
                const float angle = std::acos(ai_dot(left, diag)) + std::acos(ai_dot(right, diag));
                if (angle > 3.141592653589793238462643383279f) { start = i; break; }
            }
            tris.push_back({poly[start], poly[(start + 1) % 4], poly[(start + 2) % 4]});
            tris.push_back({poly[start], poly[(start + 2) % 4], poly[(start + 3) % 4]});
        } else {
            for (size_t k = 1; k + 1 < poly.size(); ++k) tris.push_back({poly[0], poly[(unsigned)k], poly[(unsigned)k + 1]});
        }
    }

    const size_t nv = P.size();
    if (hasN) {
        // FixInfacingNormals: compare the bbox of (vertex + normal) with the bbox of the vertices.
        V3 mn0{1e10f, 1e10f, 1e10f}, mn1 = mn0, mx0{-1e10f, -1e10f, -1e10f}, mx1 = mx0;
        for (size_t i = 0; i < nv; ++i) {
            for (int a = 0; a < 3; ++a) {
                mn1[a] = std::min(mn1[a], P[i][a]);
                mx1[a] = std::max(mx1[a], P[i][a]);
            }
            V3 w = P[i] + out.normal[i];
            for (int a = 0; a < 3; ++a) {
                mn0[a] = std::min(mn0[a], w[a]);
                mx0[a] = std::max(mx0[a], w[a]);
            }
        }
        const float d0x = mx0.x - mn0.x, d0y = mx0.y - mn0.y, d0z = mx0.z - mn0.z;
        const float d1x = mx1.x - mn1.x, d1y = mx1.y - mn1.y, d1z = mx1.z - mn1.z;
        bool flip = true;
        if ((d0x > 0.0f) != (d1x > 0.0f)) flip = false;
        if ((d0y > 0.0f) != (d1y > 0.0f)) flip = false;
        if ((d0z > 0.0f) != (d1z > 0.0f)) flip = false;
        const float d1yz = d1y * d1z;
        if (flip && d1x < 0.05f * std::sqrt(d1yz)) flip = false;
        if (flip && d1y < 0.05f * std::sqrt(d1z * d1x)) flip = false;
        if (flip && d1z < 0.05f * std::sqrt(d1y * d1x)) flip = false;
        if (flip && !(std::fabs(d0x * d0y * d0z) < std::fabs(d1x * d1yz))) flip = false;
        if (flip) {
            for (auto &n : out.normal) n = n * -1.0f;
            for (auto &t : tris) std::swap(t[0], t[2]);
        }
    } else {
        // GenSmoothNormals.
        std::vector<V3> fn(nv, V3{0, 0, 0});
        for (auto &t : tris) {
            V3 a = P[t[0]], b = P[t[1]], c = P[t[2]];
            V3 n = ai_normalize_safe(ai_cross(b - a, c - a));
            fn[t[0]] = n; fn[t[1]] = n; fn[t[2]] = n;
        }
        V3 mn = P[0], mx = P[0];
        for (size_t i = 1; i < nv; ++i)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], P[i][a]);
                mx[a] = std::max(mx[a], P[i][a]);
            }
        const float eps = ai_len(mx - mn) * 1e-4f;
        const float eps2 = eps * eps;
        // Sort along a fixed direction (assimp's SpatialSort plane normal) to find neighbours.
        V3 pn = ai_div(V3{0.8523f, 0.34321f, 0.5736f}, ai_len(V3{0.8523f, 0.34321f, 0.5736f}));
        std::vector<std::pair<float, unsigned>> order(nv);
        for (size_t i = 0; i < nv; ++i) order[i] = {ai_dot(P[i], pn), (unsigned)i};
        std::sort(order.begin(), order.end());
        std::vector<size_t> rank(nv);
        for (size_t k = 0; k < nv; ++k) rank[order[k].second] = k;
        out.normal.assign(nv, V3{0, 0, 0});
        std::vector<char> had(nv, 0);
        std::vector<unsigned> found;
        for (size_t i = 0; i < nv; ++i) {
            if (had[i]) continue;
            found.clear();
            const float di = order[rank[i]].first;
            size_t lo = rank[i];
            while (lo > 0 && order[lo - 1].first > di - eps) --lo;
            for (size_t k = lo; k < nv && order[k].first < di + eps; ++k) {
                V3 dlt = P[order[k].second] - P[i];
                if (dlt.x * dlt.x + dlt.y * dlt.y + dlt.z * dlt.z < eps2) found.push_back(order[k].second);
            }
            V3 acc{0, 0, 0};
            for (unsigned j : found) acc = acc + fn[j];
            acc = ai_normalize_safe(acc);
            for (unsigned j : found) { out.normal[j] = acc; had[j] = 1; }
        }
    }
    for (auto &t : tris) { out.indices.push_back((int)t[0]); out.indices.push_back((int)t[1]); out.indices.push_back((int)t[2]); }
}

}  // namespace rth
