// Host float math with the evaluation order of the scalar glm 0.9.9.8 paths the reference uses
// (reference src/vendor/header_only/glm/detail/func_geometric.inl:48-90, type_mat4x4.inl:561-648,
// func_matrix.inl:294-352, ext/matrix_transform.inl:10-86,122-142). Build with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rth {

struct V3 {
    float x, y, z;
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct V4 {
    float x, y, z, w;
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// Column-major 4x4: c[col][row], like glm::mat4.
struct M4 {
    V4 c[4];
    V4 &operator[](int i) { return c[i]; }
    const V4 &operator[](int i) const { return c[i]; }
};
inline M4 m4_identity() {
    M4 m;
    m.c[0] = {1, 0, 0, 0}; m.c[1] = {0, 1, 0, 0}; m.c[2] = {0, 0, 1, 0}; m.c[3] = {0, 0, 0, 1};
    return m;
}
inline M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = a.c[0] * b.c[j][0] + a.c[1] * b.c[j][1] + a.c[2] * b.c[j][2] + a.c[3] * b.c[j][3];
    return r;
}
inline V4 mul(const M4 &m, V4 v) {
    V4 a0 = m.c[0] * v.x + m.c[1] * v.y;
    V4 a1 = m.c[2] * v.z + m.c[3] * v.w;
    return a0 + a1;
}
inline M4 m4_translate(V3 v) {   // glm::translate(mat4(1), v)
    M4 m = m4_identity();
    m.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
    return m;
}
inline M4 m4_scale(V3 v) {       // glm::scale(mat4(1), v)
    M4 i = m4_identity(), m;
    m.c[0] = i.c[0] * v.x; m.c[1] = i.c[1] * v.y; m.c[2] = i.c[2] * v.z; m.c[3] = i.c[3];
    return m;
}
inline M4 m4_rotate(float angle, V3 axisIn) {   // glm::rotate(mat4(1), angle, axis)
    const float c = std::cos(angle), s = std::sin(angle);
    V3 axis = normalize(axisIn);
    V3 temp = (1.0f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    M4 m = m4_identity(), r;
    r.c[0] = m.c[0] * R[0][0] + m.c[1] * R[0][1] + m.c[2] * R[0][2];
    r.c[1] = m.c[0] * R[1][0] + m.c[1] * R[1][1] + m.c[2] * R[1][2];
    r.c[2] = m.c[0] * R[2][0] + m.c[1] * R[2][1] + m.c[2] * R[2][2];
    r.c[3] = m.c[3];
    return r;
}
inline M4 m4_inverse(const M4 &m) {   // glm::inverse(mat4), cofactor expansion
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    V4 f0{c00, c00, c02, c03}, f1{c04, c04, c06, c07}, f2{c08, c08, c10, c11};
    V4 f3{c12, c12, c14, c15}, f4{c16, c16, c18, c19}, f5{c20, c20, c22, c23};
    V4 v0{m[1][0], m[0][0], m[0][0], m[0][0]}, v1{m[1][1], m[0][1], m[0][1], m[0][1]};
    V4 v2{m[1][2], m[0][2], m[0][2], m[0][2]}, v3{m[1][3], m[0][3], m[0][3], m[0][3]};
    V4 i0 = v1 * f0 - v2 * f1 + v3 * f2;
    V4 i1 = v0 * f0 - v2 * f3 + v3 * f4;
    V4 i2 = v0 * f1 - v1 * f3 + v3 * f5;
    V4 i3 = v0 * f2 - v1 * f4 + v2 * f5;
    V4 sa{+1, -1, +1, -1}, sb{-1, +1, -1, +1};
    M4 inv;
    inv.c[0] = i0 * sa; inv.c[1] = i1 * sb; inv.c[2] = i2 * sa; inv.c[3] = i3 * sb;
    V4 row0{inv[0][0], inv[1][0], inv[2][0], inv[3][0]};
    V4 d0 = m.c[0] * row0;
    float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    float ood = 1.0f / d1;
    M4 r;
    for (int j = 0; j < 4; ++j) r.c[j] = inv.c[j] * ood;
    return r;
}
inline M4 m4_lookat_lh(V3 eye, V3 center, V3 up) {   // glm::lookAtLH (GLM_FORCE_LEFT_HANDED, Utils/Base.h:12)
    V3 f = normalize(center - eye);
    V3 s = normalize(cross(up, f));
    V3 u = cross(f, s);
    M4 r = m4_identity();
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = f.x; r[1][2] = f.y; r[2][2] = f.z;
    r[3][0] = -dot(s, eye); r[3][1] = -dot(u, eye); r[3][2] = -dot(f, eye);
    return r;
}

// Transform = matrix + inverse (reference src/core/Utils/Transform.h:10-63).
struct Xf {
    M4 m, inv;
};
inline Xf xf_identity() { return {m4_identity(), m4_identity()}; }
inline Xf xf_inverse(const Xf &t) { return {t.inv, t.m}; }
inline Xf xf_mul(const Xf &a, const Xf &b) { return {mul(a.m, b.m), mul(b.inv, a.inv)}; }   // Transform.cpp:78-81
inline Xf xf_translate(V3 d) { return {m4_translate(d), m4_translate(-d)}; }                   // Transform.cpp:10-15
inline Xf xf_scale(float x, float y, float z) { return {m4_scale({x, y, z}), m4_scale({1 / x, 1 / y, 1 / z})}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline Xf xf_rotate(float thetaDeg, V3 axis) {                                                  // Transform.cpp:45-50
    M4 m = m4_rotate(radians(thetaDeg), axis);
    return {m, m4_inverse(m)};
}
// Transform::operator()(Vec3, w) (Transform.h:75-88)
inline V3 xf_apply(const M4 &m, V3 p, float w) {
    V4 r = mul(m, V4{p.x, p.y, p.z, w});
    if (w == 0.f) return {r.x, r.y, r.z};
    if (r.w == 1) return {r.x, r.y, r.z};
    return V3{r.x, r.y, r.z} / r.w;
}

}  // namespace rth
