// Surface reconstruction, BSDFs, area lights and sampling warps on the device. Each function restates
// one reference function with the same float operation order:
//   surface           reference src/core/Shape/TriangleShape.cpp:387-452 (per-triangle constants are
//                     precomputed at upload: ss = normalize(dpdu), geometric normal, degeneracy)
//   BSDF frame, f, pdf, sample_f    src/core/Render/BSDF.h:30-54, BSDF.cpp:9-158
//   BxDFs             BSDF.cpp:162-246 (Lambertian, specular reflection with no-op or dielectric
//                     Fresnel, specular transmission)
//   materials         src/core/Material/LambertianMaterial.cpp:19-28, MirrorMaterial.cpp:19-29
//   area light        src/core/Render/Light.h:105-108, Light.cpp:109-128, src/core/Shape/Shape.cpp:28-67
//   light selection   src/core/Utils/LightDistrib.h:68-83, src/core/Utils/Math.h:638-657
//   warps             src/core/Render/Sampler.cpp:120-148, Sampler.h:71-89
#pragma once
#include "dtraverse.h"

enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum { BX_LAMBERT = 0, BX_SPEC_REFL_NOOP = 1, BX_SPEC_REFL_DIELECTRIC = 2, BX_SPEC_TRANS = 3 };

struct Surf {
    f3 p, n, ss;
    int mat, light;
};

// Surface part of ASphereShape::hit(ray, tHit, isect) (Shape/SphereShape.cpp:202-234) followed by
// Transform::operator()(SurfaceInteraction) (Utils/Transform.cpp:97-114), from the object-space ray and t.
// phi, u and v are not evaluated: nothing reads the uv of a sphere hit. Returns p, n (faced towards wo) and
// ss = normalize(dpdu) in world space.
RT_DEV void sphere_surface(const DevSphere &sp, f3 oObj, f3 dObj, float t, f3 &p, f3 &n, f3 &ss) {
    const float radius = sp.radius;
    f3 pHit = oObj + dObj * t;
    pHit = pHit * (radius / length3(pHit - mk3(0.f, 0.f, 0.f)));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    float cz = pHit.z / radius;
    cz = cz < -1 ? -1.f : (cz > 1 ? 1.f : cz);
    const float theta = acos_cr(cz);
    const float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    const float invZRadius = 1 / zRadius;
    const float cosPhi = pHit.x * invZRadius;
    const float sinPhi = pHit.y * invZRadius;
    float sinTheta, cosTheta;
    sincos_cr(theta, sinTheta, cosTheta);
    const f3 dpdu = mk3(-2 * RT_PI * pHit.y, 2 * RT_PI * pHit.x, 0.f);
    const f3 dpdv = (2 * RT_PI) * mk3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinTheta);
    const f3 nObj = normalize3(cross3(dpdu, dpdv));
    const f3 woObj = normalize3(-dObj);
    p = xf_apply(sp.o2w, pHit, 1.0f);
    n = normalize3(xf_apply(sp.o2w, nObj, 0.0f));
    const f3 wo = normalize3(xf_apply(sp.o2w, woObj, 0.0f));
    ss = normalize3(xf_apply(sp.o2w, dpdu, 0.0f));
    if (dot3(n, wo) < 0.f) n = -n;   // faceforward(isect.n, isect.wo)
}

// Rebuilds the surface at a hit. (ro, rd) is the ray that was traced (needed for spheres, whose hit record holds
// t in b0); triangles use the barycentrics only.
template <bool SPH>
RT_DEV void surface_at(const DevScene &sc, int prim, float b0, float b1, float b2, f3 ro, f3 rd, Surf &s) {
    const float4 q0 = sc.triHot[3 * prim + 0], q1 = sc.triHot[3 * prim + 1], q2 = sc.triHot[3 * prim + 2];
    if (SPH && q0.w == RT_PRIM_SPHERE) {
        const DevSphere &sp = sc.spheres[__float_as_int(q0.x)];
        const f3 oObj = xf_apply(sp.w2o, ro, 1.0f);
        const f3 dObj = normalize3(xf_apply(sp.w2o, rd, 0.0f));
        sphere_surface(sp, oObj, dObj, b0, s.p, s.n, s.ss);
        const float4 s4 = sc.triShade[5 * prim + 4];
        s.mat = __float_as_int(s4.x);
        s.light = __float_as_int(s4.y);
        return;
    }
    const float4 s0 = sc.triShade[5 * prim + 0], s1 = sc.triShade[5 * prim + 1], s2 = sc.triShade[5 * prim + 2];
    const float4 s3 = sc.triShade[5 * prim + 3], s4 = sc.triShade[5 * prim + 4];
    s.p = b0 * mk3(q0.x, q0.y, q0.z) + b1 * mk3(q1.x, q1.y, q1.z) + b2 * mk3(q2.x, q2.y, q2.z);
    s.ss = mk3(s0.w, s1.w, s2.w);
    f3 n = mk3(s3.x, s3.y, s3.z);
    if (s3.w != 0.f) {
        f3 ns = b0 * mk3(s0.x, s0.y, s0.z) + b1 * mk3(s1.x, s1.y, s1.z) + b2 * mk3(s2.x, s2.y, s2.z);
        if (dot3(ns, ns) > 0) n = normalize3(ns);
    }
    s.n = n;
    s.mat = __float_as_int(s4.x);
    s.light = __float_as_int(s4.y);
}

// At most two lobes (glass: specular reflection + specular transmission). Scalars, not arrays: runtime-indexed
// private arrays would be moved to LDS or scratch by the compiler.
struct DBsdf {
    f3 ns, ss, ts;
    float eta;
    int n;
    int kind0, kind1, type0, type1;
    f3 R;
    float etaA, etaB;
};
RT_DEV int bx_kind(const DBsdf &b, int i) { return i == 0 ? b.kind0 : b.kind1; }
RT_DEV int bx_type(const DBsdf &b, int i) { return i == 0 ? b.type0 : b.type1; }

// Material sets a kernel can be specialised for (chosen at scene upload).
enum { MATSET_ANY = 0, MATSET_LAMBERT = 1 };

template <int MATSET>
RT_DEV void bsdf_init(DBsdf &b, const Surf &s, const DevMaterial &m) {
    b.ns = s.n;
    b.ss = s.ss;
    b.ts = cross3(b.ns, b.ss);
    b.R = mk3(m.r, m.g, m.b);
    b.eta = 1;
    b.etaA = 1.f;
    b.etaB = 1.f;
    b.n = 0;
    b.kind0 = b.kind1 = 0;
    b.type0 = b.type1 = 0;
    if (black3(b.R)) return;
    if (MATSET == MATSET_LAMBERT || m.type == RT_MAT_LAMBERTIAN) {
        b.kind0 = BX_LAMBERT; b.type0 = BSDF_REFLECTION | BSDF_DIFFUSE; b.n = 1;
    } else if (m.type == RT_MAT_MIRROR) {
        b.kind0 = BX_SPEC_REFL_NOOP; b.type0 = BSDF_REFLECTION | BSDF_SPECULAR; b.n = 1;
    } else {
        b.eta = m.eta;
        b.etaB = m.eta;
        b.kind0 = BX_SPEC_REFL_DIELECTRIC; b.type0 = BSDF_REFLECTION | BSDF_SPECULAR;
        b.kind1 = BX_SPEC_TRANS; b.type1 = BSDF_TRANSMISSION | BSDF_SPECULAR;
        b.n = 2;
    }
}
RT_DEV bool bx_matches(int type, int flags) { return (type & flags) == type; }
RT_DEV int bsdf_num(const DBsdf &b, int flags) {
    int c = 0;
    for (int i = 0; i < b.n; ++i) if (bx_matches(bx_type(b, i), flags)) ++c;
    return c;
}
RT_DEV f3 w2l(const DBsdf &b, f3 v) { return mk3(dot3(v, b.ss), dot3(v, b.ts), dot3(v, b.ns)); }
RT_DEV f3 l2w(const DBsdf &b, f3 v) {
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
RT_DEV f3 bx_f(const DBsdf &b, int i) { return bx_kind(b, i) == BX_LAMBERT ? b.R * RT_INV_PI : mk3(0.f, 0.f, 0.f); }
RT_DEV float bx_pdf(const DBsdf &b, int i, f3 wo, f3 wi) {
    if (bx_kind(b, i) != BX_LAMBERT) return 0.f;
    return (wo.z * wi.z > 0) ? fabsf(wi.z) * RT_INV_PI : 0;
}
RT_DEV f3 bsdf_f(const DBsdf &b, f3 woW, f3 wiW, int flags) {
    f3 wi = w2l(b, wiW), wo = w2l(b, woW);
    (void)wi;
    f3 r = mk3(0.f, 0.f, 0.f);
    if (wo.z == 0) return r;
    const bool reflect = dot3(wiW, b.ns) * dot3(woW, b.ns) > 0;
    for (int i = 0; i < b.n; ++i)
        if (bx_matches(bx_type(b, i), flags) && ((reflect && (bx_type(b, i) & BSDF_REFLECTION)) || (!reflect && (bx_type(b, i) & BSDF_TRANSMISSION))))
            r = r + bx_f(b, i);
    return r;
}
RT_DEV float bsdf_pdf(const DBsdf &b, f3 woW, f3 wiW, int flags) {
    if (b.n == 0) return 0.f;
    f3 wo = w2l(b, woW), wi = w2l(b, wiW);
    if (wo.z == 0) return 0.f;
    float p = 0.f;
    int matching = 0;
    for (int i = 0; i < b.n; ++i)
        if (bx_matches(bx_type(b, i), flags)) { ++matching; p += bx_pdf(b, i, wo, wi); }
    return matching > 0 ? p / matching : 0.f;
}

RT_DEV void concentric_disk(float u0, float u1, float &dx, float &dy) {
    const float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { dx = 0; dy = 0; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = RT_PI_OVER_4 * (oy / ox); }
    else { r = oy; theta = RT_PI_OVER_2 - RT_PI_OVER_4 * (ox / oy); }
    float s, c;
    sincos_disk(theta, s, c);
    dx = r * c;
    dy = r * s;
}
RT_DEV f3 cosine_hemisphere(float u0, float u1) {
    float dx, dy;
    concentric_disk(u0, u1, dx, dy);
    const float z = sqrtf(gmax(0.f, 1 - dx * dx - dy * dy));
    return mk3(dx, dy, z);
}

RT_DEV float fr_dielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = cosThetaI < -1 ? -1.f : (cosThetaI > 1 ? 1.f : cosThetaI);
    const bool entering = cosThetaI > 0.f;
    if (!entering) { float tmp = etaI; etaI = etaT; etaT = tmp; cosThetaI = fabsf(cosThetaI); }
    const float sinThetaI = sqrtf(gmax(0.f, 1 - cosThetaI * cosThetaI));
    const float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    const float cosThetaT = sqrtf(gmax(0.f, 1 - sinThetaT * sinThetaT));
    const float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    const float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

// BxDF::sample_f of lobe i. Returns f; pdf = 0 means failure.
RT_DEV f3 bx_sample(const DBsdf &b, int i, f3 wo, f3 &wi, float u0, float u1, float &pdf) {
    const int k = bx_kind(b, i);
    if (k == BX_LAMBERT) {
        wi = cosine_hemisphere(u0, u1);
        if (wo.z < 0) wi.z *= -1;
        pdf = bx_pdf(b, i, wo, wi);
        return bx_f(b, i);
    }
    if (k == BX_SPEC_REFL_NOOP || k == BX_SPEC_REFL_DIELECTRIC) {
        wi = mk3(-wo.x, -wo.y, wo.z);
        pdf = 1;
        const float fr = (k == BX_SPEC_REFL_NOOP) ? 1.f : fr_dielectric(wi.z, b.etaA, b.etaB);
        return (mk3(fr, fr, fr) * b.R) / fabsf(wi.z);
    }
    // specular transmission
    const bool entering = wo.z > 0;
    const float etaI = entering ? b.etaA : b.etaB;
    const float etaT = entering ? b.etaB : b.etaA;
    // refract(wo, faceforward((0,0,1), wo), etaI / etaT, wi)  (reference src/core/Utils/Math.h:661-674)
    const f3 nrm = (wo.z < 0.f) ? mk3(-0.f, -0.f, -1.f) : mk3(0.f, 0.f, 1.f);
    const float eta = etaI / etaT;
    const float cosThetaI = dot3(nrm, wo);
    const float sin2ThetaI = gmax(0.f, 1 - cosThetaI * cosThetaI);
    const float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) { pdf = 0; return mk3(0.f, 0.f, 0.f); }
    const float cosThetaT = sqrtf(1 - sin2ThetaT);
    wi = eta * -wo + (eta * cosThetaI - cosThetaT) * nrm;
    pdf = 1;
    const float fr = fr_dielectric(wi.z, b.etaA, b.etaB);
    f3 ft = b.R * (mk3(1.f, 1.f, 1.f) - mk3(fr, fr, fr));
    ft = ft * ((etaI * etaI) / (etaT * etaT));
    return ft / fabsf(wi.z);
}

// BSDF::sample_f. sampledType is always written (the lobe's type, 0 on failure). When the local wo.z
// is 0 the function returns black and leaves pdf untouched, as the reference does.
RT_DEV f3 bsdf_sample(const DBsdf &b, f3 woW, f3 &wiW, float u0, float u1, float &pdf, int &sampledType, int type) {
    const int matching = bsdf_num(b, type);
    if (matching == 0) { pdf = 0; sampledType = 0; return mk3(0.f, 0.f, 0.f); }
    const int comp = min((int)floorf(u0 * matching), matching - 1);
    int sel = 0, count = comp;
    for (int i = 0; i < b.n; ++i)
        if (bx_matches(bx_type(b, i), type) && count-- == 0) { sel = i; break; }
    const float ur0 = gmin(u0 * matching - comp, RT_ONE_MINUS_EPS);
    f3 wi, wo = w2l(b, woW);
    if (wo.z == 0) { sampledType = 0; return mk3(0.f, 0.f, 0.f); }
    pdf = 0;
    sampledType = bx_type(b, sel);
    f3 f = bx_sample(b, sel, wo, wi, ur0, u1, pdf);
    if (pdf == 0) { sampledType = 0; return mk3(0.f, 0.f, 0.f); }
    wiW = l2w(b, wi);
    if (!(bx_type(b, sel) & BSDF_SPECULAR) && matching > 1)
        for (int i = 0; i < b.n; ++i)
            if (i != sel && bx_matches(bx_type(b, i), type)) pdf += bx_pdf(b, i, wo, wi);
    if (matching > 1) pdf /= matching;
    if (!(bx_type(b, sel) & BSDF_SPECULAR)) {
        const bool reflect = dot3(wiW, b.ns) * dot3(woW, b.ns) > 0;
        f = mk3(0.f, 0.f, 0.f);
        for (int i = 0; i < b.n; ++i)
            if (bx_matches(bx_type(b, i), type) && ((reflect && (bx_type(b, i) & BSDF_REFLECTION)) || (!reflect && (bx_type(b, i) & BSDF_TRANSMISSION))))
                f = f + bx_f(b, i);
    }
    return f;
}

RT_DEV float power_heuristic(float fPdf, float gPdf) {
    const float f = 1 * fPdf, g = 1 * gPdf;
    return (f * f) / (f * f + g * g);
}

// DiffuseAreaLight::L
RT_DEV f3 light_L(const DevLight &l, f3 n, f3 w) {
    return (l.twoSided || dot3(n, w) > 0) ? mk3(l.er, l.eg, l.eb) : mk3(0.f, 0.f, 0.f);
}

// Distribution1D::sampleDiscrete over the light selection distribution.
RT_DEV int sample_light(const DevScene &sc, float u, float &pdf) {
    const int size = (int)sc.nLights + 1;
    int first = 0, len = size;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (sc.lightCdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    pdf = (sc.lightFuncInt > 0) ? sc.lightFunc[offset] / (sc.lightFuncInt * (int)sc.nLights) : 0;
    return offset;
}

// RT_DEV coordinateSystem (Utils/Math.h:454-461)
RT_DEV void coordinate_system(f3 v1, f3 &v2, f3 &v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) v2 = mk3(-v1.z, 0.f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
    else v2 = mk3(0.f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
    v3 = cross3(v1, v2);
}

// ASphereShape::sample(ref, u, pdf) (Shape/SphereShape.cpp:42-113): uniform over the cone the sphere subtends, or
// uniform over the area (ASphereShape::sample(u, pdf), :28-40) when the reference point is inside.
RT_DEV void sphere_sample(const DevSphere &sp, float area, f3 refP, float u0, float u1, f3 &p, f3 &n, float &pdf) {
    const f3 pCenter = xf_apply(sp.o2w, mk3(0.f, 0.f, 0.f), 1.0f);
    const f3 dc0 = refP - pCenter;
    if (dot3(dc0, dc0) <= sp.radius * sp.radius) {
        const float z = 1 - 2 * u0;
        const float r = sqrtf(gmax(0.f, 1.f - z * z));
        const float phi = 2 * RT_PI * u1;
        float sphi, cphi;
        sincos_cr(phi, sphi, cphi);
        f3 pObj = mk3(0.f, 0.f, 0.f) + sp.radius * mk3(r * cphi, r * sphi, z);
        n = normalize3(xf_apply(sp.o2w, pObj, 0.0f));
        pObj = pObj * (sp.radius / length3(pObj - mk3(0.f, 0.f, 0.f)));
        p = xf_apply(sp.o2w, pObj, 1.0f);
        pdf = 1 / area;
        f3 w = p - refP;
        if (dot3(w, w) == 0) pdf = 0;
        else {
            w = normalize3(w);
            const f3 dd = refP - p;
            pdf *= dot3(dd, dd) / absdot3(n, -w);
        }
        if (isinf(pdf)) pdf = 0.f;
        return;
    }
    const float dc = length3(refP - pCenter);
    const float invDc = 1 / dc;
    const f3 wc = (pCenter - refP) * invDc;
    f3 wcX, wcY;
    coordinate_system(wc, wcX, wcY);
    const float sinThetaMax = sp.radius * invDc;
    const float sinThetaMax2 = sinThetaMax * sinThetaMax;
    const float invSinThetaMax = 1 / sinThetaMax;
    const float cosThetaMax = sqrtf(gmax(0.f, 1.0f - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u0 + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u0;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    const float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(gmax(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    const float sinAlpha = sqrtf(gmax(0.f, 1.f - cosAlpha * cosAlpha));
    const float phi = u1 * 2 * RT_PI;
    float sphi, cphi;
    sincos_cr(phi, sphi, cphi);
    // sphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc) (Utils/Math.h:468-472)
    const f3 nWorld = (sinAlpha * cphi) * (-wcX) + (sinAlpha * sphi) * (-wcY) + cosAlpha * (-wc);
    p = pCenter + sp.radius * mk3(nWorld.x, nWorld.y, nWorld.z);
    n = nWorld;
    pdf = 1 / (2 * RT_PI * (1 - cosThetaMax));
}

// DiffuseAreaLight::sample_Li for a triangle or sphere light. Returns Li; pdf = 0 means no sample.
// pShape receives the sampled point (the shadow ray's target).
template <bool SPH>
RT_DEV f3 light_sample_Li(const DevScene &sc, const DevLight &l, f3 refP, float u0, float u1, f3 &wi, float &pdf, f3 &pShape) {
    const float4 q0 = sc.triHot[3 * l.prim + 0], q1 = sc.triHot[3 * l.prim + 1], q2 = sc.triHot[3 * l.prim + 2];
    f3 p, n;
    if (SPH && q0.w == RT_PRIM_SPHERE) {
        sphere_sample(sc.spheres[__float_as_int(q0.x)], l.area, refP, u0, u1, p, n, pdf);
    } else {
        const f3 p0 = mk3(q0.x, q0.y, q0.z), p1 = mk3(q1.x, q1.y, q1.z), p2 = mk3(q2.x, q2.y, q2.z);
        const float su0 = sqrtf(u0);
        const float b0 = 1 - su0, b1 = u1 * su0;
        p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        n = normalize3(cross3(p1 - p0, p2 - p0));
        pdf = 1 / l.area;
        f3 w = p - refP;
        if (dot3(w, w) == 0) pdf = 0;
        else {
            w = normalize3(w);
            const f3 dd = refP - p;
            pdf *= dot3(dd, dd) / absdot3(n, -w);
            if (isinf(pdf)) pdf = 0.f;
        }
    }
    const f3 d2 = p - refP;
    if (pdf == 0 || dot3(d2, d2) == 0) { pdf = 0; return mk3(0.f, 0.f, 0.f); }
    wi = normalize3(p - refP);
    pShape = p;
    return light_L(l, n, -wi);
}

// DiffuseAreaLight::pdf_Li -> Shape::pdf(ref, wi): one closest-flavour triangle test against the light's shape.
template <bool SPH>
RT_DEV float light_pdf_Li(const DevScene &sc, const DevLight &l, f3 refP, f3 wi) {
    const float4 q0 = sc.triHot[3 * l.prim + 0], q1 = sc.triHot[3 * l.prim + 1], q2 = sc.triHot[3 * l.prim + 2];
    const f3 rd = normalize3(wi);   // Interaction::spawnRay -> Ray constructor
    Surf s;
    if (SPH && q0.w == RT_PRIM_SPHERE) {
        // ASphereShape::pdf (Shape/SphereShape.cpp:115-127)
        const DevSphere &sp = sc.spheres[__float_as_int(q0.x)];
        const f3 pCenter = xf_apply(sp.o2w, mk3(0.f, 0.f, 0.f), 1.0f);
        const f3 dc = refP - pCenter;
        if (!(dot3(dc, dc) <= sp.radius * sp.radius)) {
            const float sinThetaMax2 = sp.radius * sp.radius / dot3(dc, dc);
            const float cosThetaMax = sqrtf(gmax(0.f, 1 - sinThetaMax2));
            return 1 / (2 * RT_PI * (1 - cosThetaMax));   // uniformConePdf
        }
        f3 oObj, dObj;
        float t;
        if (!sphere_test(sp, refP, rd, RT_INF, oObj, dObj, t)) return 0;
        sphere_surface(sp, oObj, dObj, t, s.p, s.n, s.ss);
    } else {
        RayCtx r;
        ray_setup(r, refP, rd);
        float t, b0, b1, b2;
        if (!tri_test<false>(r, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z), RT_INF, t, b0, b1, b2)) return 0;
        if (q0.w != 0.f) return 0;
        surface_at<false>(sc, l.prim, b0, b1, b2, refP, rd, s);
    }
    const f3 dd = refP - s.p;
    float pdf = dot3(dd, dd) / (absdot3(s.n, -wi) * l.area);
    if (isinf(pdf)) pdf = 0.f;
    return pdf;
}
