// grt_device.h — device-side scene layout and the per-ray device functions (gfx950 only).
//
// What each function restates (reference paths relative to the upstream repo):
//   make_primary      Camera.Ray                       pkg/tracer/camera.go:34-39
//   aabb_hit32        AABB.Hit                         pkg/tracer/aabb.go:25-45
//   sphere_hit32      Sphere.Hit                       pkg/tracer/sphere.go:23-46
//   intersect32       Scene.intersect                  pkg/tracer/scene.go:169-181
//   surface32         Sphere.Surface / AABB.Surface    sphere.go:56-58 / aabb.go:55-72
//   sample_hemi32     geom.SampleHemisphereNormal      pkg/geom/vec3.go:202-218
//   reflect32/refract32  Vec3.Reflect / Vec3.Refract   vec3.go:129-131 / 134-159
//   path_step         one level of Scene.trace         scene.go:237-349
//   *64               the same in float64 with the reference's operation order (no FMA
//                     contraction), used to re-test lanes whose FP32 decision margin is small
//
// FP32 formulation notes (decisions equal the reference's away from ties; ties are caught by
// ambiguous32 and re-tested in FP64 by the parity kernel):
//   - slabs are evaluated as fma(bound, 1/D, -O/D): 6 FMA per box instead of 6 sub + 6 div;
//     a wave whose rays have a zero direction component takes the CAREFUL path that keeps
//     the reference's (bound-O)/D form and Go's NaN/Inf rules for math.Min/Max;
//   - the sphere discriminant uses l = oc - (oc.D/|D|^2) D, disc/|D|^2 = R^2 - |l|^2, which
//     does not cancel catastrophically for distant origins (the naive halfB^2 - a*c loses
//     ~3 digits at |oc| ~ 1000);
//   - AABB.Surface's epsilon cascade is evaluated on a hit point whose entry-axis coordinate
//     is snapped to the face (in float64 it lies within 1e-13 of it; in FP32 the 1e-4
//     epsilon is the size of an ulp at 555 and the literal cascade would misfire).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grt_rng.h"

#define GRT_DEV __device__ __forceinline__

struct f3 {
  float x, y, z;
};
GRT_DEV f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
GRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
GRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
GRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
GRT_DEV f3 operator*(f3 a, float k) { return mk3(a.x * k, a.y * k, a.z * k); }
GRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
GRT_DEV float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
GRT_DEV f3 cross3(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
GRT_DEV f3 fma3(f3 a, float k, f3 c) { return mk3(fmaf(a.x, k, c.x), fmaf(a.y, k, c.y), fmaf(a.z, k, c.z)); }
GRT_DEV float rcp32(float x) { return __builtin_amdgcn_rcpf(x); }
GRT_DEV float rsq32(float x) { return __builtin_amdgcn_rsqf(x); }
GRT_DEV float sqrt32(float x) { return __builtin_amdgcn_sqrtf(x); }
GRT_DEV f3 unit3(f3 a) { return a * rsq32(dot3(a, a)); }
GRT_DEV float comp(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

// ---- scene as the kernels see it -----------------------------------------------------
// prims[2i]   = (a.x, a.y, a.z, radius)          sphere centre | box MinBound
// prims[2i+1] = (b.x, b.y, b.z, bits)            box MaxBound  | sphere: b.x = radius^2
//               bits (int): bit0 kind (1 = AABB), bit1 emissive (Emittance > 0), bit2 transparent,
//               mat << 8
// mats[3m]    = (r, g, b, refr_index)
// mats[3m+1]  = (reflectivity, roughness, emittance, flags)   flags: 1 transparent,
//               2 lambert, 4 normal
// mats[3m+2]  = what Refract (vec3.go:134-159) derives from the index alone, in float64 on the host:
//               (ratio = 1/RefrIndex, 1/ratio, Schlick r0 of ratio, r0 of 1/ratio)
// lights[2k]  = (Pos().xyz, prim index as a float), lights[2k+1] = (Color.rgb, prim index as int bits)   scene.go:42-46;
//               in ascending prim index (NewScene appends in object order)
struct KCamera {
  float ox, oy, oz;  // origin
  float lx, ly, lz;  // lowerLeft - origin (folded on the host in float64)
  float hx, hy, hz;  // horizontal
  float vx, vy, vz;  // vertical
};
struct BvhGrid {
  f3 gmin, gs, grs;  // grt_bvh.h: origin, cell size and 1 / cell size of the hierarchy's 16-bit grid
};
struct KScene {
  KCamera cam;
  int W, H, max_depth, n_prims;
  float bias;
  uint32_t seed32;
  int n_lights;
  uint32_t box_mask;  // bit i set = object i is an AABB (objects 0..31)
  const float4* __restrict__ prims;
  const float4* __restrict__ mats;
  const float4* __restrict__ lights;
  // scenes above GRT_MAX_LINEAR_PRIMS objects (or grt_scene_desc.accel = GRT_ACCEL_BVH): grt_bvh.h
  const uint4* __restrict__ bvh_nodes;
  const float4* __restrict__ bvh_prims;
  BvhGrid bvh_grid;
  int bvh_depth;
#ifdef GRT_COUNT_DEPTH  // experiment (tools/exp_c5_treelets.py): depth of the node in every slot of bvh_nodes
  const uint8_t* __restrict__ bvh_slot_depth;
#endif
  int n_mats;
  // tile_live[ty * tiles_x + tx] == 0: no camera ray through 8x8 tile (tx, ty) can reach the scene's
  // bounding box (frustum-plane test in float64 on the host, grt_api.hip); its pixels are black
  const uint8_t* __restrict__ tile_live;
};

#define GRT_MAT_STRIDE 3
#define GRT_MAT_TRANSPARENT 1
#define GRT_MAT_LAMBERT 2
#define GRT_MAT_NORMAL 4
// which branch of trace() a material takes (scene.go:274-305), bits 4-6 of the flags word; bits 8-10: of irradiance() (scene.go:207-228)
#define GRT_CLS_NORMAL 0
#define GRT_CLS_EMIT 1
#define GRT_CLS_LAMBERT 2
#define GRT_CLS_METAL 3
#define GRT_CLS_GLASS 4
#define GRT_CLS_DIFFUSE 5
#define GRT_CLS_MISS 6
#define GRT_MAT_CLASS(flags) (((flags) >> 4) & 7)
#define GRT_MAT_VIEW_CLASS(flags) (((flags) >> 8) & 7)

struct Ray32 {
  f3 O, D, invD, OinvD;
  float a, inva;  // |D|^2 and its reciprocal
};
GRT_DEV void ray_setup(Ray32& r, f3 O, f3 D) {
  r.O = O;
  r.D = D;
  r.invD = mk3(rcp32(D.x), rcp32(D.y), rcp32(D.z));
  r.OinvD = O * r.invD;
  r.a = dot3(D, D);
  r.inva = rcp32(r.a);
}
// Ray with |D| = 1: a = inva = 1 fold away inside sphere_hit32.
GRT_DEV void ray_setup_unit(Ray32& r, f3 O, f3 D) {
  r.O = O;
  r.D = D;
  r.invD = mk3(rcp32(D.x), rcp32(D.y), rcp32(D.z));
  r.OinvD = O * r.invD;
  r.a = 1.0f;
  r.inva = 1.0f;
}
// A unit-length ray that keeps no reciprocals (the hierarchy walk of k_pathtrace_bvh: box OBJECTS
// recompute them, which keeps 6 VGPRs free across the walk; the values are those of ray_setup_unit).
struct UnitRay {
  f3 O, D;
};
GRT_DEV const Ray32& full_ray(const Ray32& r) { return r; }
GRT_DEV Ray32 full_ray(const UnitRay& u) {
  Ray32 r;
  ray_setup_unit(r, u.O, u.D);
  return r;
}
GRT_DEV float ray_inva(const Ray32& r) { return r.inva; }
GRT_DEV float ray_inva(const UnitRay&) { return 1.0f; }
GRT_DEV bool ray_has_zero(const UnitRay& r) { return (r.D.x == 0.0f) | (r.D.y == 0.0f) | (r.D.z == 0.0f); }
GRT_DEV bool ray_has_zero(const Ray32& r) { return (r.D.x == 0.0f) | (r.D.y == 0.0f) | (r.D.z == 0.0f); }

// Camera.Ray (camera.go:34-39): D = lowerLeft + horizontal*u + vertical*v - origin, unnormalised.
// FP32 version used by the renderer (the pixel jitter is random anyway).
GRT_DEV void make_primary(const KScene& S, int x, int y, float ju, float jv, f3& O, f3& D) {
  float u = ((float)x + ju) * (1.0f / (float)S.W);
  float v = ((float)y + jv) * (1.0f / (float)S.H);
  O = mk3(S.cam.ox, S.cam.oy, S.cam.oz);
  D = mk3(fmaf(S.cam.vx, v, fmaf(S.cam.hx, u, S.cam.lx)), fmaf(S.cam.vy, v, fmaf(S.cam.hy, u, S.cam.ly)),
          fmaf(S.cam.vz, v, fmaf(S.cam.hz, u, S.cam.lz)));
}

// Go math.Min/Max in FP32 (vec3.go:104-111): +-Inf beats NaN, else NaN propagates.
GRT_DEV float gomax32(float x, float y) {
  if ((x == INFINITY) | (y == INFINITY)) return INFINITY;
  if ((x != x) | (y != y)) return NAN;
  return x > y ? x : y;
}
GRT_DEV float gomin32(float x, float y) {
  if ((x == -INFINITY) | (y == -INFINITY)) return -INFINITY;
  if ((x != x) | (y != y)) return NAN;
  return x < y ? x : y;
}

// CAREFUL is chosen per WAVE (some lane's ray has a zero direction component); the form is chosen per LANE: a lane
// whose own ray has no zero component takes the usual form inside a careful wave too, so what a lane computes never
// depends on which other rays share its wave (with 16-bit uniforms a Lambert bounce leaves exactly along the normal —
// SampleHemisphereNormal at r2 = 0 — once in 65 536 times, and the hierarchy kernel's lanes meet different wave-mates
// under different scheduling thresholds).
template <bool CAREFUL>
GRT_DEV float aabb_hit32(float4 pa, float4 pb, const Ray32& r, float tmin, float tmax) {
  if (!CAREFUL || !ray_has_zero(r)) {
    float x0 = fmaf(pa.x, r.invD.x, -r.OinvD.x), x1 = fmaf(pb.x, r.invD.x, -r.OinvD.x);
    float y0 = fmaf(pa.y, r.invD.y, -r.OinvD.y), y1 = fmaf(pb.y, r.invD.y, -r.OinvD.y);
    float z0 = fmaf(pa.z, r.invD.z, -r.OinvD.z), z1 = fmaf(pb.z, r.invD.z, -r.OinvD.z);
    float t0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float t1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    float lo = fmaxf(tmin, t0), hi = fminf(tmax, t1);
    return hi > lo ? t0 : -1.0f;  // aabb.go:41-44: returns the ENTRY distance t0
  } else {
    float x0 = (pa.x - r.O.x) * r.invD.x, x1 = (pb.x - r.O.x) * r.invD.x;
    float y0 = (pa.y - r.O.y) * r.invD.y, y1 = (pb.y - r.O.y) * r.invD.y;
    float z0 = (pa.z - r.O.z) * r.invD.z, z1 = (pb.z - r.O.z) * r.invD.z;
    float t0 = gomax32(gomax32(gomin32(x0, x1), gomin32(y0, y1)), gomin32(z0, z1));
    float t1 = gomin32(gomin32(gomax32(x0, x1), gomax32(y0, y1)), gomax32(z0, z1));
    float lo = tmin, hi = tmax;
    if (t0 > lo) lo = t0;
    if (t1 < hi) hi = t1;
    if (hi <= lo) return -1.0f;
    return t0;
  }
}

// Sphere.Hit (sphere.go:23-46) for tmin >= 0 (grt_scene_create refuses a negative bias).  The reference returns the near root
// t1 when it lies inside (tMin, tMax), else the far root t2 when that does, and intersect() keeps a result > 0
// (scene.go:174).  With t1 <= t2 that is: t = t1 > tMin ? t1 : t2, accepted when tMin < t < tMax — if t1 > tMin but
// t1 >= tMax, t2 >= tMax fails too; t > tMin >= 0 implies the caller's t > 0.  Three compares and one select.
GRT_DEV float sphere_hit32(float4 pa, float4 pb, const Ray32& r, float tmin, float tmax) {
  f3 oc = mk3(r.O.x - pa.x, r.O.y - pa.y, r.O.z - pa.z);
  float nk = -dot3(oc, r.D) * r.inva;  // -halfB / a
  f3 l = fma3(r.D, nk, oc);            // component of oc perpendicular to D
  float s2 = (pb.x - dot3(l, l)) * r.inva;  // disc / a^2
  float sq = sqrt32(s2);              // NaN when disc < 0: every window test then fails
  float t1 = nk - sq, t2 = nk + sq;
  const float t = t1 > tmin ? t1 : t2;
  return ((t > tmin) & (t < tmax)) ? t : -1.0f;
}

// Scene.intersect (scene.go:169-181): nearest hit over all objects in order, tMax shrinking,
// a candidate accepted when ht > 0.  Returns the object index or -1.
// tp: the object records, in HBM (S.prims, read through the scalar cache) or an LDS copy of them
template <bool CAREFUL>
GRT_DEV int intersect_tab_t(const float4* __restrict__ tp, int n_prims, float bias, const Ray32& r, float& t_out) {
  float tbest = 3.402823466e38f;
  int best = -1;
#pragma unroll 4
  for (int i = 0; i < n_prims; ++i) {
    float4 pa = tp[2 * i], pb = tp[2 * i + 1];
    float ht;
    if (__float_as_int(pb.w) & 1)
      ht = aabb_hit32<CAREFUL>(pa, pb, r, bias, tbest);
    else
      ht = sphere_hit32(pa, pb, r, bias, tbest);
    if (ht > 0.0f) {
      best = i;
      tbest = ht;
    }
  }
  t_out = tbest;
  return best;
}
// The same result from a copy of the records sorted by kind (boxes first, each kind in index order,
// last word = index << 3 | flags as in the hierarchy's leaf records): two loops without a kind
// branch.  Each loop is the reference's loop restricted to one kind; by the order-free form of the
// result (grt_bvh.h) the winner of the whole loop is the box loop's winner when that is a box entered
// within the bias, else the nearer of the two winners, the lower index on equal distance.
GRT_DEV int intersect_sorted(const float4* __restrict__ ts, int n_boxes, int n_prims, float bias, const Ray32& r, float& t_out) {
  float tb = 3.402823466e38f, tq = 3.402823466e38f;
  int ib = -1, iq = -1;
#pragma unroll 4
  for (int i = 0; i < n_boxes; ++i) {
    const float4 pa = ts[2 * i], pb = ts[2 * i + 1];
    const float ht = aabb_hit32<false>(pa, pb, r, bias, tb);
    if (ht > 0.0f) ib = (int)((uint32_t)__float_as_int(pb.w) >> 3), tb = ht;
  }
#pragma unroll 4
  for (int i = n_boxes; i < n_prims; ++i) {
    const float4 pa = ts[2 * i], pb = ts[2 * i + 1];
    const float ht = sphere_hit32(pa, pb, r, bias, tq);
    if (ht > 0.0f) iq = (int)((uint32_t)__float_as_int(pb.w) >> 3), tq = ht;
  }
  const bool box = (ib >= 0) & ((tb <= bias) | (tb < tq) | ((tb == tq) & (ib < iq)) | (iq < 0));
  t_out = box ? tb : tq;
  return box ? ib : iq;
}
template <bool CAREFUL>
GRT_DEV int intersect32_t(const KScene& S, const Ray32& r, float& t_out) {
  float tbest = 3.402823466e38f;
  int best = -1;
  for (int i = 0; i < S.n_prims; ++i) {
    float4 pa = S.prims[2 * i], pb = S.prims[2 * i + 1];
    float ht;
    if (__float_as_int(pb.w) & 1)
      ht = aabb_hit32<CAREFUL>(pa, pb, r, S.bias, tbest);
    else
      ht = sphere_hit32(pa, pb, r, S.bias, tbest);
    if (ht > 0.0f) {
      best = i;
      tbest = ht;
    }
  }
  t_out = tbest;
  return best;
}
GRT_DEV int intersect32(const KScene& S, const Ray32& r, float& t_out) {
  if (__builtin_expect(__any(ray_has_zero(r)), 0)) return intersect32_t<true>(S, r, t_out);
  return intersect32_t<false>(S, r, t_out);
}

// Surface(p) for the hit object (sphere.go:56-58, aabb.go:55-72).  For a box the hit point's
// entry-axis coordinate is snapped to the face; p is updated in place so the next ray
// starts exactly on the face.  Returns the material index.
GRT_DEV int surface32(const KScene& S, int id, const Ray32& r, float t, f3& p, f3& n) {
  float4 pa = S.prims[2 * id], pb = S.prims[2 * id + 1];
  int bits = __float_as_int(pb.w);
  p = fma3(r.D, t, r.O);
  if (bits & 1) {
    // entry axis = argmax of the per-axis near distances (first axis wins ties)
    float nx = fminf((pa.x - r.O.x) * r.invD.x, (pb.x - r.O.x) * r.invD.x);
    float ny = fminf((pa.y - r.O.y) * r.invD.y, (pb.y - r.O.y) * r.invD.y);
    float nz = fminf((pa.z - r.O.z) * r.invD.z, (pb.z - r.O.z) * r.invD.z);
    int ax = (nx >= ny) ? ((nx >= nz) ? 0 : 2) : ((ny >= nz) ? 1 : 2);
    if (ax == 0) p.x = r.D.x > 0.0f ? pa.x : pb.x;
    if (ax == 1) p.y = r.D.y > 0.0f ? pa.y : pb.y;
    if (ax == 2) p.z = r.D.z > 0.0f ? pa.z : pb.z;
    const float eps = 1e-4f;
    n = mk3(0.0f, 0.0f, 0.0f);
    if (p.x - pa.x < eps) n.x = -1.0f;
    else if (pb.x - p.x < eps) n.x = 1.0f;
    else if (p.y - pa.y < eps) n.y = -1.0f;
    else if (pb.y - p.y < eps) n.y = 1.0f;
    else if (p.z - pa.z < eps) n.z = -1.0f;
    else if (pb.z - p.z < eps) n.z = 1.0f;
  } else {
    f3 d = mk3(p.x - pa.x, p.y - pa.y, p.z - pa.z);
    float s = rsq32(dot3(d, d));
    n = d * (pa.w < 0.0f ? -s : s);  // Scale(Radius).Unit(): the sign of the radius survives
  }
  return bits >> 8;
}

// geom.SampleHemisphereNormal (vec3.go:202-218); u1,u2 in draw order.  v_sin/v_cos take
// revolutions, so cos(2*pi*u1) is one instruction.  The result is unit to rounding; the
// reference's final Unit() is not repeated.
GRT_DEV f3 sample_hemi32(f3 n, float u1, float u2) {
  float r2s = sqrt32(u2);
  const bool big_x = fabsf(n.x) > 0.1f;
  f3 u = big_x ? mk3(n.z, 0.0f, -n.x) : mk3(0.0f, -n.z, n.y);  // (0,1,0)xn | (1,0,0)xn
  u = unit3(u);
  f3 v = cross3(n, u);
  float c = __builtin_amdgcn_cosf(u1) * r2s, s = __builtin_amdgcn_sinf(u1) * r2s;
  float w = sqrt32(1.0f - u2);
  return mk3(fmaf(u.x, c, fmaf(v.x, s, n.x * w)), fmaf(u.y, c, fmaf(v.y, s, n.y * w)),
             fmaf(u.z, c, fmaf(v.z, s, n.z * w)));
}
// The same for a normal that is a signed unit axis (every face normal AABB.Surface returns, aabb.go:57-70): u and v of
// the basis are signed axes too, so the result's components are +-c, +-s, +-w in some order — the general function's
// own values bit for bit (its products with 0 and 1 are exact): n = +-e_x: u = (0, 0, -n.x), v = (0, 1, 0), result
// (n.x w, s, -n.x c); n = +-e_y: u = (0, 0, n.y), v = (1, 0, 0), result (s, n.y w, n.y c); n = +-e_z: u = (0, -n.z, 0),
// v = (1, 0, 0), result (s, -n.z c, n.z w).  16 VALU instead of 57; Cornell walls are boxes, so nearly every Lambert
// bounce takes it.
GRT_DEV f3 sample_hemi32_axis(f3 n, float u1, float u2) {
  const float r2s = sqrt32(u2);
  const float c = __builtin_amdgcn_cosf(u1) * r2s, s = __builtin_amdgcn_sinf(u1) * r2s;
  const float w = sqrt32(1.0f - u2);
  const bool ax = n.x != 0.0f, ay = n.y != 0.0f;
  return mk3(ax ? n.x * w : s, ax ? s : (ay ? n.y * w : -n.z * c), ax ? -n.x * c : (ay ? n.y * c : n.z * w));
}
// vec3.go:129-131.  v and n are unit vectors, so v - 2(v.n)n is unit to rounding; the reference's Unit() is not
// repeated (as in sample_hemi32).
GRT_DEV f3 reflect32(f3 v, f3 n) { return fma3(n, -2.0f * dot3(v, n), v); }
// Vec3.Refract (vec3.go:134-159); xi replaces rand.Float64() of vec3.go:152.  m2 = the material's third record:
// ratio, 1/ratio and r0 = ((1-ratio)/(1+ratio))^2 for both orientations (no reciprocal in the kernel).  The
// refracted direction r1 + r2 has |r1|^2 + |r2|^2 = 1 by construction: not normalised again.
GRT_DEV bool refract32(f3 v, f3 n, float4 m2, float xi, f3& out) {
  // With |v| = |n| = 1: r1 = (v + rn cosi) ratio has |r1|^2 = ratio^2 (1 - cosi^2), so the reference's
  // r1 + rn (-sqrt|1 - |r1|^2|) is v ratio + rn (ratio cosi - sqrt(1 - ratio^2 sini^2)) — Snell's form, one square
  // root instead of two — and total internal reflection (ratio sini > 1) is 1 - ratio^2 sini^2 < 0.
  const float dvn = dot3(v, n);
  const bool flip = dvn >= 0.0f;                   // rn = flip ? -n : n
  const float ratio = flip ? m2.y : m2.x;
  float r0 = flip ? m2.w : m2.z;
  const float cosi = fminf(fabsf(dvn), 1.0f);      // min(-v.rn, 1)
  const float sin2 = fmaf(-cosi, cosi, 1.0f);
  const float k2 = fmaf(-(ratio * ratio), sin2, 1.0f);
  const float m = 1.0f - cosi, mm = m * m;
  r0 = fmaf(1.0f - r0, mm * mm * m, r0);
  if ((k2 < 0.0f) | (r0 > xi)) return false;
  const float c = fmaf(ratio, cosi, -sqrt32(k2));
  out = fma3(n, flip ? -c : c, v * ratio);
  return true;
}

#include "grt_bvh.h"

// ---- one camera path as a per-lane state machine ------------------------------------------
// Scene.trace (scene.go:237-349) is recursive with at most max_depth-1 levels and every level
// returns (emission | 0 | weight * trace(next ray)), so a path is a throughput product and a
// terminal radiance.  A lane holds one path; path_advance() performs ONE level for it:
// intersect, Surface, material dispatch, and either the next ray of the same path or — when
// the path ended — the primary ray of the lane's next sample.  The two uniforms a level needs
// (hemisphere sample / Schlick roulette, or the pixel jitter of a fresh path) are drawn in one
// place for every lane, so the wave does not run the RNG once per material branch.
struct LaneState {
  f3 O, D;       // current ray; D is unit length (primary rays are normalised at birth: only
                 // p = O + tD and D/|D| are observable in trace(), scene.go:253-254)
  f3 thr, acc;   // path throughput, sum of finished paths' radiance (scene.go:96)
  grt_key pkey;  // RNG key of the current path (two words, grt_rng.h)
  int depth;     // scene.go:237 `depth` of the ray in flight (1 = primary)
  int s;         // sample index of the current path
  int pix;       // wave-queue mode: which of the tile's 64 pixels the current path belongs to
};

// Wave-queue mode (k_pathtrace): the (pixel, sample) pairs of a wave's tile x chunk form one
// queue; a lane whose path ends takes the NEXT pair of the queue instead of the next sample of
// "its" pixel, so all lanes run out of work within one path length of each other (per-pixel
// sample loops drain at the pace of the slowest pixel).  MEASURED SLOWER than the per-pixel loops
// at C3 (45.9 ms vs 44.1 ms: 92 VGPRs instead of 76, and the drain it removes is small next to the
// divergence inside a level), so it is an A/B variant (opts.flags & 15 == 4), not the default.  A path's
// radiance is added to its pixel's sum in LDS when it ends.  Element q of the queue is sample
// s0 + q / 64 of pixel q % 64; the assignment depends only on the wave's own paths, so a frame
// is reproducible and independent of how it is sharded.
// Photon-mapped trace() (scene.go:308-320): a path that ends on a plain diffuse surface leaves one
// gather request in slot (pixel, sample) of these arrays — position, normal and the weight
// throughput * m.Color / pi the two irradiance estimates are multiplied by.  A path has at most one
// such hit (the diffuse branch does not recurse), so the slot is its own and no queue is needed.
struct PmOut {
  float* pos;  // [slots * 3], pre-filled with NaN = no request
  float* nrm;  // [slots * 3]
  float* w;    // [slots * 3]
  int view;    // photon-map view kernels (VIEW): 1 = Scene.irradiance (scene.go:184-232), 2 = trace()'s debugging block (:261-271)
};

struct WaveQueue {
  int next, qend;          // same value in every live lane
  int s0, x0, y0;          // first sample of the chunk, tile origin
  float* acc;              // LDS: 64 x RGB sums of this wave's tile            (mode 1 only)
  const uint32_t* pixkey;  // LDS: RNG pixel keys of the tile's 64 pixels       (mode 1 only)
};
// path_shade's QUEUE parameter: 0 = every lane runs the samples of its own pixel; 1 = the wave queue above with the
// sums in LDS (float atomics; the small scenes' A/B variant); 2 = the wave queue of k_pathtrace_bvh: the level's
// radiance x throughput comes back in L.acc and the kernel adds it, in fixed point, to the sum the pixel's own lane
// holds in registers — exact integer addition, so a frame does not depend on the order in which paths end.

// Scene.intersect with the objects in LDS: the loop is fully unrolled, every lane reads the
// same address (an LDS broadcast, no bank conflict), the compiler hoists the reads ahead of
// the arithmetic, and the box/sphere choice is a scalar branch on one SGPR bit mask.  (Holding
// the 8 x 7 words in SGPRs instead overflows the 102-SGPR file: the allocator spills them to
// VGPR lanes and pays a v_readlane per operand.)
// One object of Scene.intersect's loop (scene.go:173-178) with the accept test folded in:
// AABB.Hit returns t0 when min(tMax, t1) > max(tMin, t0) (aabb.go:33-44) and the caller keeps it
// when t0 > 0; Sphere.Hit returns the first root inside (tMin, tMax) (sphere.go:36-43).
GRT_DEV void hit_update(float4 pa, float4 pb, bool is_box, int i, const Ray32& r, float bias, float& tbest, int& best) {
  if (is_box) {
    float x0 = fmaf(pa.x, r.invD.x, -r.OinvD.x), x1 = fmaf(pb.x, r.invD.x, -r.OinvD.x);
    float y0 = fmaf(pa.y, r.invD.y, -r.OinvD.y), y1 = fmaf(pb.y, r.invD.y, -r.OinvD.y);
    float z0 = fmaf(pa.z, r.invD.z, -r.OinvD.z), z1 = fmaf(pb.z, r.invD.z, -r.OinvD.z);
    float t0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float t1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const bool acc = (fminf(tbest, t1) > fmaxf(bias, t0)) & (t0 > 0.0f);
    tbest = acc ? t0 : tbest;
    best = acc ? i : best;
  } else {
    f3 oc = mk3(r.O.x - pa.x, r.O.y - pa.y, r.O.z - pa.z);
    float nk = -dot3(oc, r.D) * r.inva;
    f3 l = fma3(r.D, nk, oc);
    float sq = sqrt32((pb.x - dot3(l, l)) * r.inva);  // NaN when there is no real root
    float t1 = nk - sq, t2 = nk + sq;
    const float t = t1 > bias ? t1 : t2;              // sphere_hit32: bias >= 0
    const bool acc = (t > bias) & (t < tbest);
    tbest = acc ? t : tbest;
    best = acc ? i : best;
  }
}

// Scene.intersect with the objects in LDS: the loop is fully unrolled, every lane reads the
// same address (an LDS broadcast, no bank conflict), the record of object i+1 is requested
// before object i is tested so its LDS latency hides behind ~20 VALU instructions, and the
// box/sphere choice is a scalar branch on one SGPR bit mask.  (Holding the 8 x 7 words in SGPRs
// instead overflows the 102-SGPR file: the allocator spills them to VGPR lanes and pays a
// v_readlane per operand.)
template <int NP, bool PREFETCH = false>
GRT_DEV int intersect_lds(const float4* __restrict__ tp, uint32_t box_mask, const Ray32& r, float bias, float& t_out) {
  float tbest = 3.402823466e38f;
  int best = -1;
  // Records beyond n_prims are padding the host fills with a sphere of negative squared radius
  // (never hit), so the unrolled loop carries no per-object guard.
  if (PREFETCH) {
    float4 pa = tp[0], pb = tp[1];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      float4 na = pa, nb = pb;
      if (i + 1 < NP) {
        na = tp[2 * i + 2];
        nb = tp[2 * i + 3];
      }
      hit_update(pa, pb, (box_mask >> i) & 1u, i, r, bias, tbest, best);
      pa = na;
      pb = nb;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NP; ++i) hit_update(tp[2 * i], tp[2 * i + 1], (box_mask >> i) & 1u, i, r, bias, tbest, best);
  }
  t_out = tbest;
  return best;
}

// ---- the same loop over direction-sorted box bounds ----------------------------------------
// AABB.Hit orders the two slab distances of every axis with min/max (aabb.go:29): which bound is the near
// one depends only on the sign of the direction component.  The LDS holds the objects once per direction
// octant with the box bounds already in (near, far) order; a lane reads the table of ITS ray's octant (bit k
// of the octant = sign bit of D.k), so the per-axis min/max disappear: near = 3 FMA, far = 3 FMA, then one
// v_max3, one v_min3 and the two clamps against bias and tbest.  Same values bit for bit: for inv > 0 and
// lo <= hi, fma(lo, inv, c) <= fma(hi, inv, c) (rounding is monotone), so min(x0, x1) IS the near product.
// Rays with a zero direction component never come here (the wave takes the Go NaN/Inf path).
// Record of object i in an octant's table (8 floats): box (n.x, n.y, n.z, f.x, f.y, f.z, -, -),
// sphere (c.x, c.y, c.z, R^2, ...).  Consecutive tables are GRT_OCT_STRIDE(NP) floats apart: 16 bytes more
// than the records take, so that the 8 tables' copies of one record lie in 8 different groups of 4 LDS banks
// and a wave whose lanes look into all 8 octants still reads without a bank conflict.
#define GRT_OCT_STRIDE(NP) (8 * (NP) + 4)
GRT_DEV int ray_octant(f3 D) {
  return (int)((__float_as_uint(D.x) >> 31) | ((__float_as_uint(D.y) >> 31) << 1) | ((__float_as_uint(D.z) >> 31) << 2));
}
// One record of an octant table from the staged object record (pa, pb).
GRT_DEV void octant_record(float4 pa, float4 pb, int oct, float* q) {
  if (__float_as_int(pb.w) & 1) {
    const float lx = fminf(pa.x, pb.x), hx = fmaxf(pa.x, pb.x), ly = fminf(pa.y, pb.y), hy = fmaxf(pa.y, pb.y);
    const float lz = fminf(pa.z, pb.z), hz = fmaxf(pa.z, pb.z);
    q[0] = (oct & 1) ? hx : lx, q[1] = (oct & 2) ? hy : ly, q[2] = (oct & 4) ? hz : lz;
    q[3] = (oct & 1) ? lx : hx, q[4] = (oct & 2) ? ly : hy, q[5] = (oct & 4) ? lz : hz;
    q[6] = 0.0f, q[7] = 0.0f;
  } else {
    q[0] = pa.x, q[1] = pa.y, q[2] = pa.z, q[3] = pb.x;
    q[4] = 0.0f, q[5] = 0.0f, q[6] = 0.0f, q[7] = 0.0f;
  }
}
// v_min / v_max / v_min3 as the instructions themselves.  fminf / fmaxf go through llvm.minnum / maxnum, which in
// IEEE mode must quiet a signalling NaN first: the compiler puts a v_max_f32 x, x in front of every operand it
// cannot prove canonical — tbest (a select) and bias (a kernel argument), two extra instructions per box.  No
// operand here can be a signalling NaN (FMA results, a select of computed distances, finite scene data).
GRT_DEV float vmin_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
GRT_DEV float vmax_f32_s(float uniform, float b) {  // first operand wave-uniform (an SGPR)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b));
  return r;
}
GRT_DEV float vmin3_f32(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// ---- the object index from the loop's accept masks --------------------------------------------
// Scene.intersect keeps the LAST object accepted (scene.go:173-178: every accepted object is strictly nearer
// than the one before).  The unrolled loops below leave one accept mask per object in an SGPR pair; the
// winner of a lane is the highest i whose mask holds the lane's bit.  Round 2 turned the masks into the index
// with one v_cndmask per object (9 VALU for 8 objects); here the scalar unit forms, per BIT b of best + 1,
// the mask of the lanes whose winner has that bit set (W_i = m_i & ~(m_{i+1} | ...), bit_b = OR of the W_i
// with bit b of i + 1 set) and the lanes shift the bits in with v_addc_co_u32 (v = v + v + carry-in, the
// carry-in read from an SGPR pair): 4 VALU for 8 objects.  The scalar work runs beside other waves' VALU.
GRT_DEV uint32_t lane_bit_of(unsigned long long m) {  // 1 in the lanes whose bit of m is set, else 0
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
  return r;
}
GRT_DEV uint32_t shift_in_lane_bit(uint32_t v, unsigned long long m) {  // 2 v + (the lane's bit of m)
  uint32_t r;
  asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(v), "s"(m) : "vcc");
  return r;
}
GRT_DEV float select_by_mask(float if_clear, float if_set, unsigned long long m) {  // v_cndmask on an SGPR-pair mask
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}
template <int NP>
GRT_DEV int index_from_masks(const unsigned long long (&m)[NP]) {  // best, or -1 when no mask holds the lane
  constexpr int NB = NP < 2 ? 1 : (NP < 4 ? 2 : (NP < 8 ? 3 : (NP < 16 ? 4 : (NP < 32 ? 5 : 6))));  // bits of NP
  unsigned long long bit[NB], later = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) bit[b] = 0;
#pragma unroll
  for (int i = NP - 1; i >= 0; --i) {
    const unsigned long long w = m[i] & ~later;
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (((i + 1) >> b) & 1) bit[b] |= w;
    later |= m[i];
  }
  uint32_t v = lane_bit_of(bit[NB - 1]);
#pragma unroll
  for (int b = NB - 2; b >= 0; --b) v = shift_in_lane_bit(v, bit[b]);
  return (int)v - 1;
}

// One object of the loop; returns the accept mask (wave-uniform register pair).
GRT_DEV unsigned long long hit_update_oct(const float* __restrict__ q, bool is_box, const Ray32& r, float bias, float& tbest) {
  // (the select and the ballot sit inside each arm: merged behind the scalar branch, the compiler turns the lane
  // predicate into a 0/1 register and compares it again — two more VALU per object)
  const float4 qa = *(const float4*)q;
  if (is_box) {
    const float2 qb = *(const float2*)(q + 4);
    const float nx = fmaf(qa.x, r.invD.x, -r.OinvD.x), ny = fmaf(qa.y, r.invD.y, -r.OinvD.y), nz = fmaf(qa.z, r.invD.z, -r.OinvD.z);
    const float fx = fmaf(qa.w, r.invD.x, -r.OinvD.x), fy = fmaf(qb.x, r.invD.y, -r.OinvD.y), fz = fmaf(qb.y, r.invD.z, -r.OinvD.z);
    const float t = fmaxf(fmaxf(nx, ny), nz);                     // v_max3
    const float lo = vmax_f32_s(bias, t);
    const float hi = vmin3_f32(fx, fy, vmin_f32(fz, tbest));
    // aabb.go:33-44, scene.go:174.  The two compare masks are ANDed by the scalar unit and the select reads the SGPR pair
    // (a ballot of the lane predicate `a & b` costs a v_cndmask 0/1 and a v_cmp_ne more).
    const unsigned long long acc = __ballot(hi > lo) & __ballot(t > 0.0f);
    tbest = select_by_mask(tbest, t, acc);
    return acc;
  } else {
    f3 oc = mk3(r.O.x - qa.x, r.O.y - qa.y, r.O.z - qa.z);
    float nk = -dot3(oc, r.D) * r.inva;
    f3 l = fma3(r.D, nk, oc);
    float sq = sqrt32((qa.w - dot3(l, l)) * r.inva);  // NaN when there is no real root
    float t1 = nk - sq, t2 = nk + sq;
    const float t = t1 > bias ? t1 : t2;              // sphere_hit32: bias >= 0
    const unsigned long long acc = __ballot(t > bias) & __ballot(t < tbest);
    tbest = select_by_mask(tbest, t, acc);
    return acc;
  }
}
// oct: the lane's octant table (oct_tables + ray_octant(D) * GRT_OCT_STRIDE(NP))
template <int NP>
GRT_DEV int intersect_lds_oct(const float* __restrict__ oct, uint32_t box_mask, const Ray32& r, float bias, float& t_out) {
  float tbest = 3.402823466e38f;
  unsigned long long m[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) m[i] = hit_update_oct(oct + 8 * i, (box_mask >> i) & 1u, r, bias, tbest);
  t_out = tbest;
  return index_from_masks<NP>(m);
}

// ---- primary rays: camera-relative tables, one octant per tile ------------------------------------
// Every primary ray starts at the eye, so (bound - O) of a box and (O - C) of a sphere are constants of the
// scene: the tables below hold them (differences taken in float64 of the FP32 records, rounded once), a slab
// distance is ONE multiply, (bound - O) * (1 / D), and a sphere test starts from its oc.  Camera.Ray is affine
// in (u, v) (camera.go:34-39), so the directions of a tile's rays — every jitter included — are convex
// combinations of the directions through the tile's four corners: when those agree in the sign of every
// component (and stay clear of zero), the whole wave reads ONE octant's table at a wave-uniform address and no
// ray can have a zero component (no per-ray test for the Go NaN path).  EYE_OUT: the eye lies outside every
// box (closed) — then `t0 > 0` (scene.go:174) follows from `hi > lo`: some axis k has the eye outside the slab;
// moving towards it, near_k > 0, so t0 > 0; moving away, far_k < 0 <= bias and the window test fails.
GRT_DEV void octant_record_rel(float4 pa, float4 pb, int oct, f3 O, float* q) {
  const double ox = (double)O.x, oy = (double)O.y, oz = (double)O.z;
  if (__float_as_int(pb.w) & 1) {
    const float lx = fminf(pa.x, pb.x), hx = fmaxf(pa.x, pb.x), ly = fminf(pa.y, pb.y), hy = fmaxf(pa.y, pb.y);
    const float lz = fminf(pa.z, pb.z), hz = fmaxf(pa.z, pb.z);
    q[0] = (float)((double)((oct & 1) ? hx : lx) - ox), q[1] = (float)((double)((oct & 2) ? hy : ly) - oy);
    q[2] = (float)((double)((oct & 4) ? hz : lz) - oz);
    q[3] = (float)((double)((oct & 1) ? lx : hx) - ox), q[4] = (float)((double)((oct & 2) ? ly : hy) - oy);
    q[5] = (float)((double)((oct & 4) ? lz : hz) - oz);
    q[6] = 0.0f, q[7] = 0.0f;
  } else {
    q[0] = (float)(ox - (double)pa.x), q[1] = (float)(oy - (double)pa.y), q[2] = (float)(oz - (double)pa.z), q[3] = pb.x;
    q[4] = 0.0f, q[5] = 0.0f, q[6] = 0.0f, q[7] = 0.0f;
  }
}
// q: ONE octant's camera-relative table, the same address in every lane.  D unnormalised (camera.go:37), t in its units.
template <int NP, bool EYE_OUT>
GRT_DEV int intersect_primary_rel(const float* __restrict__ q, uint32_t box_mask, f3 D, f3 invD, float inva, float bias, float& t_out) {
  float tbest = 3.402823466e38f;
  unsigned long long m[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const float4 qa = *(const float4*)(q + 8 * i);
    if ((box_mask >> i) & 1u) {
      const float2 qb = *(const float2*)(q + 8 * i + 4);
      const float nx = qa.x * invD.x, ny = qa.y * invD.y, nz = qa.z * invD.z;
      const float fx = qa.w * invD.x, fy = qb.x * invD.y, fz = qb.y * invD.z;
      const float t = fmaxf(fmaxf(nx, ny), nz);
      const float lo = vmax_f32_s(bias, t);
      const float hi = vmin3_f32(fx, fy, vmin_f32(fz, tbest));
      m[i] = EYE_OUT ? __ballot(hi > lo) : (__ballot(hi > lo) & __ballot(t > 0.0f));
      tbest = select_by_mask(tbest, t, m[i]);
    } else {
      const f3 oc = mk3(qa.x, qa.y, qa.z);
      const float nk = -dot3(oc, D) * inva;
      const f3 l = fma3(D, nk, oc);
      const float sq = sqrt32((qa.w - dot3(l, l)) * inva);
      const float t1 = nk - sq, t2 = nk + sq;
      const float t = t1 > bias ? t1 : t2;
      m[i] = __ballot(t > bias) & __ballot(t < tbest);
      tbest = select_by_mask(tbest, t, m[i]);
    }
  }
  t_out = tbest;
  return index_from_masks<NP>(m);
}

// Surface(p) from an already fetched object record (see surface32).
// axis (optional): set when n is a signed unit axis (a box face the epsilon cascade named), see sample_hemi32_axis
GRT_DEV int surface_rec(float4 pa, float4 pb, const Ray32& r, float t, f3& p, f3& n, bool* axis = nullptr) {
  int bits = __float_as_int(pb.w);
  p = fma3(r.D, t, r.O);
  if (axis) *axis = false;
  if (bits & 1) {
    float nx = fminf((pa.x - r.O.x) * r.invD.x, (pb.x - r.O.x) * r.invD.x);
    float ny = fminf((pa.y - r.O.y) * r.invD.y, (pb.y - r.O.y) * r.invD.y);
    float nz = fminf((pa.z - r.O.z) * r.invD.z, (pb.z - r.O.z) * r.invD.z);
    const bool ax0 = (nx >= ny) & (nx >= nz), ax1 = !ax0 & (ny >= nz), ax2 = !ax0 & !ax1;
    p.x = ax0 ? (r.D.x > 0.0f ? pa.x : pb.x) : p.x;
    p.y = ax1 ? (r.D.y > 0.0f ? pa.y : pb.y) : p.y;
    p.z = ax2 ? (r.D.z > 0.0f ? pa.z : pb.z) : p.z;
    const float eps = 1e-4f;  // aabb.go:56; the cascade of aabb.go:57-70 without branches
    const bool x0 = p.x - pa.x < eps, x1 = pb.x - p.x < eps, y0 = p.y - pa.y < eps, y1 = pb.y - p.y < eps;
    const bool z0 = p.z - pa.z < eps, z1 = pb.z - p.z < eps;
    const bool cx = x0 | x1, cy = !cx & (y0 | y1), cz = !cx & !(y0 | y1) & (z0 | z1);
    n.x = cx ? (x0 ? -1.0f : 1.0f) : 0.0f;
    n.y = cy ? (y0 ? -1.0f : 1.0f) : 0.0f;
    n.z = cz ? (z0 ? -1.0f : 1.0f) : 0.0f;
    if (axis) *axis = cx | cy | cz;
  } else {
    f3 d = mk3(p.x - pa.x, p.y - pa.y, p.z - pa.z);
    float s = rsq32(dot3(d, d));
    n = d * (pa.w < 0.0f ? -s : s);
  }
  return bits >> 8;
}

// Surface(p) of a box from its record in the lane's octant table (intersect_lds_oct): the three near slab
// distances are the very values the loop took its max3 of, so the entry axis is the one that produced t (first
// axis on ties), and the near bound of that axis is the face the hit point is snapped to.  The epsilon cascade
// of aabb.go:57-70 runs on MinBound / MaxBound (pa, pb) as in the reference.
GRT_DEV int surface_rec_oct(float4 pa, float4 pb, const float* __restrict__ q, const Ray32& r, float t, f3& p, f3& n, bool* axis = nullptr) {
  const int bits = __float_as_int(pb.w);
  p = fma3(r.D, t, r.O);
  if (axis) *axis = false;
  if (bits & 1) {
    const float4 qa = *(const float4*)q;
    const float nx = fmaf(qa.x, r.invD.x, -r.OinvD.x), ny = fmaf(qa.y, r.invD.y, -r.OinvD.y), nz = fmaf(qa.z, r.invD.z, -r.OinvD.z);
    const bool ax0 = (nx >= ny) & (nx >= nz), ax1 = !ax0 & (ny >= nz), ax2 = !ax0 & !ax1;
    p.x = ax0 ? qa.x : p.x;
    p.y = ax1 ? qa.y : p.y;
    p.z = ax2 ? qa.z : p.z;
    const float eps = 1e-4f;  // aabb.go:56
    const bool x0 = p.x - pa.x < eps, x1 = pb.x - p.x < eps, y0 = p.y - pa.y < eps, y1 = pb.y - p.y < eps;
    const bool z0 = p.z - pa.z < eps, z1 = pb.z - p.z < eps;
    const bool cx = x0 | x1, cy = !cx & (y0 | y1), cz = !cx & !(y0 | y1) & (z0 | z1);
    n.x = cx ? (x0 ? -1.0f : 1.0f) : 0.0f;
    n.y = cy ? (y0 ? -1.0f : 1.0f) : 0.0f;
    n.z = cz ? (z0 ? -1.0f : 1.0f) : 0.0f;
    if (axis) *axis = cx | cy | cz;
  } else {
    f3 d = mk3(p.x - pa.x, p.y - pa.y, p.z - pa.z);
    float s = rsq32(dot3(d, d));
    n = d * (pa.w < 0.0f ? -s : s);
  }
  return bits >> 8;
}

// Direct illumination of the plain "diffuse" material (scene.go:322-346): every light is a
// point at Pos(); the shadow loop visits every object with a shrinking tNear and marks the
// light hidden when any non-emissive object is accepted (no distance-to-light cap).
GRT_DEV f3 direct_light32(const KScene& S, f3 p, f3 n, f3 color) {
  f3 result = mk3(0.0f, 0.0f, 0.0f);
  for (int li = 0; li < S.n_lights; ++li) {
    float4 lp = S.lights[2 * li], lc = S.lights[2 * li + 1];
    f3 dir = unit3(mk3(lp.x - p.x, lp.y - p.y, lp.z - p.z));
    float fd = fmaxf(dot3(n, dir), 0.0f);
    Ray32 sr;
    ray_setup(sr, p, dir);
    bool careful = __any(ray_has_zero(sr));
    float visible = 1.0f, tn = 3.402823466e38f;
    for (int i = 0; i < S.n_prims; ++i) {
      float4 pa = S.prims[2 * i], pb = S.prims[2 * i + 1];
      int bits = __float_as_int(pb.w);
      float ht;
      if (bits & 1)
        ht = careful ? aabb_hit32<true>(pa, pb, sr, S.bias, tn) : aabb_hit32<false>(pa, pb, sr, S.bias, tn);
      else
        ht = sphere_hit32(pa, pb, sr, S.bias, tn);
      if (ht > 0.0f) {
        if (!(bits & 2)) visible = 0.0f;
        tn = ht;
      }
    }
    float k = fd * visible;
    result = result + mk3(color.x * k * lc.x, color.y * k * lc.y, color.z * k * lc.z);
  }
  return result;
}

// The same for hierarchy scenes.  The shadow loop is order-dependent as written (tNear shrinks
// with every accepted object, in index order), but its RESULT has an order-free form.  Call an
// object a record when it is accepted at its turn.  The light is hidden iff some non-emissive
// object is a record.  Up to the first such record every record is emissive, so the cap tNear
// that a non-emissive object o meets is M(o) = the value left by the emissive records below o —
// a chain computed from the (few) emissive objects alone, in index order.  Hence
//     hidden  <=>  some non-emissive o has Hit_o(bias, M(o)) > 0
// (if a non-emissive record exists, the first one meets exactly M(o); if none exists, every cap
// is exactly M(o) and nothing is accepted).  M is constant between two emissive indices, so the
// test is one any-hit query of the hierarchy per non-empty index range, with the same Hit calls
// as the linear loop.  With the light first in Scene.Objects (the reference's scenes) that is one
// query, capped at the distance to the light.
GRT_DEV f3 direct_light_bvh(const KScene& S, f3 p, f3 n, f3 color, bool active, int* stack, int stride) {
  f3 result = mk3(0.0f, 0.0f, 0.0f);
  for (int li = 0; li < S.n_lights; ++li) {
    float4 lp = S.lights[2 * li], lc = S.lights[2 * li + 1];
    f3 dir = unit3(mk3(lp.x - p.x, lp.y - p.y, lp.z - p.z));
    float fd = fmaxf(dot3(n, dir), 0.0f);
    Ray32 sr;
    ray_setup(sr, p, dir);
    const bool careful = __any(active & ray_has_zero(sr));
    float tn = 3.402823466e38f;
    int prev = -1;
    bool hidden = false;
    for (int e = 0; e <= S.n_lights; ++e) {
      const int ei = e < S.n_lights ? __float_as_int(S.lights[2 * e + 1].w) : S.n_prims;  // wave-uniform
      if (ei > prev + 1) {  // non-emissive objects with index in (prev, ei) meet the cap tn
        const bool go = active & !hidden;
        const bool f = careful ? bvh_any_hit<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, sr, S.bias, tn, prev, ei, 0, 2, go, stack, stride)
                               : bvh_any_hit<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, sr, S.bias, tn, prev, ei, 0, 2, go, stack, stride);
        hidden |= f;
      }
      if (e < S.n_lights) {  // the emissive object itself: accepted, it shrinks the cap and hides nothing
        const float4 pa = S.prims[2 * ei], pb = S.prims[2 * ei + 1];
        float ht;
        if (__float_as_int(pb.w) & 1)
          ht = careful ? aabb_hit32<true>(pa, pb, sr, S.bias, tn) : aabb_hit32<false>(pa, pb, sr, S.bias, tn);
        else
          ht = sphere_hit32(pa, pb, sr, S.bias, tn);
        if (ht > 0.0f) tn = ht;
      }
      prev = ei;
    }
    float k = fd * (hidden ? 0.0f : 1.0f);
    result = result + mk3(color.x * k * lc.x, color.y * k * lc.y, color.z * k * lc.z);
  }
  return result;
}

// Birth of sample `s` of the lane's pixel: scene.go:93-95 with the jitter (u1,u2) supplied.
GRT_DEV void primary_unit(const KScene& S, float fx, float fy, float u1, float u2, f3& O, f3& D) {
  float u = (fx + u1) * (1.0f / (float)S.W);
  float v = (fy + u2) * (1.0f / (float)S.H);
  O = mk3(S.cam.ox, S.cam.oy, S.cam.oz);
  D = unit3(mk3(fmaf(S.cam.vx, v, fmaf(S.cam.hx, u, S.cam.lx)), fmaf(S.cam.vy, v, fmaf(S.cam.hy, u, S.cam.ly)),
                fmaf(S.cam.vz, v, fmaf(S.cam.hz, u, S.cam.lz))));
}
GRT_DEV void lane_begin(const KScene& S, float fx, float fy, uint32_t pix_key, int s, LaneState& L) {
  L.s = s;
  L.pkey = grt_rng_path_key(pix_key, (uint32_t)s);
  float ju, jv;
  grt_rng_draw2(L.pkey, GRT_PAIR_JITTER, &ju, &jv);
  primary_unit(S, fx, fy, ju, jv, L.O, L.D);
  L.thr = mk3(1.0f, 1.0f, 1.0f);
  L.acc = mk3(0.0f, 0.0f, 0.0f);
  L.depth = 1;
}

// DIRECT: how the plain-diffuse branch finds its shadows: 0 linear loop (direct_light32), 1 the scene has
// no plain-diffuse object (hierarchy scenes without one: the branch is compiled out), 2 hierarchy.
template <int QUEUE, int DIRECT = 0, bool VIEW = false>
GRT_DEV void path_shade(const KScene& S, const float4* __restrict__ tp, const float4* __restrict__ tm, float fx, float fy,
                        uint32_t pix_key, int s_end, LaneState& L, bool& alive, const Ray32& r, int id, float t,
                        WaveQueue* Q = nullptr, const PmOut* pm = nullptr, int pm_base = 0, int* bvh_stack = nullptr,
                        int bvh_stride = 0, const float* octq = nullptr);

// One level of trace() for one lane.  NP > 0: tp (objects) and tm (materials) are LDS copies
// and the intersect loop is unrolled over NP records; NP == 0: tp/tm are the HBM-resident
// arrays of S and the loop runs over S.n_prims.  `alive` turns false when the lane has finished
// sample s_end-1.  n_rays counts intersect() calls (scene.go:242).
template <int NP, bool RELOAD = false, bool PREFETCH = false, bool QUEUE = false, bool VIEW = false>
GRT_DEV void path_advance(const KScene& S, const float4* __restrict__ tp, const float4* __restrict__ tm, float fx, float fy,
                          uint32_t pix_key, int s_end, LaneState& L, bool& alive, int& n_rays, int* bvh_stack = nullptr,
                          WaveQueue* Q = nullptr, const PmOut* pm = nullptr, int pm_base = 0, const float4* ts = nullptr,
                          int n_boxes = 0, const float* oct = nullptr) {
  // RELOAD: forbid hoisting the LDS object reads out of the wave loop (keeps them out of
  // long-lived VGPRs: more waves per SIMD, 16 ds_read_b128 per level)
  if (RELOAD && NP > 0) asm volatile("" ::: "memory");
  Ray32 r;
  ray_setup_unit(r, L.O, L.D);
  float t;
  int id;
  const float* octq = nullptr;  // the lane's direction-octant table (scenes of <= 8 objects)
  if (NP < 0)  // NP < 0: bounding-volume hierarchy (grt_bvh.h), per-lane stack in LDS
    id = __any(ray_has_zero(r)) ? intersect_bvh_t<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, bvh_stack, 256)
                                : intersect_bvh_t<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, bvh_stack, 256);
  else if (NP == 0 && RELOAD)  // mid-size scenes: LDS copies of the objects, sorted by kind for the usual rays
    id = __any(ray_has_zero(r)) ? intersect_tab_t<true>(tp, S.n_prims, S.bias, r, t) : intersect_sorted(ts, n_boxes, S.n_prims, S.bias, r, t);
  else if (__builtin_expect(__any(ray_has_zero(r)), 0))
    id = intersect32_t<true>(S, r, t);  // Go NaN/Inf rules, objects from memory (rare)
  else if (NP > 0 && oct)
    id = intersect_lds_oct<(NP > 0 ? NP : 1)>(octq = oct + ray_octant(r.D) * GRT_OCT_STRIDE(NP > 0 ? NP : 1), S.box_mask, r, S.bias, t);
  else if (NP > 0)
    id = intersect_lds<NP, PREFETCH>(tp, S.box_mask, r, S.bias, t);
  else
    id = intersect32_t<false>(S, r, t);
  n_rays++;                                         // scene.go:242
  path_shade<QUEUE, NP < 0 ? 2 : 0, VIEW>(S, tp, tm, fx, fy, pix_key, s_end, L, alive, r, id, t, Q, pm, pm_base, bvh_stack, 256, octq);
}

// Everything of one trace() level after intersect(): Surface, material dispatch, the radiance a
// terminating level carries, and the lane's next ray (same path, or the next sample's primary ray).
// VIEW: the photon-map view instead of trace() — Scene.irradiance (scene.go:184-232: metal and glass are followed,
// every other surface, lights and Lambert walls included, shows IrradianceEst(p, n, 0, 100)) or, pm->view == 2,
// trace()'s debugging block (scene.go:261-271: the estimate at the first hit).  The level leaves a gather request
// with the path's throughput as its weight; nothing else carries radiance.
template <int QUEUE, int DIRECT, bool VIEW>
GRT_DEV void path_shade(const KScene& S, const float4* __restrict__ tp, const float4* __restrict__ tm, float fx, float fy,
                        uint32_t pix_key, int s_end, LaneState& L, bool& alive, const Ray32& r, int id, float t, WaveQueue* Q,
                        const PmOut* pm, int pm_base, int* bvh_stack, int bvh_stride, const float* octq) {
#ifdef GRT_TRACE_PATH  // experiment: the hits of one path, printed by whichever kernel traces it
  {
    const grt_key want = grt_rng_path_key(grt_rng_pixel_key(S.seed32, GRT_TRACE_PIXEL), GRT_TRACE_SAMPLE);
    if (L.pkey.a == want.a && L.pkey.b == want.b)
      printf("Q%d depth %d id %d t %.9g O %.9g %.9g %.9g D %.9g %.9g %.9g\n", QUEUE, L.depth, id, (double)t, (double)r.O.x, (double)r.O.y,
             (double)r.O.z, (double)r.D.x, (double)r.D.y, (double)r.D.z);
  }
#endif
  const bool hit = id >= 0;                         // scene.go:244
  const int idc = hit ? id : 0;
  const float4 pa = tp[2 * idc], pb = tp[2 * idc + 1];
  f3 p, n;
  bool axis_n;  // the normal is a signed unit axis (a box face)
  const int m = octq ? surface_rec_oct(pa, pb, octq + 8 * idc, r, t, p, n, &axis_n) : surface_rec(pa, pb, r, t, p, n, &axis_n);  // scene.go:254-255
  const float4 m0 = tm[GRT_MAT_STRIDE * m], m1 = tm[GRT_MAT_STRIDE * m + 1];
  const int flags = __float_as_int(m1.w);
  const f3 color = mk3(m0.x, m0.y, m0.z);
  // material priority of scene.go:274-305 — Normal, Emittance, Lambert, Reflectivity, Transparent, diffuse — decided once
  // per material when the scene is staged (GRT_MAT_CLASS, from the float64 fields); the photon-map view has its own
  // order (scene.go:207-228: metal, glass, everything else).  One small integer per hit instead of a chain of flags.
  const int cls = !hit ? GRT_CLS_MISS
                       : (VIEW ? (pm->view == 1 ? GRT_MAT_VIEW_CLASS(flags) : GRT_CLS_DIFFUSE) : GRT_MAT_CLASS(flags));
  const bool is_normal = cls == GRT_CLS_NORMAL, is_emit = cls == GRT_CLS_EMIT, is_lamb = cls == GRT_CLS_LAMBERT;
  const bool is_metal = cls == GRT_CLS_METAL;                                                    // scene.go:207 / :287
  const bool is_glass = cls == GRT_CLS_GLASS;                                                    // scene.go:215 / :295
  const bool is_diff = cls == GRT_CLS_DIFFUSE;
  f3 rad = mk3(0.0f, 0.0f, 0.0f);
  if (is_emit) rad = color * m1.z;                                                              // scene.go:279
  if (is_normal) rad = mk3((n.x + 0.5f) * 0.5f, (n.y + 0.5f) * 0.5f, (n.z + 0.5f) * 0.5f);      // scene.go:275
  if (VIEW) {
    if (is_diff) {                                   // scene.go:228 (k = 100) / :262,268 (k = 50): the gather runs after this launch
      const size_t slot = 3 * (size_t)(pm_base + L.s);
      pm->pos[slot] = p.x, pm->pos[slot + 1] = p.y, pm->pos[slot + 2] = p.z;
      pm->nrm[slot] = n.x, pm->nrm[slot + 1] = n.y, pm->nrm[slot + 2] = n.z;
      pm->w[slot] = L.thr.x, pm->w[slot + 1] = L.thr.y, pm->w[slot + 2] = L.thr.z;
    }
  } else if (DIRECT != 1 && __builtin_expect(__any(is_diff), 0)) {
    f3 direct = mk3(0.0f, 0.0f, 0.0f);
    if (DIRECT == 2) direct = direct_light_bvh(S, p, n, color, is_diff, bvh_stack, bvh_stride);  // wave-cooperative
    if (is_diff) {
      rad = DIRECT == 2 ? direct : direct_light32(S, p, n, color);                               // scene.go:322-346
      if (pm && pm->pos) {                                                                       // scene.go:308-320
        const size_t slot = 3 * (size_t)(pm_base + L.s);
        const f3 w = (L.thr * color) * 0.318309886183790672f;  // BRDF = 1/pi (:309), Times(m.Color) (:319)
        pm->pos[slot] = p.x, pm->pos[slot + 1] = p.y, pm->pos[slot + 2] = p.z;
        pm->nrm[slot] = n.x, pm->nrm[slot + 1] = n.y, pm->nrm[slot + 2] = n.z;
        pm->w[slot] = w.x, pm->w[slot + 1] = w.y, pm->w[slot + 2] = w.z;
      }
    }
  }
  // a scattering material at the last level: the recursive call returns black (scene.go:238)
  bool term = (cls < GRT_CLS_LAMBERT) | (cls > GRT_CLS_GLASS) | (L.depth + 1 >= S.max_depth);  // miss, Normal, emitter, diffuse
  f3 dir = n;
  if ((is_metal | is_glass) & !term) dir = reflect32(r.D, n);  // metal; glass that reflects (total internal reflection, Schlick)
  if (is_metal & !term) {                           // scene.go:287-294
    if (m1.y != 0.0f) {
      float z1, z2;
      grt_rng_draw2(L.pkey, GRT_PAIR_HIT(L.depth), &z1, &z2);
      f3 fz = sample_hemi32(n, z1, z2);
      dir = unit3(fma3(fz, m1.y, dir));             // keep |D| = 1 (direction unchanged)
    }
    // scene.go:291, tested whatever the roughness: no ray and black when the reflected direction does not
    // leave the surface — a mirror met from inside (the eye or a secondary origin within a mirror sphere),
    // or a box whose X->Y->Z epsilon cascade (aabb.go:57-70) names the neighbouring face at an edge
    if (!(dot3(dir, n) > 0.0f)) term = true;
  }
  if (QUEUE) {
    // only a terminating level carries radiance (emission, NormalMaterial, direct light)
    const f3 c = L.thr * rad;
    if (QUEUE == 1) {
      if (term & ((c.x != 0.0f) | (c.y != 0.0f) | (c.z != 0.0f))) {
        atomicAdd(Q->acc + 3 * L.pix, c.x);
        atomicAdd(Q->acc + 3 * L.pix + 1, c.y);
        atomicAdd(Q->acc + 3 * L.pix + 2, c.z);
      }
    } else {
      L.acc = c;                                    // QUEUE == 2: the caller adds it to the sum of the path's pixel
    }
    const unsigned long long ended = __ballot(term);  // every live lane sees the same mask
    const int q = Q->next + __builtin_amdgcn_mbcnt_hi((uint32_t)(ended >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ended, 0));
    if (QUEUE == 1) Q->next += __builtin_popcountll(ended);  // QUEUE == 2: the caller advances the counter (a path that has just been born has depth 1)
    if (term) {                                     // next (pixel, sample) of the wave's queue
      alive = q < Q->qend;
      L.pix = q & 63;
      L.s = Q->s0 + (q >> 6);
      const int px = Q->x0 + (L.pix & 7), py = Q->y0 + (L.pix >> 3);
      pix_key = QUEUE == 1 ? Q->pixkey[L.pix] : grt_rng_pixel_key(S.seed32, (uint32_t)py * (uint32_t)S.W + (uint32_t)px);
      fx = (float)px, fy = (float)py;
      L.pkey = grt_rng_path_key(pix_key, (uint32_t)L.s);
      L.depth = 0;
      L.thr = mk3(1.0f, 1.0f, 1.0f);
    }
  } else {
    L.acc = mk3(fmaf(L.thr.x, rad.x, L.acc.x), fmaf(L.thr.y, rad.y, L.acc.y), fmaf(L.thr.z, rad.z, L.acc.z));
    if (term) {                                     // next sample of this pixel (scene.go:92)
      L.s++;
      alive = L.s < s_end;
      L.pkey = grt_rng_next_path_key(pix_key, (uint32_t)L.s, L.pkey);
      L.depth = 0;
      L.thr = mk3(1.0f, 1.0f, 1.0f);
    }
  }
  // one hash for the level's two uniforms: pair L.depth — a path that has just ended has depth 0 = GRT_PAIR_JITTER,
  // a bounce at depth d draws GRT_PAIR_HIT(d) = d
  float u1, u2;
  grt_rng_draw2(L.pkey, (uint32_t)L.depth, &u1, &u2);
  if (term) {
    primary_unit(S, fx, fy, u1, u2, L.O, L.D);
  } else {
    L.O = p;
    if (is_lamb) {                                  // scene.go:280-286
      // box faces (every wall of the Cornell room): the basis is made of signed axes, same values in 16 VALU instead of 57
      if (__any(!axis_n)) L.D = sample_hemi32(n, u1, u2);
      else L.D = sample_hemi32_axis(n, u1, u2);
      L.thr = L.thr * color;
    } else if (is_metal) {
      L.D = dir;
      L.thr = L.thr * (VIEW ? color : color * m1.x);  // f.Times(...) (scene.go:213) | .Times(m.Color).Scale(m.Reflectivity) (:293)
    } else {                                        // scene.go:295-304; u1 is the roulette draw
      f3 d;
      if (!refract32(r.D, n, tm[GRT_MAT_STRIDE * m + 2], u1, d)) d = dir;
      L.D = d;
    }
  }
  L.depth++;
}

// ======================================================================================
// FP64 re-test path: the reference's arithmetic, operation for operation.
// ======================================================================================
struct d3 {
  double x, y, z;
};
GRT_DEV d3 mkd(double x, double y, double z) {
  d3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
struct KScene64 {
  double cam[12];  // origin, lower_left, horizontal, vertical (camera.go:10-13)
  double bias;
  const double* __restrict__ prims;  // 8 doubles per prim: a[3], b[3], radius, kind
};
GRT_DEV double gomax64(double x, double y) {
  if ((x == (double)INFINITY) | (y == (double)INFINITY)) return (double)INFINITY;
  if ((x != x) | (y != y)) return (double)NAN;
  if (x == 0.0 && x == y) return __builtin_signbit(x) ? y : x;
  return x > y ? x : y;
}
GRT_DEV double gomin64(double x, double y) {
  if ((x == -(double)INFINITY) | (y == -(double)INFINITY)) return -(double)INFINITY;
  if ((x != x) | (y != y)) return (double)NAN;
  if (x == 0.0 && x == y) return __builtin_signbit(x) ? x : y;
  return x < y ? x : y;
}
GRT_DEV double prim_hit64(const double* P, d3 O, d3 D, double tMin, double tMax) {
#pragma clang fp contract(off)
  if (P[7] == 0.0) {  // sphere.go:23-46
    double ocx = O.x - P[0], ocy = O.y - P[1], ocz = O.z - P[2];
    double a = D.x * D.x + D.y * D.y + D.z * D.z;
    double halfB = ocx * D.x + ocy * D.y + ocz * D.z;
    double c = (ocx * ocx + ocy * ocy + ocz * ocz) - P[6] * P[6];
    double disc = halfB * halfB - a * c;
    if (disc < 0.0) return -1.0;
    double sqrtd = __builtin_sqrt(disc);
    double t = (-halfB - sqrtd) / a;
    if (t > tMin && t < tMax) return t;
    t = (-halfB + sqrtd) / a;
    if (t > tMin && t < tMax) return t;
    return -1.0;
  }
  // aabb.go:25-45
  double nx = (P[0] - O.x) / D.x, ny = (P[1] - O.y) / D.y, nz = (P[2] - O.z) / D.z;
  double fx = (P[3] - O.x) / D.x, fy = (P[4] - O.y) / D.y, fz = (P[5] - O.z) / D.z;
  double n2x = gomin64(nx, fx), n2y = gomin64(ny, fy), n2z = gomin64(nz, fz);
  double f2x = gomax64(nx, fx), f2y = gomax64(ny, fy), f2z = gomax64(nz, fz);
  double t0 = gomax64(gomax64(n2x, n2y), n2z);
  double t1 = gomin64(gomin64(f2x, f2y), f2z);
  double lo = tMin, hi = tMax;
  if (t0 > lo) lo = t0;
  if (t1 < hi) hi = t1;
  if (hi <= lo) return -1.0;
  return t0;
}
GRT_DEV int intersect64(const KScene64& S, int n_prims, d3 O, d3 D, double& t_out) {
  double t = 1.7976931348623157e308;
  int hit = -1;
  for (int i = 0; i < n_prims; ++i) {
    double ht = prim_hit64(S.prims + 8 * i, O, D, S.bias, t);
    if (ht > 0.0) {
      hit = i;
      t = ht;
    }
  }
  t_out = t;
  return hit;
}
GRT_DEV d3 surface64(const double* P, d3 O, d3 D, double t) {
#pragma clang fp contract(off)
  d3 p = mkd(O.x + D.x * t, O.y + D.y * t, O.z + D.z * t);  // ray.go:15-17
  if (P[7] == 0.0) {                                        // sphere.go:56-58
    double x = (p.x - P[0]) * P[6], y = (p.y - P[1]) * P[6], z = (p.z - P[2]) * P[6];
    double k = 1.0 / __builtin_sqrt(x * x + y * y + z * z);
    return mkd(x * k, y * k, z * k);
  }
  const double eps = 1e-4;  // aabb.go:55-72
  if (p.x < P[0] + eps) return mkd(-1, 0, 0);
  if (p.x > P[3] - eps) return mkd(1, 0, 0);
  if (p.y < P[1] + eps) return mkd(0, -1, 0);
  if (p.y > P[4] - eps) return mkd(0, 1, 0);
  if (p.z < P[2] + eps) return mkd(0, 0, -1);
  if (p.z > P[5] - eps) return mkd(0, 0, 1);
  return mkd(0, 0, 0);
}
GRT_DEV void make_primary64(const KScene64& S, int W, int H, int x, int y, double ju, double jv, d3& O, d3& D) {
#pragma clang fp contract(off)
  double u = ((double)x + ju) / (double)W, v = ((double)y + jv) / (double)H;  // scene.go:93-94
  const double* c = S.cam;
  O = mkd(c[0], c[1], c[2]);
  D = mkd(((c[3] + c[6] * u) + c[9] * v) - c[0], ((c[4] + c[7] * u) + c[10] * v) - c[1],
          ((c[5] + c[8] * u) + c[11] * v) - c[2]);  // camera.go:37
}

// ======================================================================================
// Decision-margin analysis for PRIMARY rays: returns true when any FP32 comparison made by
// intersect32/surface32 could come out differently in exact arithmetic, or when the FP32
// t / normal may miss the parity tolerance (1e-5 relative).  First-order interval
// propagation; u = 2^-24 is the FP32 unit roundoff.  Preconditions: r.D is the float64
// direction rounded once to FP32 (k_primary_hits), r.O is exact or rounded once.
// Conservative by construction and validated against the float64 oracle in
// tests/test_gpu_primary.py.
// ======================================================================================
GRT_DEV float slab_err(float lo, float hi, float o, float inv, float k0, float k1) {
  const float u = 6.0e-8f;
  // t = b*inv - o*inv: each product carries the roundings of b, 1/D (rcp, 1 ulp), D and the
  // product itself; the fma adds one more on the result
  return 3.0f * u * (fmaxf(fabsf(lo), fabsf(hi)) + fabsf(o)) * fabsf(inv) + 2.0f * u * fmaxf(fabsf(k0), fabsf(k1));
}

// One object of the loop under first-order error bounds.  (win_lo, win_hi) bracket the exact value of
// the tMax the object meets (the running best +- its bound; BIG, BIG for tMax = +inf).  Returns
//   0  the object is certainly not accepted,
//   1  it is certainly accepted (scene.go:174) with a distance whose exact value lies in [t - e, t + e],
//   2  FP32 cannot tell.
GRT_DEV int hit_margin(float4 pa, float4 pb, const Ray32& r, float bias, float Dlen, float Olen, float win_lo, float win_hi,
                       float& t_out, float& e_out) {
  const float u = 6.0e-8f;
  if (__float_as_int(pb.w) & 1) {
    float x0 = fmaf(pa.x, r.invD.x, -r.OinvD.x), x1 = fmaf(pb.x, r.invD.x, -r.OinvD.x);
    float y0 = fmaf(pa.y, r.invD.y, -r.OinvD.y), y1 = fmaf(pb.y, r.invD.y, -r.OinvD.y);
    float z0 = fmaf(pa.z, r.invD.z, -r.OinvD.z), z1 = fmaf(pb.z, r.invD.z, -r.OinvD.z);
    float ex = slab_err(pa.x, pb.x, r.O.x, r.invD.x, x0, x1);
    float ey = slab_err(pa.y, pb.y, r.O.y, r.invD.y, y0, y1);
    float ez = slab_err(pa.z, pb.z, r.O.z, r.invD.z, z0, z1);
    float nx = fminf(x0, x1), ny = fminf(y0, y1), nz = fminf(z0, z1);
    float fx = fmaxf(x0, x1), fy = fmaxf(y0, y1), fz = fmaxf(z0, z1);
    float t0 = fmaxf(fmaxf(nx, ny), nz);
    float t0lo = fmaxf(fmaxf(nx - ex, ny - ey), nz - ez), t0hi = fmaxf(fmaxf(nx + ex, ny + ey), nz + ez);
    float t1lo = fminf(fminf(fx - ex, fy - ey), fz - ez), t1hi = fminf(fminf(fx + ex, fy + ey), fz + ez);
    float lo_lo = fmaxf(bias, t0lo), lo_hi = fmaxf(bias, t0hi);
    float hi_lo = fminf(win_lo, t1lo), hi_hi = fminf(win_hi, t1hi);
    bool def_hit = hi_lo > lo_hi, def_miss = hi_hi < lo_lo;
    if (!def_hit && !def_miss) return 2;
    if (def_hit) {
      if (t0lo > 0.0f) {
        t_out = t0;
        e_out = fmaxf(t0hi - t0, t0 - t0lo);
        return 1;
      }
      if (!(t0hi < 0.0f)) return 2;
    }
    return 0;
  }
  f3 C = mk3(pa.x, pa.y, pa.z);
  f3 oc = r.O - C;
  float ocl = sqrt32(dot3(oc, oc));
  float eoc = 1.5f * u * (sqrt32(dot3(C, C)) + Olen + ocl);  // |delta oc|
  float halfB = dot3(oc, r.D);
  float k = halfB * r.inva;
  f3 l = fma3(r.D, -k, oc);
  float ll = dot3(l, l), ln = sqrt32(ll);
  float dp = pb.x - ll;
  float e_hb = Dlen * eoc + 4.0f * u * ocl * Dlen;
  float ek = e_hb * r.inva + 8.0f * u * fabsf(k);
  float el = eoc + ek * Dlen + 2.0f * u * (fabsf(k) * Dlen + ln);
  float edp = 2.0f * ln * el + el * el + 3.0f * u * (ll + pb.x);
  if (dp + edp < 0.0f) return 0;  // definitely no real root
  if (dp - edp <= 0.0f) return 2;
  float s2 = dp * r.inva, es2 = edp * r.inva + 8.0f * u * s2;
  float sq = sqrt32(s2);
  float es = es2 * rcp32(sq + sqrt32(fmaxf(s2 - es2, 0.0f))) + 2.0f * u * sq;
  float et = ek + es + 2.0f * u * (fabsf(k) + sq);
  float t1 = -k - sq, t2 = -k + sq;
  float t = -1.0f;
  bool in1 = (t1 - et > bias) && (t1 + et < win_lo), out1 = (t1 + et <= bias) || (t1 - et >= win_hi);
  if (in1) {
    t = t1;
  } else if (!out1) {
    return 2;
  } else {
    bool in2 = (t2 - et > bias) && (t2 + et < win_lo), out2 = (t2 + et <= bias) || (t2 - et >= win_hi);
    if (in2) t = t2;
    else if (!out2) return 2;
  }
  if (t > 0.0f) {
    t_out = t;
    e_out = et;
    return 1;
  }
  return 0;
}

// The winner is certain (object `best` at tb +- eb): can the FP32 t and Surface() normal miss the parity
// tolerance (1e-5 relative), or the face choice of AABB.Surface's cascade (aabb.go:57-70) come out
// differently in exact arithmetic?
GRT_DEV bool hit_accuracy_doubtful(const KScene& S, const Ray32& r, int best, float tb, float eb, float t_fast) {
  const float u = 6.0e-8f;
  const float Dlen = sqrt32(r.a);
  if (!(fabsf(tb - t_fast) <= eb)) return true;
  if (eb > 8e-6f * fabsf(tb)) return true;  // t must hold 1e-5 relative
  float4 pa = S.prims[2 * best], pb = S.prims[2 * best + 1];
  f3 p = fma3(r.D, tb, r.O);
  if (__float_as_int(pb.w) & 1) {
    float x0 = (pa.x - r.O.x) * r.invD.x, x1 = (pb.x - r.O.x) * r.invD.x;
    float y0 = (pa.y - r.O.y) * r.invD.y, y1 = (pb.y - r.O.y) * r.invD.y;
    float z0 = (pa.z - r.O.z) * r.invD.z, z1 = (pb.z - r.O.z) * r.invD.z;
    float ex = slab_err(pa.x, pb.x, r.O.x, r.invD.x, x0, x1);
    float ey = slab_err(pa.y, pb.y, r.O.y, r.invD.y, y0, y1);
    float ez = slab_err(pa.z, pb.z, r.O.z, r.invD.z, z0, z1);
    float nx = fminf(x0, x1), ny = fminf(y0, y1), nz = fminf(z0, z1);
    int ax = (nx >= ny) ? ((nx >= nz) ? 0 : 2) : ((ny >= nz) ? 1 : 2);
    float na = ax == 0 ? nx : (ax == 1 ? ny : nz), ea = ax == 0 ? ex : (ax == 1 ? ey : ez);
    if (ax != 0 && nx + ex >= na - ea) return true;
    if (ax != 1 && ny + ey >= na - ea) return true;
    if (ax != 2 && nz + ez >= na - ea) return true;
    const float eps = 1e-4f;
    // axes the cascade (aabb.go:57-70) tests before the entry axis must miss clearly:
    // error of p.k = O.k + t D.k, plus the rounding of the bound
    float mx = 2.0f * (fabsf(tb * r.D.x) * 2.0f * u + eb * fabsf(r.D.x) + u * (fabsf(p.x) + fabsf(pa.x) + fabsf(pb.x))) + 1e-7f;
    float my = 2.0f * (fabsf(tb * r.D.y) * 2.0f * u + eb * fabsf(r.D.y) + u * (fabsf(p.y) + fabsf(pa.y) + fabsf(pb.y))) + 1e-7f;
    if (ax > 0 && (fabsf(p.x - pa.x - eps) <= mx || fabsf(pb.x - p.x - eps) <= mx)) return true;
    if (ax > 1 && (fabsf(p.y - pa.y - eps) <= my || fabsf(pb.y - p.y - eps) <= my)) return true;
    float w = ax == 0 ? pb.x - pa.x : (ax == 1 ? pb.y - pa.y : pb.z - pa.z);
    if (fabsf(w - eps) <= 1e-5f) return true;  // degenerate slab as thin as the epsilon
  } else {
    // n = (p - C)/|p - C|: lateral error of p over R, plus the error along the ray scaled by
    // the sine of the angle between ray and normal (= |l| / R)
    f3 C = mk3(pa.x, pa.y, pa.z);
    f3 oc = r.O - C;
    float k = dot3(oc, r.D) * r.inva;
    f3 l = fma3(r.D, -k, oc);
    float R = fabsf(pa.w), sinphi = fminf(1.0f, sqrt32(dot3(l, l)) * rcp32(R));
    float lateral = fabsf(tb) * 2.0f * u * Dlen + 2.0f * u * (sqrt32(dot3(p, p)) + sqrt32(dot3(C, C)));
    float en = (lateral + eb * Dlen * sinphi) * rcp32(R) + 6.0f * u;
    if (en > 8e-6f) return true;
  }
  return false;
}

GRT_DEV bool ambiguous32(const KScene& S, const Ray32& r, int id_fast, float t_fast) {
  const float BIG = 3.402823466e38f;
  const float Dlen = sqrt32(r.a);
  const float Olen = sqrt32(dot3(r.O, r.O));
  bool amb = ray_has_zero(r);
  float tb = BIG, eb = 0.0f;  // running best t and its error bound
  int best = -1;
  for (int i = 0; i < S.n_prims; ++i) {
    float t, e;
    const int st = hit_margin(S.prims[2 * i], S.prims[2 * i + 1], r, S.bias, Dlen, Olen, tb - eb, tb + eb, t, e);
    if (st == 2) amb = true;
    if (st == 1) best = i, tb = t, eb = e;
  }
  if (best != id_fast) amb = true;  // cannot happen for unambiguous lanes; belt and braces
  if (best < 0 || amb) return amb;
  return hit_accuracy_doubtful(S, r, best, tb, eb, t_fast);
}
