/*
 * grt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C float64 restatement of the go-raytracer per-pixel rendering hot path, written
 * from the reference's source text (no Go toolchain exists in the build container, so the
 * reference itself cannot be run).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; libgrt_hip.so never does.
 *
 * PARITY PIN.  The reference holds no tests, golden vectors or fixtures for this path
 * (SURVEY.md §4).  The restatement is pinned by (a) known-answer tests derived by hand from
 * the cited formulas (tests/test_oracle_kat.py), and (b) the only artefact the reference
 * ships: images/scene_1000.png (640x640, 1000 spp), compared by PSNR because the
 * reference's RNG streams are time-seeded and unreproducible (scene.go:114, vec3.go:152).
 * The photon path has no artefact at all and depends on gonum v0.9.3 kdtree (absent from
 * /root/reference): for it this file says PARITY UNPINNED.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Arithmetic is float64 with the reference's operation order and must
 * be compiled with -ffp-contract=off (Go on amd64 does not fuse multiply-add).
 *
 * The one deliberate substitution is the RNG: the reference's sequential math/rand streams
 * (scene.go:86,114; global source vec3.go:152) are replaced by the counter-based generator
 * the GPU uses, so that a (pixel, sample, dimension) triple names the same uniform on both
 * sides and per-sample parity becomes testable.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../include/grt.h"

#define GRTO_API __attribute__((visibility("default")))

typedef struct {
  double x, y, z;
} v3;

/* ------------------------------------------------------------------------------------
 * geom.Vec3 (pkg/geom/vec3.go:39-111) and Ray.At (pkg/geom/ray.go:15-17)
 * ---------------------------------------------------------------------------------- */
static inline v3 V(double x, double y, double z) {
  v3 r = {x, y, z};
  return r;
}
static inline v3 v_inv(v3 a) { return V(-a.x, -a.y, -a.z); }                         /* vec3.go:39 */
static inline v3 v_plus(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }    /* :44 */
static inline v3 v_minus(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :49 */
static inline v3 v_times(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :54 */
static inline v3 v_div(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }     /* :59 */
static inline double v_lensq(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }       /* :64 */
static inline double v_len(v3 a) { return sqrt(v_lensq(a)); }                          /* :69 */
static inline v3 v_scale(v3 a, double n) { return V(a.x * n, a.y * n, a.z * n); }      /* :74 */
static inline double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* :79 */
static inline v3 v_cross(v3 a, v3 b) {                                                 /* :84 */
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v_unit(v3 a) { /* vec3.go:92-95: k = 1/Len, Scale(k) */
  double k = 1.0 / v_len(a);
  return v_scale(a, k);
}
static inline int v_nearzero(v3 a) { /* vec3.go:98-101 */
  const double s = 1e-8;
  return fabs(a.x) < s && fabs(a.y) < s && fabs(a.z) < s;
}

/* Go math.Min / math.Max (called from vec3.go:104-111 and aabb.go:30-31, color.go:53):
 * +-Inf beats NaN, otherwise NaN propagates, and -0 < +0. */
static double go_max(double x, double y) {
  if ((isinf(x) && x > 0) || (isinf(y) && y > 0)) return INFINITY;
  if (isnan(x) || isnan(y)) return NAN;
  if (x == 0 && x == y) return signbit(x) ? y : x;
  return x > y ? x : y;
}
static double go_min(double x, double y) {
  if ((isinf(x) && x < 0) || (isinf(y) && y < 0)) return -INFINITY;
  if (isnan(x) || isnan(y)) return NAN;
  if (x == 0 && x == y) return signbit(x) ? x : y;
  return x < y ? x : y;
}
static inline v3 v_min(v3 a, v3 b) { return V(go_min(a.x, b.x), go_min(a.y, b.y), go_min(a.z, b.z)); }
static inline v3 v_max(v3 a, v3 b) { return V(go_max(a.x, b.x), go_max(a.y, b.y), go_max(a.z, b.z)); }

typedef struct {
  v3 orig, dir;
} ray;
static inline v3 ray_at(ray r, double t) { return v_plus(r.orig, v_scale(r.dir, t)); } /* ray.go:15 */

static inline v3 from3(const double* p) { return V(p[0], p[1], p[2]); }
static inline void to3(double* p, v3 a) {
  p[0] = a.x;
  p[1] = a.y;
  p[2] = a.z;
}

/* ------------------------------------------------------------------------------------
 * Counter-based RNG (substitution for math/rand, see header).  Stated identically in
 * go-raytracer_amd/csrc/grt_rng.h; tests/test_rng.py checks both agree.
 * ---------------------------------------------------------------------------------- */
static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
static inline uint32_t rng_seed32(uint64_t seed) {
  return mix32((uint32_t)seed ^ 0x9e3779b9U) ^ mix32((uint32_t)(seed >> 32) + 0x85ebca6bU);
}
static inline uint32_t rng_pixel_key(uint32_t seed32, uint32_t pixel) {
  return mix32(seed32 ^ (pixel * 0x9E3779B1U));
}
/* Two-word path key: no two (pixel, sample) pairs of a frame share it (csrc/grt_rng.h). */
typedef struct {
  uint32_t a, b;
} rng_key;
static inline rng_key rng_path_key(uint32_t pixel_key, uint32_t sample) {
  rng_key k;
  k.a = mix32(pixel_key + sample * 0x85EBCA77U + 0xC2B2AE3DU);
  k.b = pixel_key + sample * 0x27D4EB2EU + 0x165667B1U;
  return k;
}
/* One hash per PAIR of uniforms: dimensions 2p and 2p + 1 are the low and the high 16 bits of hash(key, p). */
static inline double rng_draw(rng_key path_key, uint32_t dim) {
  uint32_t h = path_key.a + ((dim >> 1) + 1U) * 0x9E3779B9U;
  h ^= h >> 16;
  h *= 0x7feb352dU;
  h ^= (h >> 15) ^ path_key.b;
  h *= 0x846ca68bU;
  h ^= h >> 16;
  return (double)((dim & 1U) ? h >> 16 : h & 0xffffU) * (1.0 / 65536.0);
}
GRTO_API double grto_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
  return rng_draw(rng_path_key(rng_pixel_key(rng_seed32(seed), pixel), sample), dim);
}

/* Dimension layout of one camera path: 0,1 = pixel jitter (scene.go:93-94); the hit at depth d
 * (1-based, scene.go:237) owns 2+2(d-1)+{0,1}: SampleHemisphereNormal's two draws in call
 * order (vec3.go:203-204), or - the material branches of trace() are exclusive - Refract's
 * roulette draw (vec3.go:152) in slot 0. */
#define DIM_JITTER_U 0u
#define DIM_JITTER_V 1u
#define DIM_HIT(depth, k) (2u + 2u * (uint32_t)((depth)-1) + (uint32_t)(k))

/* ------------------------------------------------------------------------------------
 * Camera (pkg/tracer/camera.go)
 * ---------------------------------------------------------------------------------- */
GRTO_API void grto_camera_new(const double* eye, const double* lookat, const double* vup, double vfov,
                              double aspect, grt_camera* c) { /* camera.go:16-30 */
  double theta = vfov * M_PI / 180;
  double halfH = tan(theta / 2);
  double halfW = aspect * halfH;
  v3 e = from3(eye);
  v3 w = v_unit(v_minus(e, from3(lookat)));
  v3 u = v_unit(v_cross(from3(vup), w));
  v3 v = v_unit(v_cross(u, w));
  to3(c->origin, e);
  to3(c->lower_left, v_minus(v_minus(v_minus(e, v_scale(u, halfW)), v_scale(v, halfH)), w));
  to3(c->horizontal, v_scale(u, 2 * halfW));
  to3(c->vertical, v_scale(v, 2 * halfH));
}

static ray camera_ray(const grt_camera* c, double u, double v) { /* camera.go:34-39 */
  ray r;
  r.orig = from3(c->origin);
  r.dir = v_minus(v_plus(v_plus(from3(c->lower_left), v_scale(from3(c->horizontal), u)),
                         v_scale(from3(c->vertical), v)),
                  from3(c->origin));
  return r;
}
GRTO_API void grto_camera_ray(const grt_camera* c, double u, double v, double* orig, double* dir) {
  ray r = camera_ray(c, u, v);
  to3(orig, r.orig);
  to3(dir, r.dir);
}

/* ------------------------------------------------------------------------------------
 * Primitives (pkg/tracer/sphere.go, pkg/tracer/aabb.go)
 * ---------------------------------------------------------------------------------- */
static double sphere_hit(const grt_prim* s, ray r, double tMin, double tMax) { /* sphere.go:23-46 */
  v3 oc = v_minus(r.orig, from3(s->a));
  double a = v_lensq(r.dir);
  double halfB = v_dot(oc, r.dir);
  double c = v_lensq(oc) - s->radius * s->radius;
  double disc = halfB * halfB - a * c;
  if (disc < 0.0) return -1.0;
  double sqrtd = sqrt(disc);
  double t = (-halfB - sqrtd) / a;
  if (t > tMin && t < tMax) return t;
  t = (-halfB + sqrtd) / a;
  if (t > tMin && t < tMax) return t;
  return -1.0;
}
static v3 sphere_surface(const grt_prim* s, v3 p) { /* sphere.go:56-58 */
  return v_unit(v_scale(v_minus(p, from3(s->a)), s->radius));
}

static double aabb_hit(const grt_prim* b, ray r, double tMin, double tMax) { /* aabb.go:25-45 */
  v3 n = v_div(v_minus(from3(b->a), r.orig), r.dir);
  v3 f = v_div(v_minus(from3(b->b), r.orig), r.dir);
  v3 n2 = v_min(n, f), f2 = v_max(n, f);
  double t0 = go_max(go_max(n2.x, n2.y), n2.z);
  double t1 = go_min(go_min(f2.x, f2.y), f2.z);
  double ltMin = tMin, ltMax = tMax;
  if (t0 > ltMin) ltMin = t0;
  if (t1 < ltMax) ltMax = t1;
  if (ltMax <= ltMin) return -1.0;
  return t0; /* the ENTRY distance, even when below tMin (aabb.go:44) */
}
static v3 aabb_surface(const grt_prim* b, v3 p) { /* aabb.go:55-72 */
  const double eps = 1e-4;
  if (p.x < b->a[0] + eps) return V(-1, 0, 0);
  if (p.x > b->b[0] - eps) return V(1, 0, 0);
  if (p.y < b->a[1] + eps) return V(0, -1, 0);
  if (p.y > b->b[1] - eps) return V(0, 1, 0);
  if (p.z < b->a[2] + eps) return V(0, 0, -1);
  if (p.z > b->b[2] - eps) return V(0, 0, 1);
  return V(0, 0, 0);
}
static double prim_hit(const grt_prim* p, ray r, double tMin, double tMax) {
  return p->kind == GRT_KIND_SPHERE ? sphere_hit(p, r, tMin, tMax) : aabb_hit(p, r, tMin, tMax);
}
static v3 prim_surface(const grt_prim* p, v3 pt) {
  return p->kind == GRT_KIND_SPHERE ? sphere_surface(p, pt) : aabb_surface(p, pt);
}
static v3 prim_pos(const grt_prim* p) { /* sphere.go:52-54; aabb.go:19,51-53 */
  if (p->kind == GRT_KIND_SPHERE) return from3(p->a);
  return v_scale(v_plus(from3(p->b), from3(p->a)), 0.5);
}

GRTO_API double grto_prim_hit(const grt_prim* p, const double* orig, const double* dir, double tMin,
                              double tMax) {
  ray r = {from3(orig), from3(dir)};
  return prim_hit(p, r, tMin, tMax);
}
GRTO_API void grto_prim_surface(const grt_prim* p, const double* pt, double* n) {
  to3(n, prim_surface(p, from3(pt)));
}

/* ------------------------------------------------------------------------------------
 * Scene (pkg/tracer/scene.go:17-64)
 * ---------------------------------------------------------------------------------- */
struct grto_scene {
  grt_scene_desc d;
  grt_prim* prims;
  grt_material* mats;
  int32_t n_prims, n_mats;
  int32_t* lights; /* indices of objects with Emittance > 0 (scene.go:42-46) */
  int32_t n_lights;
  int32_t* tobjects; /* indices of Transparent objects (scene.go:48-50) */
  int32_t n_tobjects;
  double light_area;
  /* scene.globalPmap / scene.causticPmap (scene.go:22-23); NULL = the photon block of trace()
   * stays commented out as in the reference (scene.go:308-320) */
  const struct grto_pmap* global_pmap;
  const struct grto_pmap* caustic_pmap;
  /* photon-map view (no caller in the reference): 0 off; 1 = Render with irradiance() (scene.go:184-232) in
   * place of trace(); 2 = trace() with its debugging block enabled (scene.go:261-271): IrradianceEst(p, n, 0, 50)
   * of view_pmap at the first hit */
  int view;
  const struct grto_pmap* view_pmap;
};
typedef struct grto_scene grto_scene;
static void irradiance_est(const struct grto_pmap* m, const double* pos, const double* nrm, double radius, int k, int brute,
                           double* out);
/* Enables the photon-mapping block of trace() (scene.go:308-320) with these two maps. */
GRTO_API void grto_scene_set_pmaps(grto_scene* s, const struct grto_pmap* global_map, const struct grto_pmap* caustic_map) {
  s->global_pmap = global_map;
  s->caustic_pmap = caustic_map;
}

GRTO_API void grto_scene_set_view(grto_scene* s, const struct grto_pmap* map, int32_t mode) {
  s->view = map ? mode : 0;
  s->view_pmap = map;
}

GRTO_API int grto_scene_create(const grt_scene_desc* desc, const grt_prim* prims, int32_t n_prims,
                               const grt_material* mats, int32_t n_mats, grto_scene** out) {
  if (!desc || !out || n_prims < 0 || n_mats < 0 || (n_prims && !prims) || (n_mats && !mats))
    return GRT_EINVAL;
  for (int i = 0; i < n_prims; i++)
    if (prims[i].material < 0 || prims[i].material >= n_mats) return GRT_EINVAL;
  grto_scene* s = (grto_scene*)calloc(1, sizeof *s);
  s->d = *desc;
  s->n_prims = n_prims;
  s->n_mats = n_mats;
  s->prims = (grt_prim*)malloc(sizeof(grt_prim) * (size_t)(n_prims ? n_prims : 1));
  s->mats = (grt_material*)malloc(sizeof(grt_material) * (size_t)(n_mats ? n_mats : 1));
  memcpy(s->prims, prims, sizeof(grt_prim) * (size_t)n_prims);
  memcpy(s->mats, mats, sizeof(grt_material) * (size_t)n_mats);
  s->lights = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_prims ? n_prims : 1));
  s->tobjects = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_prims ? n_prims : 1));
  for (int i = 0; i < n_prims; i++) { /* scene.go:40-51 */
    const grt_material* m = &s->mats[s->prims[i].material];
    if (m->emittance > 0) {
      s->lights[s->n_lights++] = i;
      s->light_area += m->color[0] + m->color[1] + m->color[2];
    }
    if (m->transparent) s->tobjects[s->n_tobjects++] = i;
  }
  *out = s;
  return GRT_OK;
}
GRTO_API void grto_scene_destroy(grto_scene* s) {
  if (!s) return;
  free(s->prims);
  free(s->mats);
  free(s->lights);
  free(s->tobjects);
  free(s);
}

/* Scene.intersect (scene.go:169-181) over an index list (NULL = all objects). */
static int intersect(const grto_scene* s, ray r, const int32_t* list, int n, double* t_out) {
  double tMin = s->d.bias, tMax = 1.7976931348623157e308; /* math.MaxFloat64 */
  double t = tMax;
  int hit = -1;
  for (int k = 0; k < n; k++) {
    int i = list ? list[k] : k;
    double ht = prim_hit(&s->prims[i], r, tMin, t);
    if (ht > 0.0) {
      hit = i;
      t = ht;
    }
  }
  *t_out = t;
  return hit;
}
GRTO_API int grto_intersect(const grto_scene* s, const double* orig, const double* dir, double* t) {
  ray r = {from3(orig), from3(dir)};
  return intersect(s, r, NULL, s->n_prims, t);
}

/* ------------------------------------------------------------------------------------
 * BSDF sampling (pkg/geom/vec3.go:129-159, 202-218)
 * ---------------------------------------------------------------------------------- */
static v3 reflect(v3 v, v3 n) { /* vec3.go:129-131 */
  return v_unit(v_minus(v, v_scale(n, 2 * v_dot(v, n))));
}
/* vec3.go:134-159.  xi stands for the rand.Float64() of vec3.go:152 (not consumed on TIR —
 * immaterial for a counter-based stream). */
static int refract(v3 v, v3 n, double etaRatio, double xi, v3* out) {
  v3 refrN = n;
  double ratio = etaRatio;
  if (v_dot(v, n) >= 0) {
    refrN = v_inv(n);
    ratio = 1.0 / ratio;
  }
  double cosi = go_min(v_dot(v_inv(v), refrN), 1.0);
  double sini = sqrt(1.0 - cosi * cosi);
  int tir = ratio * sini > 1.0;
  double r0 = (1 - ratio) / (1 + ratio);
  r0 = r0 * r0;
  r0 = r0 + (1 - r0) * pow(1 - cosi, 5);
  if (tir || r0 > xi) {
    *out = V(0, 0, 0);
    return 0;
  }
  v3 r1 = v_scale(v_plus(v, v_scale(refrN, cosi)), ratio);
  v3 r2 = v_scale(refrN, -1 * sqrt(fabs(1.0 - v_lensq(r1))));
  *out = v_unit(v_plus(r1, r2));
  return 1;
}
static v3 sample_hemisphere_normal(v3 n, double u1, double u2) { /* vec3.go:202-218 */
  double r1 = 2 * M_PI * u1;
  double r2 = u2;
  double r2s = sqrt(r2);
  v3 w = n;
  v3 u = V(1, 0, 0);
  if (fabs(w.x) > 0.1) u = V(0, 1, 0);
  u = v_unit(v_cross(u, w));
  v3 v = v_cross(w, u);
  v3 uc = v_scale(u, cos(r1) * r2s);
  v3 vc = v_scale(v, sin(r1) * r2s);
  v3 wc = v_scale(w, sqrt(1 - r2));
  return v_unit(v_plus(v_plus(uc, vc), wc));
}
GRTO_API void grto_reflect(const double* v, const double* n, double* out) {
  to3(out, reflect(from3(v), from3(n)));
}
GRTO_API int grto_refract(const double* v, const double* n, double ratio, double xi, double* out) {
  v3 o;
  int ok = refract(from3(v), from3(n), ratio, xi, &o);
  to3(out, o);
  return ok;
}
GRTO_API void grto_sample_hemisphere_normal(const double* n, double u1, double u2, double* out) {
  to3(out, sample_hemisphere_normal(from3(n), u1, u2));
}

/* ------------------------------------------------------------------------------------
 * Scene.trace (scene.go:237-349), recursive as in the reference.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  rng_key path_key;
  int n_rays; /* intersect() calls (scene.go:242), shadow loops not counted */
} pathctx;

/* Scene.irradiance (scene.go:184-232; dead code upstream): direct visualisation of a photon map through
 * specular chains.  Note what it does NOT test: Normal, Emittance and Lambert materials are "diffuse" here. */
static v3 irradiance(const grto_scene* s, const struct grto_pmap* pmap, ray r, int depth, pathctx* px) {
  v3 black = V(0, 0, 0);
  if (depth >= s->d.max_depth) return black; /* :187-189 */
  double tNear;
  px->n_rays++;
  int id = intersect(s, r, NULL, s->n_prims, &tNear); /* :191 */
  if (id < 0) return black;
  v3 incident = v_unit(r.dir); /* :197 */
  v3 p = ray_at(r, tNear);
  const grt_prim* o = &s->prims[id];
  v3 n = prim_surface(o, p);
  const grt_material* m = &s->mats[o->material];
  v3 f = from3(m->color); /* :205 */
  if (m->reflectivity > 0) { /* :207-214 */
    v3 refl = reflect(incident, n);
    v3 fz = sample_hemisphere_normal(n, rng_draw(px->path_key, DIM_HIT(depth, 0)), rng_draw(px->path_key, DIM_HIT(depth, 1)));
    refl = v_plus(refl, v_scale(fz, m->roughness));
    if (v_dot(refl, n) > 0) {
      ray r2 = {p, refl};
      return v_times(f, irradiance(s, pmap, r2, depth + 1, px));
    }
  } else if (m->transparent) { /* :215-224 */
    double refrRatio = 1.0 / m->refr_index;
    v3 dir;
    int ok = refract(incident, n, refrRatio, rng_draw(px->path_key, DIM_HIT(depth, 0)), &dir);
    if (!ok) dir = reflect(incident, n);
    ray r2 = {p, dir};
    return irradiance(s, pmap, r2, depth + 1, px);
  } else { /* :225-230 */
    double pp[3], nn[3], e[3];
    to3(pp, p);
    to3(nn, n);
    irradiance_est(pmap, pp, nn, 0, 100, 0, e);
    return from3(e);
  }
  return black;
}

static v3 trace(const grto_scene* s, ray r, int depth, pathctx* px) {
  v3 black = V(0, 0, 0);
  if (depth >= s->d.max_depth) return black; /* scene.go:238-240 */
  double tNear;
  px->n_rays++;
  int id = intersect(s, r, NULL, s->n_prims, &tNear); /* :242 */
  if (id < 0) return black;                           /* :244-250 */
  v3 result = black;
  v3 incident = v_unit(r.dir); /* :253 */
  v3 p = ray_at(r, tNear);     /* :254 */
  const grt_prim* o = &s->prims[id];
  v3 n = prim_surface(o, p); /* :255 */
  const grt_material* m = &s->mats[o->material];
  v3 mc = from3(m->color);
  /* orientedN (:256-259) is computed and never used in trace */
  if (s->view == 2) { /* the debugging block, :261-271 */
    double pp[3], nn[3], e[3];
    to3(pp, p);
    to3(nn, n);
    irradiance_est(s->view_pmap, pp, nn, 0, 50, 0, e);
    return v_plus(result, from3(e));
  }
  if (m->normal) /* :274-276 */
    return v_scale(V(n.x + 0.5, n.y + 0.5, n.z + 0.5), 0.5);
  if (m->emittance > 0) { /* :278-279 */
    result = v_scale(mc, m->emittance);
  } else if (m->lambert) { /* :280-286 */
    v3 sc = sample_hemisphere_normal(n, rng_draw(px->path_key, DIM_HIT(depth, 0)),
                                     rng_draw(px->path_key, DIM_HIT(depth, 1)));
    if (v_nearzero(sc)) sc = n;
    ray r2 = {p, sc};
    result = v_plus(result, v_times(trace(s, r2, depth + 1, px), mc));
  } else if (m->reflectivity > 0) { /* :287-294 */
    v3 refl = reflect(incident, n);
    v3 fz = sample_hemisphere_normal(n, rng_draw(px->path_key, DIM_HIT(depth, 0)),
                                     rng_draw(px->path_key, DIM_HIT(depth, 1)));
    refl = v_plus(refl, v_scale(fz, m->roughness));
    if (v_dot(refl, n) > 0) {
      ray r2 = {p, refl};
      result = v_plus(result, v_scale(v_times(trace(s, r2, depth + 1, px), mc), m->reflectivity));
    }
  } else if (m->transparent) { /* :295-304 */
    double refrRatio = 1.0 / m->refr_index;
    v3 dir;
    int ok = refract(incident, n, refrRatio, rng_draw(px->path_key, DIM_HIT(depth, 0)), &dir);
    if (!ok) dir = reflect(incident, n);
    ray r2 = {p, dir};
    result = v_plus(result, trace(s, r2, depth + 1, px));
  } else {
    if (s->global_pmap && s->caustic_pmap) { /* photon mapping, :308-320 (commented out in the reference) */
      const double BRDF = 1 / M_PI;
      double pp[3], nn[3], c[3], g[3];
      to3(pp, p);
      to3(nn, n);
      irradiance_est(s->caustic_pmap, pp, nn, 1, 100, 0, c); /* :313 */
      irradiance_est(s->global_pmap, pp, nn, 0, 100, 0, g);  /* :316 */
      v3 irrad = v_plus(v_scale(from3(c), BRDF), v_scale(from3(g), BRDF));
      result = v_plus(result, v_times(irrad, mc)); /* :318-319 */
    }
    /* direct illumination, :322-346 */
    for (int li = 0; li < s->n_lights; li++) {
      const grt_prim* l = &s->prims[s->lights[li]];
      v3 pos = prim_pos(l);
      v3 dir = v_unit(v_minus(pos, p));
      v3 power = from3(s->mats[l->material].color);
      double fd = v_dot(n, dir);
      if (fd < 0) fd = 0;
      double visible = 1.0;
      double tMin = s->d.bias, tN = 1.7976931348623157e308;
      ray sh = {p, dir};
      for (int i = 0; i < s->n_prims; i++) {
        double ht = prim_hit(&s->prims[i], sh, tMin, tN);
        if (ht > 0.0) {
          if (s->mats[s->prims[i].material].emittance == 0) visible = 0.0;
          tN = ht;
        }
      }
      result = v_plus(result, v_scale(v_times(v_scale(mc, fd), power), visible));
    }
  }
  return result;
}

/* One pixel-sample: scene.go:93-96. */
static v3 trace_sample(const grto_scene* s, uint32_t seed32, int x, int y, uint32_t sample, int* n_rays) {
  uint32_t pixel = (uint32_t)y * (uint32_t)s->d.width + (uint32_t)x;
  pathctx px = {rng_path_key(rng_pixel_key(seed32, pixel), sample), 0};
  double u = ((double)x + rng_draw(px.path_key, DIM_JITTER_U)) / (double)s->d.width;
  double v = ((double)y + rng_draw(px.path_key, DIM_JITTER_V)) / (double)s->d.height;
  ray r = camera_ray(&s->d.cam, u, v);
  v3 c = s->view == 1 ? irradiance(s, s->view_pmap, r, 1, &px) : trace(s, r, 1, &px);
  if (n_rays) *n_rays = px.n_rays;
  return c;
}

GRTO_API int grto_trace_samples(const grto_scene* s, uint64_t seed, const int32_t* pixel_xy,
                                int32_t n_pixels, int32_t samples, double* rgb, int32_t* n_rays) {
  if (!s || !pixel_xy || !rgb || n_pixels < 0 || samples < 0) return GRT_EINVAL;
  uint32_t seed32 = rng_seed32(seed);
  for (int i = 0; i < n_pixels; i++)
    for (int k = 0; k < samples; k++) {
      int nr = 0;
      v3 c = trace_sample(s, seed32, pixel_xy[2 * i], pixel_xy[2 * i + 1], (uint32_t)k, &nr);
      to3(&rgb[((size_t)i * (size_t)samples + (size_t)k) * 3], c);
      if (n_rays) n_rays[(size_t)i * (size_t)samples + (size_t)k] = nr;
    }
  return GRT_OK;
}

/* ------------------------------------------------------------------------------------
 * Scene.Render (scene.go:79-136) and WriteColor (scene.go:67-74)
 * ---------------------------------------------------------------------------------- */
GRTO_API void grto_write_color(const double* mean_rgb, uint8_t* bgr) {
  /* scene.go:98-99: Scale(1/samples) already applied by the caller; Gamma(2) = Pow(c, 0.5)
   * (color.go:57-64; Go's Pow special-cases y == 0.5 to Sqrt); Clamp = Min(1, c)
   * (color.go:52-54); scene.go:68-73: uint8(255.99*c) truncation, byte order B,G,R. */
  double c[3];
  for (int k = 0; k < 3; k++) c[k] = go_min(1.0, sqrt(mean_rgb[k]));
  /* A negative mean (NormalMaterial: (n+0.5)*0.5 with n=-1, scene.go:275) gives NaN; Go leaves
   * uint8(NaN) implementation-defined, amd64 yields 0 (CVTTSD2SL = 0x80000000, low byte). */
  for (int k = 0; k < 3; k++)
    if (isnan(c[k])) c[k] = 0.0;
  bgr[0] = (uint8_t)(255.99 * c[2]);
  bgr[1] = (uint8_t)(255.99 * c[1]);
  bgr[2] = (uint8_t)(255.99 * c[0]);
}

typedef struct {
  const grto_scene* s;
  uint32_t seed32;
  int samples, row_begin, row_end;
  volatile int* next_row;
  double* rgb_sum;    /* may be NULL */
  uint8_t* pixels;    /* may be NULL */
  int pitch, bpp;
  uint64_t n_rays;
} job;

static void* worker(void* arg) { /* scene.go:86-104 */
  job* j = (job*)arg;
  const grto_scene* s = j->s;
  const int W = s->d.width;
  uint64_t rays = 0;
  for (;;) {
    int y = __sync_fetch_and_add(j->next_row, 1);
    if (y >= j->row_end) break;
    for (int x = 0; x < W; x++) {
      v3 c = V(0, 0, 0);
      for (int k = 0; k < j->samples; k++) {
        int nr;
        c = v_plus(c, trace_sample(s, j->seed32, x, y, (uint32_t)k, &nr)); /* :93-96 */
        rays += (uint64_t)nr;
      }
      if (j->rgb_sum) to3(&j->rgb_sum[((size_t)y * (size_t)W + (size_t)x) * 3], c);
      if (j->pixels) {
        v3 m = v_scale(c, 1 / (double)j->samples); /* :98 */
        double mm[3];
        to3(mm, m);
        grto_write_color(mm, &j->pixels[(size_t)y * (size_t)j->pitch + (size_t)x * (size_t)j->bpp]);
      }
    }
  }
  j->n_rays = rays;
  return NULL;
}

static uint64_t g_last_rays;
GRTO_API uint64_t grto_last_ray_count(void) { return g_last_rays; }

/* nthreads <= 0 -> NumCPU()+1 as in scene.go:106. */
static int render_rows(const grto_scene* s, uint64_t seed, int samples, int row_begin, int row_end,
                       double* rgb_sum, uint8_t* pixels, int pitch, int nthreads) {
  if (!s || samples <= 0 || row_begin < 0 || row_end > s->d.height) return GRT_EINVAL;
  if (nthreads <= 0) nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN) + 1;
  if (nthreads > 1024) nthreads = 1024;
  volatile int next = row_begin;
  job* jobs = (job*)calloc((size_t)nthreads, sizeof(job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int i = 0; i < nthreads; i++) {
    job j = {s, rng_seed32(seed), samples, row_begin, row_end, &next, rgb_sum, pixels, pitch,
             pixels ? pitch / s->d.width : 0, 0};
    jobs[i] = j;
    pthread_create(&th[i], NULL, worker, &jobs[i]);
  }
  uint64_t rays = 0;
  for (int i = 0; i < nthreads; i++) {
    pthread_join(th[i], NULL);
    rays += jobs[i].n_rays;
  }
  g_last_rays = rays;
  free(jobs);
  free(th);
  return GRT_OK;
}

GRTO_API int grto_render(const grto_scene* s, uint64_t seed, uint8_t* pixels, int32_t pitch,
                         int32_t samples, int32_t nthreads) {
  if (!s || !pixels || pitch < 3 * s->d.width) return GRT_EINVAL;
  return render_rows(s, seed, samples, 0, s->d.height, NULL, pixels, pitch, nthreads);
}
/* rgb_sum is the full W*H*3 image; only rows [row_begin,row_end) are written. */
GRTO_API int grto_render_accum(const grto_scene* s, uint64_t seed, double* rgb_sum, int32_t samples,
                               int32_t row_begin, int32_t row_end, int32_t nthreads) {
  if (!rgb_sum) return GRT_EINVAL;
  return render_rows(s, seed, samples, row_begin, row_end, rgb_sum, NULL, 0, nthreads);
}

/* Primary rays only: Camera.Ray + intersect + Surface (config C2). */
GRTO_API int grto_primary_hits(const grto_scene* s, uint64_t seed, int32_t jitter_mode, int32_t* id,
                               double* t, double* nrm) {
  if (!s || !id || !t || !nrm) return GRT_EINVAL;
  const int W = s->d.width, H = s->d.height;
  uint32_t seed32 = rng_seed32(seed);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      double ju = 0.5, jv = 0.5;
      if (jitter_mode == GRT_JITTER_RNG) {
        rng_key pk = rng_path_key(rng_pixel_key(seed32, (uint32_t)y * (uint32_t)W + (uint32_t)x), 0);
        ju = rng_draw(pk, DIM_JITTER_U);
        jv = rng_draw(pk, DIM_JITTER_V);
      }
      ray r = camera_ray(&s->d.cam, ((double)x + ju) / (double)W, ((double)y + jv) / (double)H);
      double tn;
      int h = intersect(s, r, NULL, s->n_prims, &tn);
      size_t i = (size_t)y * (size_t)W + (size_t)x;
      id[i] = h;
      v3 n = V(0, 0, 0);
      if (h >= 0) n = prim_surface(&s->prims[h], ray_at(r, tn));
      t[i] = h >= 0 ? tn : 0.0;
      to3(&nrm[i * 3], n);
    }
  return GRT_OK;
}

#include "grt_oracle_photon.h"

GRTO_API const char* grto_version(void) { return "grt-oracle 0.1 (float64 C restatement; test infrastructure)"; }
