// grt_api.hip — the C ABI of include/grt.h over the kernels in grt_kernels.h.
// gfx950 only; no torch types, no CPU fallback: every pixel comes out of a HIP kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/grt.h"
#include "grt_internal.h"
#include "grt_kernels.h"

#define GRT_API extern "C" __attribute__((visibility("default")))

// ---- error channel ---------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local uint64_t g_last_rays = 0;
static thread_local double g_last_ms = 0.0;

int grt_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// Every int-returning entry point is a function-try-block that ends here: a C++ exception (a vector
// that cannot grow for 10^7 objects, a thread that cannot start) must not unwind into a C or Go caller.
int grt_fail_exception(const char* entry_point) {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    return grt_fail(GRT_EHOST, "%s: out of host memory", entry_point);
  } catch (const std::exception& e) {
    return grt_fail(GRT_EHOST, "%s: %s", entry_point, e.what());
  } catch (...) {
    return grt_fail(GRT_EHOST, "%s: unknown C++ exception", entry_point);
  }
}
GRT_API const char* grt_last_error(void) { return g_err.c_str(); }
GRT_API const char* grt_version(void) { return "grt-hip 0.1 (gfx950)"; }
GRT_API uint64_t grt_last_ray_count(void) { return g_last_rays; }
GRT_API double grt_last_kernel_ms(void) { return g_last_ms; }
GRT_API int grt_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} catch (...) {
  return grt_fail_exception("grt_device_count");
}
GRT_API double grt_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
  return grt_rng_draw64(grt_rng_path_key(grt_rng_pixel_key(grt_rng_seed32(seed), pixel), sample), dim);
}

// ---- library scratch for the entry point that allocates nothing ---------------------------
// grt_render_accum_noalloc_dev keeps the per-chunk partial sums (the "partial sums" buffer of DESIGN.md §4,
// which grt_render_accum_dev takes from the caller's workspace) in this file-scope device array.  Every GPU
// the library runs on holds its own copy, at its own address, until the process ends.  In host code the
// array's NAME is only a stand-in: the device address comes from hipGetSymbolAddress, asked once per device
// with that device current, and is what kernels and memsets are given.
// (GRT_NOALLOC_SCRATCH_BYTES: include/grt.h)
#define GRT_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[GRT_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[GRT_NOALLOC_MAX_DEVICES] = {};

// The current device's copy of `scratch`.  First call on a device: resolves and keeps the address
// (stage() calls this when a scene is staged on a device, so a later no-alloc call only reads the table).
static int scratch_of_current_device(void** out) {
  int dev = 0;
  GRT_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= GRT_NOALLOC_MAX_DEVICES) return grt_fail(GRT_ELIMIT, "device ordinal %d beyond %d", dev, GRT_NOALLOC_MAX_DEVICES - 1);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    GRT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return grt_fail(GRT_EHIP, "hipGetSymbolAddress(scratch) returned NULL on device %d", dev);
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return GRT_OK;
}

// ---- scene staging ---------------------------------------------------------------------
static int validate_scene(const grt_scene_desc* d, const grt_prim* prims, int n_prims, const grt_material* mats,
                          int n_mats) {
  if (!d) return grt_fail(GRT_EINVAL, "grt_scene_create: desc is NULL");
  if (d->width <= 0 || d->height <= 0) return grt_fail(GRT_EINVAL, "grt_scene_create: bad size %dx%d", d->width, d->height);
  if ((int64_t)d->width * d->height > (int64_t)1 << 28) return grt_fail(GRT_ELIMIT, "grt_scene_create: image above 2^28 pixels");
  if (n_prims < 0 || n_mats < 0 || (n_prims && !prims) || (n_mats && !mats))
    return grt_fail(GRT_EINVAL, "grt_scene_create: bad prim/material arrays");
  if (n_prims > GRT_MAX_PRIMS) return grt_fail(GRT_ELIMIT, "grt_scene_create: %d objects; at most %d", n_prims, GRT_MAX_PRIMS);
  if (d->accel != GRT_ACCEL_AUTO && d->accel != GRT_ACCEL_BVH && d->accel != GRT_ACCEL_LINEAR)
    return grt_fail(GRT_EINVAL, "grt_scene_create: unknown accel %d", d->accel);
  if (d->accel == GRT_ACCEL_LINEAR && n_prims > GRT_MAX_LINEAR_PRIMS)
    return grt_fail(GRT_ELIMIT, "grt_scene_create: the linear scan takes at most %d objects (got %d)", GRT_MAX_LINEAR_PRIMS, n_prims);
  // the reference's bias is the constant 0.001 (scene.go:14); the sphere test's three-compare form (sphere_hit32) needs it >= 0
  if (!(d->bias >= 0.0) || !std::isfinite(d->bias)) return grt_fail(GRT_EINVAL, "grt_scene_create: bias %g; it must be finite and >= 0", d->bias);
  for (int i = 0; i < n_prims; ++i) {
    if (prims[i].kind != GRT_KIND_SPHERE && prims[i].kind != GRT_KIND_AABB)
      return grt_fail(GRT_EINVAL, "grt_scene_create: object %d has unknown kind %d", i, prims[i].kind);
    if (prims[i].material < 0 || prims[i].material >= n_mats)
      return grt_fail(GRT_EINVAL, "grt_scene_create: object %d material %d out of range", i, prims[i].material);
  }
  return GRT_OK;
}

GRT_API int grt_scene_create(const grt_scene_desc* desc, const grt_prim* prims, int32_t n_prims,
                             const grt_material* mats, int32_t n_mats, grt_scene** out) try {
  if (!out) return grt_fail(GRT_EINVAL, "grt_scene_create: out is NULL");
  int rc = validate_scene(desc, prims, n_prims, mats, n_mats);
  if (rc) return rc;
  int diffuse_obj = -1;  // a plain DiffuseMaterial object (material.go:20): direct light with shadow rays, photon storage
  for (int i = 0; i < n_prims && diffuse_obj < 0; ++i) {
    const grt_material& m = mats[prims[i].material];
    if (!m.normal && !(m.emittance > 0) && !m.lambert && !(m.reflectivity > 0) && !m.transparent) diffuse_obj = i;
  }
  const bool bvh_forced = n_prims > GRT_MAX_LINEAR_PRIMS || (desc->accel == GRT_ACCEL_BVH && n_prims > 0);
  const bool use_bvh = bvh_forced || (desc->accel == GRT_ACCEL_AUTO && n_prims > GRT_BVH_AUTO_PRIMS);
  std::unique_ptr<grt_scene> owner(new grt_scene);  // released to the caller only when everything below has succeeded
  grt_scene* s = owner.get();
  s->d = *desc;
  s->use_bvh = use_bvh;
  s->bvh_forced = bvh_forced;
  s->has_diffuse = diffuse_obj >= 0;
  s->prims.assign(prims, prims + n_prims);
  s->mats.assign(mats, mats + n_mats);
  for (int i = 0; i < n_prims; ++i) {  // NewScene, scene.go:40-51
    const grt_material& m = s->mats[s->prims[i].material];
    if (m.emittance > 0) {
      s->lights.push_back(i);
      s->light_area += m.color[0] + m.color[1] + m.color[2];
    }
    if (m.transparent) s->tobjects.push_back(i);
  }
  // What one path can add to its pixel at most: throughput (a product of <= max_depth colours x reflectivities, each
  // counted as >= 1) x the brightest source — an emitter's Color x Emittance (scene.go:279), NormalMaterial's 0.75
  // (:275), or the direct light of the diffuse branch, <= its Color x the sum of the lights' colours (:322-346).
  {
    double gain = 1.0, source = 0.0, light_sum = 0.0, diffuse = 0.0;
    auto cmax = [](const grt_material& m) { return std::max(std::fabs(m.color[0]), std::max(std::fabs(m.color[1]), std::fabs(m.color[2]))); };
    for (int i : s->lights) light_sum += cmax(s->mats[s->prims[i].material]);
    for (const grt_material& m : s->mats) {
      gain = std::max(gain, cmax(m) * std::max(1.0, std::fabs(m.reflectivity)));
      if (m.emittance > 0) source = std::max(source, cmax(m) * m.emittance);
      if (m.normal) source = std::max(source, 0.75);  // (n + 0.5) * 0.5 with |n| <= 1 (scene.go:275)
      const bool plain = !m.normal && !(m.emittance > 0) && !m.lambert && !(m.reflectivity > 0) && !m.transparent;
      if (plain) diffuse = std::max(diffuse, cmax(m) * light_sum);  // only the plain-diffuse branch adds the lights' COLOURS (scene.go:322-346)
    }
    const double bound = std::pow(gain, std::max(1, desc->max_depth)) * std::max(source, diffuse);
    // log2 of the bound, rounded up; dim scenes give a negative value and a finer fixed-point unit.  A scene that can
    // return nothing (no light, no Normal material) or a bound that is not finite is marked (INT_MIN / 1024).
    s->radiance_log2 = !std::isfinite(bound) ? 1024 : (bound > 0.0 ? (int)std::ceil(std::log2(bound)) : INT32_MIN);
  }
  *out = owner.release();
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_scene_create");
}

// hipFree(nullptr) is a no-op: a partly staged scene is released the same way.
static void free_device_scene(DeviceScene& d) {
  (void)hipFree(d.prims);
  (void)hipFree(d.mats);
  (void)hipFree(d.lights);
  (void)hipFree(d.prims64);
  (void)hipFree(d.bvh_nodes);
  (void)hipFree(d.bvh_prims);
  (void)hipFree(d.tile_live);
  d = DeviceScene();
}

GRT_API void grt_scene_destroy(grt_scene* s) {
  if (!s) return;
  int cur = 0;
  bool have = hipGetDevice(&cur) == hipSuccess;
  for (auto& kv : s->dev) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    free_device_scene(kv.second);
  }
  for (auto& kv : s->rcache) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    HostRenderCache& c = kv.second;
    (void)hipFree(c.d_sum), (void)hipFree(c.d_ws), (void)hipFree(c.d_rays), (void)hipFree(c.d_bgr);
    (void)hipHostFree(c.h_sum), (void)hipHostFree(c.h_bgr);
    if (c.e0) (void)hipEventDestroy(c.e0);
    if (c.e1) (void)hipEventDestroy(c.e1);
  }
  if (have) (void)hipSetDevice(cur);
  delete s;
}

const grt_scene_desc* grt_scene_desc_of(const grt_scene* s) { return &s->d; }

// ---- bounding-volume hierarchy (grt_bvh.h) ----------------------------------------------
// Median split on the axis of largest centroid extent down to leaves of <= GRT_BVH_LEAF
// objects: balanced, so the traversal stack depth is log2(n / leaf) + 1 (checked against
// GRT_BVH_STACK).  Built on the host from the FP32 records the kernels test.
// GRT_TIMING=1: phases of scene staging on stderr (diagnostics)
struct StageClock {
  const bool on = getenv("GRT_TIMING") != nullptr;
  double t = now();
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void lap(const char* what) {
    if (!on) return;
    const double u = now();
    fprintf(stderr, "[grt stage] %-28s %7.1f ms\n", what, (u - t) * 1e3);
    t = u;
  }
};

// fn(i0, i1) over [0, n) cut into one range per worker (host-side passes over 10^6 objects)
template <class F>
static void parallel_ranges(int n, F fn) {
  const int workers = n < 65536 ? 1 : (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (workers == 1) return fn(0, n);
  std::vector<std::future<void>> fs;
  for (int w = 0; w < workers; ++w) {
    const int i0 = (int)((long long)n * w / workers), i1 = (int)((long long)n * (w + 1) / workers);
    fs.push_back(std::async(std::launch::async, [=] { fn(i0, i1); }));
  }
  for (auto& f : fs) f.get();
}

struct BvhItem {
  float k[3];  // lo + hi of the object's box per axis (twice its centre): the split key
  int idx;     // object index
};
struct BvhBuild {
  std::vector<float> lo, hi;       // padded object boxes, 3 per object
  std::vector<BvhItem> items;      // slot -> object, partitioned in place (contiguous: every pass of the build streams)
  std::vector<int> order;          // slot -> object index, once the build is done
  std::vector<float4> nodes;
  std::atomic<int> n_nodes{0};
  std::atomic<int> max_depth{0};
  int leaf = GRT_BVH_LEAF;       // objects per leaf (<= 7: the count lives in 3 bits of the reference)
};
static void bvh_range_box(const BvhBuild& B, int a, int b, float* lo, float* hi) {
  for (int c = 0; c < 3; ++c) lo[c] = 3e38f, hi[c] = -3e38f;
  for (int i = a; i < b; ++i)
    for (int c = 0; c < 3; ++c) {
      lo[c] = std::min(lo[c], B.lo[3 * (size_t)B.items[i].idx + c]);
      hi[c] = std::max(hi[c], B.hi[3 * (size_t)B.items[i].idx + c]);
    }
}
// Returns the reference of the subtree over items[a, b) and its box (the union of its children's).
static int bvh_build(BvhBuild& B, int a, int b, float* lo, float* hi, int depth, int par_depth) {
  if (b - a <= B.leaf) {
    bvh_range_box(B, a, b, lo, hi);
    int cur = B.max_depth.load(std::memory_order_relaxed);  // (a lock here was taken once per leaf: 10^6 times)
    while (depth > cur && !B.max_depth.compare_exchange_weak(cur, depth, std::memory_order_relaxed)) {
    }
    return ~((a << 3) | (b - a));
  }
  float cl[3] = {3e38f, 3e38f, 3e38f}, ch[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = a; i < b; ++i)
    for (int c = 0; c < 3; ++c) cl[c] = std::min(cl[c], B.items[i].k[c]), ch[c] = std::max(ch[c], B.items[i].k[c]);
  int ax = 0;
  if (ch[1] - cl[1] > ch[ax] - cl[ax]) ax = 1;
  if (ch[2] - cl[2] > ch[ax] - cl[ax]) ax = 2;
  const int mid = a + (b - a) / 2;
  // a strict total order (ties by object index): the two halves are the same sets whatever nth_element does inside
  std::nth_element(B.items.begin() + a, B.items.begin() + mid, B.items.begin() + b,
                   [ax](const BvhItem& x, const BvhItem& y) { return x.k[ax] < y.k[ax] || (x.k[ax] == y.k[ax] && x.idx < y.idx); });
  const int me = B.n_nodes.fetch_add(1);
  float l0[3], h0[3], l1[3], h1[3];
  int r0, r1;
  if (par_depth > 0 && b - a > 16384) {
    auto fut = std::async(std::launch::async, [&] { return bvh_build(B, a, mid, l0, h0, depth + 1, par_depth - 1); });
    r1 = bvh_build(B, mid, b, l1, h1, depth + 1, par_depth - 1);
    r0 = fut.get();
  } else {
    r0 = bvh_build(B, a, mid, l0, h0, depth + 1, 0);
    r1 = bvh_build(B, mid, b, l1, h1, depth + 1, 0);
  }
  float4* n = &B.nodes[4 * (size_t)me];
  n[0] = make_float4(l0[0], l0[1], l0[2], __builtin_bit_cast(float, r0));
  n[1] = make_float4(h0[0], h0[1], h0[2], 0.0f);
  n[2] = make_float4(l1[0], l1[1], l1[2], __builtin_bit_cast(float, r1));
  n[3] = make_float4(h1[0], h1[1], h1[2], 0.0f);
  for (int c = 0; c < 3; ++c) lo[c] = std::min(l0[c], l1[c]), hi[c] = std::max(h0[c], h1[c]);
  return me;
}

static int stage_bvh(const grt_scene* s, const std::vector<float4>& hp, DeviceScene& d) {
  const int n = (int)s->prims.size();
  StageClock clk;
  BvhBuild B;
  B.lo.resize(3 * (size_t)n), B.hi.resize(3 * (size_t)n), B.items.resize(n), B.order.resize(n);
  const float eye_r = (float)std::max({std::fabs(s->d.cam.origin[0]), std::fabs(s->d.cam.origin[1]), std::fabs(s->d.cam.origin[2])});
  parallel_ranges(n, [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      const float4 pa = hp[2 * (size_t)i], pb = hp[2 * (size_t)i + 1];
      const bool box = (__builtin_bit_cast(int, pb.w) & 1) != 0;
      const float c[3] = {pa.x, pa.y, pa.z}, e[3] = {pb.x, pb.y, pb.z};
      for (int k = 0; k < 3; ++k) {
        float l = box ? std::min(c[k], e[k]) : c[k] - std::fabs(pa.w);
        float h = box ? std::max(c[k], e[k]) : c[k] + std::fabs(pa.w);
        // padding: the slab arithmetic of a box around a sphere and sphere_hit32 round differently, both
        // with absolute errors that grow with the coordinates of the ray origin (the eye for primary rays)
        const float pad = 4e-6f * (std::fabs(l) + std::fabs(h) + (h - l)) + 1e-6f * eye_r + 1e-30f;
        B.lo[3 * (size_t)i + k] = l - pad, B.hi[3 * (size_t)i + k] = h + pad;
        B.items[i].k[k] = (l - pad) + (h + pad);
      }
      B.items[i].idx = i;
    }
  });
  clk.lap("hierarchy: object boxes");
  B.nodes.resize(4 * (size_t)std::max(1, n));
  float lo[3], hi[3];
  // Leaves of one object (GRT_BVH_LEAF) by default (GRT_BVH_LEAF_OBJS: A/B knob); scenes too deep for the per-lane traversal
  // stack take larger leaves.
  if (const char* e = getenv("GRT_BVH_LEAF_OBJS")) B.leaf = std::max(1, std::min(7, atoi(e)));  // A/B knob
  while (B.leaf < 7 && (((long long)B.leaf << (GRT_BVH_STACK - 2)) < (long long)n)) B.leaf = B.leaf < 2 ? 2 : (B.leaf < 4 ? 4 : 7);
  if (n <= B.leaf) {  // a single leaf still needs a root node: the second child is empty
    bvh_range_box(B, 0, n, lo, hi);
    B.n_nodes = 1;
    B.nodes[0] = make_float4(lo[0], lo[1], lo[2], __builtin_bit_cast(float, ~((0 << 3) | n)));
    B.nodes[1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    B.nodes[2] = make_float4(3e38f, 3e38f, 3e38f, __builtin_bit_cast(float, ~0));
    B.nodes[3] = make_float4(-3e38f, -3e38f, -3e38f, 0.0f);
  } else {
    const int root = bvh_build(B, 0, n, lo, hi, 0, 4);
    if (root != 0) return grt_fail(GRT_EINVAL, "internal: BVH root is node %d", root);
    if (B.max_depth.load() + 1 > GRT_BVH_STACK)
      return grt_fail(GRT_ELIMIT, "grt_scene: hierarchy depth %d exceeds the traversal stack (%d)", B.max_depth.load(), GRT_BVH_STACK);
  }
  clk.lap("hierarchy: median splits");
  // leaf-order object records; the last word becomes index << 3 | the three flag bits (grt_bvh.h)
  std::vector<float4> bp(2 * (size_t)n);
  parallel_ranges(n, [&](int i0, int i1) {
    for (int i = i0; i < i1; ++i) {
      const int o = B.order[i] = B.items[i].idx;
      float4 pb = hp[2 * (size_t)o + 1];
      pb.w = __builtin_bit_cast(float, (uint32_t)o << 3 | ((uint32_t)__builtin_bit_cast(int, pb.w) & 7u));  // kind, emissive, transparent
      bp[2 * (size_t)i] = hp[2 * (size_t)o], bp[2 * (size_t)i + 1] = pb;
    }
  });
  // The 16-bit grid: the scene's bounds widened by 1/1024 on each side, 65535 cells per axis.  Boxes
  // round outward and take one more cell, which covers the FP32 rounding of the slab arithmetic in
  // grid coordinates (< 0.01 cell: coordinates <= 65535 at 2^-24 relative).
  // (lo, hi: the root box, left by the build above)
  float gmin[3], gs[3];
  for (int c = 0; c < 3; ++c) {
    const double ext = std::max((double)hi[c] - (double)lo[c], 1e-30 + 1e-6 * (std::fabs((double)hi[c]) + std::fabs((double)lo[c])));
    // a cell is never smaller than the rounding error of the slab arithmetic in world units (a small
    // scene seen from far away): the one-cell padding below must cover that error
    const double reach = std::max({std::fabs((double)lo[c]), std::fabs((double)hi[c]), (double)eye_r});
    gs[c] = (float)std::max(ext * (1.0 + 2.0 / 1024.0) / 65535.0, 1e-6 * reach);
    gmin[c] = (float)((double)lo[c] - std::max(ext / 1024.0, 4.0 * (double)gs[c]));
    d.bvh_grid[c] = gmin[c], d.bvh_grid[3 + c] = gs[c], d.bvh_grid[6 + c] = 1.0f / gs[c];
  }
  clk.lap("hierarchy: leaf records, grid");
  d.bvh_depth = B.max_depth.load();
  // Memory order of the nodes: the inner children of a node are neighbours (one 64-byte pair, even
  // slot first), followed by the pair of the first child's children, then the rest of the first
  // child's subtree, then the second child's.  A 128-byte cache line so holds a pair of siblings and
  // the children of one of them: the far child, pushed while the near one is walked, arrives with it.
  const size_t built = (size_t)B.n_nodes.load();
  std::vector<int> slot_of(built, -1), node_at;
  node_at.reserve(built + built / 2 + 2);
  {
    auto child = [&](int node, int c) { return __builtin_bit_cast(int, B.nodes[4 * (size_t)node + 2 * c].w); };
    std::vector<int> todo{0};
    slot_of[0] = 0;
    node_at.push_back(0), node_at.push_back(-1);  // the root and a pad: pairs start on even slots
    while (!todo.empty()) {
      const int nd = todo.back();
      todo.pop_back();
      const int c0 = child(nd, 0), c1 = child(nd, 1);
      if (c0 < 0 && c1 < 0) continue;
      for (int c : {c0, c1}) {
        if (c >= 0) slot_of[c] = (int)node_at.size();
        node_at.push_back(c >= 0 ? c : -1);  // a leaf child leaves a pad
      }
      if (c1 >= 0) todo.push_back(c1);
      if (c0 >= 0) todo.push_back(c0);  // walked first
    }
  }
  clk.lap("hierarchy: node order");
#ifdef GRT_COUNT_DEPTH
  {
    auto child = [&](int node, int c) { return __builtin_bit_cast(int, B.nodes[4 * (size_t)node + 2 * c].w); };
    std::vector<uint8_t> sd(node_at.size(), 0);
    std::vector<std::pair<int, int>> todo{{0, 0}};
    while (!todo.empty()) {
      const auto [nd, dp] = todo.back();
      todo.pop_back();
      sd[slot_of[nd]] = (uint8_t)dp;
      for (int c : {child(nd, 0), child(nd, 1)})
        if (c >= 0) todo.push_back({c, dp + 1});
    }
    GRT_HIP(hipMalloc(&d.bvh_slot_depth, sd.size()));
    GRT_HIP(hipMemcpy(d.bvh_slot_depth, sd.data(), sd.size(), hipMemcpyHostToDevice));
  }
#endif
  const size_t n_nodes = node_at.size();
  std::vector<uint4> qn(2 * n_nodes, make_uint4(0, 0, 0, (uint32_t)~0));
  std::atomic<bool> grid_ok{true};
  parallel_ranges((int)(2 * n_nodes), [&](int k0, int k1) {
  for (size_t k = (size_t)k0; k < (size_t)k1; ++k) {
    if (node_at[k / 2] < 0) continue;  // pad: two empty leaves, never referenced
    const size_t src = 2 * (size_t)node_at[k / 2] + (k & 1);
    float4 l = B.nodes[2 * src];
    const float4 h = B.nodes[2 * src + 1];
    if (__builtin_bit_cast(int, l.w) >= 0) l.w = __builtin_bit_cast(float, slot_of[__builtin_bit_cast(int, l.w)]);
    const float fl[3] = {l.x, l.y, l.z}, fh[3] = {h.x, h.y, h.z};
    uint32_t ql[3], qh[3];
    for (int c = 0; c < 3; ++c) {
      if (fl[c] > fh[c]) {  // the empty second child of a single-leaf tree (a leaf of 0 objects)
        ql[c] = 0, qh[c] = 0;
        continue;
      }
      const double a = std::floor(((double)fl[c] - (double)gmin[c]) / (double)gs[c]) - 1.0;
      const double b = std::ceil(((double)fh[c] - (double)gmin[c]) / (double)gs[c]) + 1.0;
      if (a < 0.0 || b > 65535.0) grid_ok = false;
      ql[c] = (uint32_t)std::max(0.0, a), qh[c] = (uint32_t)std::min(65535.0, b);
    }
    qn[k] = make_uint4(ql[0] | ql[1] << 16, ql[2] | qh[0] << 16, qh[1] | qh[2] << 16, (uint32_t)__builtin_bit_cast(int, l.w));
  }
  });
  clk.lap("hierarchy: quantise");
  if (!grid_ok.load()) return grt_fail(GRT_EINVAL, "internal: hierarchy box outside its grid");
  GRT_HIP(hipMalloc(&d.bvh_nodes, qn.size() * sizeof(uint4)));
  GRT_HIP(hipMalloc(&d.bvh_prims, bp.size() * sizeof(float4)));
  GRT_HIP(hipMemcpy(d.bvh_nodes, qn.data(), qn.size() * sizeof(uint4), hipMemcpyHostToDevice));
  GRT_HIP(hipMemcpy(d.bvh_prims, bp.data(), bp.size() * sizeof(float4), hipMemcpyHostToDevice));
  clk.lap("hierarchy: upload");
  return GRT_OK;
}

// ---- tile culling ------------------------------------------------------------------------
// live[ty * tiles_x + tx] = 0 when the pyramid of camera rays through 8x8 tile (tx, ty) — all
// jitters included: u in [x0/W, (x0+8)/W], v likewise (scene.go:93-94) — misses the bounding box of
// all objects: one of the pyramid's four side planes (or the plane through the eye across the
// tile's central ray) has the whole box on its outer side.  Ray directions inside the tile are convex
// combinations of the four corner directions (Camera.Ray is affine in u and v, camera.go:37) and
// the box is convex, so no ray of the tile reaches it.  Float64, with a margin of 1e-5 (sine of
// an angle) that covers the FP32 ray generation of the kernels a hundred times over.
static std::vector<uint8_t> cull_tiles(const grt_scene* s) {
  const int W = s->d.width, H = s->d.height, tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
  std::vector<uint8_t> live((size_t)tiles_x * tiles_y, 1);
  if (getenv("GRT_NO_CULL")) return live;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (const grt_prim& p : s->prims)
    for (int c = 0; c < 3; ++c) {
      const double a = p.kind == GRT_KIND_AABB ? std::min(p.a[c], p.b[c]) : p.a[c] - std::fabs(p.radius);
      const double b = p.kind == GRT_KIND_AABB ? std::max(p.a[c], p.b[c]) : p.a[c] + std::fabs(p.radius);
      lo[c] = std::min(lo[c], a), hi[c] = std::max(hi[c], b);
    }
  if (s->prims.empty()) {
    std::fill(live.begin(), live.end(), 0);
    return live;
  }
  for (int c = 0; c < 3; ++c)
    if (!std::isfinite(lo[c]) || !std::isfinite(hi[c])) return live;  // no culling with non-finite geometry
  const grt_camera& cam = s->d.cam;
  double rel[8][3];  // box corners relative to the eye
  for (int k = 0; k < 8; ++k)
    for (int c = 0; c < 3; ++c) rel[k][c] = ((k >> c) & 1 ? hi[c] : lo[c]) - cam.origin[c];
  auto dir = [&](double u, double v, double* d) {
    for (int c = 0; c < 3; ++c) d[c] = cam.lower_left[c] + cam.horizontal[c] * u + cam.vertical[c] * v - cam.origin[c];
  };
  auto outside = [&](const double* m) {  // the whole box strictly on the negative side of the plane m.x = 0
    const double mlen = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    if (!(mlen > 0.0)) return false;
    for (int k = 0; k < 8; ++k) {
      const double blen = std::sqrt(rel[k][0] * rel[k][0] + rel[k][1] * rel[k][1] + rel[k][2] * rel[k][2]);
      if (!(m[0] * rel[k][0] + m[1] * rel[k][1] + m[2] * rel[k][2] < -1e-5 * mlen * blen)) return false;
    }
    return true;
  };
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx) {
      const double u0 = (double)(8 * tx) / W, u1 = (double)(8 * tx + 8) / W, v0 = (double)(8 * ty) / H, v1 = (double)(8 * ty + 8) / H;
      double d[4][3], dc[3];
      dir(u0, v0, d[0]), dir(u1, v0, d[1]), dir(u1, v1, d[2]), dir(u0, v1, d[3]);
      dir(0.5 * (u0 + u1), 0.5 * (v0 + v1), dc);
      bool dead = outside(dc);  // the scene lies behind the eye as seen along this tile
      for (int e = 0; e < 4 && !dead; ++e) {
        const double* a = d[e];
        const double* b = d[(e + 1) & 3];
        double m[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        if (m[0] * dc[0] + m[1] * dc[1] + m[2] * dc[2] < 0.0) m[0] = -m[0], m[1] = -m[1], m[2] = -m[2];  // inside = positive
        dead = outside(m);
      }
      live[(size_t)ty * tiles_x + tx] = dead ? 0 : 1;
    }
  return live;
}

// Stage the SoA copies on the current device (first use only).
static int stage(grt_scene* s, DeviceScene& out) {
  int dev = 0;
  GRT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(s->mu);
  auto it = s->dev.find(dev);
  if (it != s->dev.end()) {
    out = it->second;
    return GRT_OK;
  }
  StageClock clk;
  const size_t np = s->prims.size(), nm = s->mats.size(), nl = s->lights.size();
  // object and material tables are padded to GRT_FAST_PRIMS records (never-hit spheres / zero materials) so the register-resident kernel loads them whole
  std::vector<float4> hp(2 * std::max<size_t>(np, GRT_FAST_PRIMS)), hm(GRT_MAT_STRIDE * std::max<size_t>(nm, GRT_FAST_PRIMS)),
      hl(2 * std::max<size_t>(nl, 1));
  for (size_t i = np; i < hp.size() / 2; ++i) {  // never-hit padding: a sphere with R^2 = -1
    hp[2 * i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    hp[2 * i + 1] = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
  }
  std::vector<double> h64(8 * std::max<size_t>(np, 1));
  parallel_ranges((int)np, [&](int i0, int i1) {
    for (size_t i = (size_t)i0; i < (size_t)i1; ++i) {
      const grt_prim& p = s->prims[i];
      const grt_material& m = s->mats[p.material];
      int bits = (p.kind == GRT_KIND_AABB ? 1 : 0) | (m.emittance > 0 ? 2 : 0) | (m.transparent ? 4 : 0) | (p.material << 8);
      float w;
      memcpy(&w, &bits, 4);
      hp[2 * i] = make_float4((float)p.a[0], (float)p.a[1], (float)p.a[2], (float)p.radius);
      if (p.kind == GRT_KIND_AABB)
        hp[2 * i + 1] = make_float4((float)p.b[0], (float)p.b[1], (float)p.b[2], w);
      else
        hp[2 * i + 1] = make_float4((float)(p.radius * p.radius), 0.0f, 0.0f, w);
      double* q = &h64[8 * i];
      q[0] = p.a[0], q[1] = p.a[1], q[2] = p.a[2], q[3] = p.b[0], q[4] = p.b[1], q[5] = p.b[2];
      q[6] = p.radius, q[7] = p.kind == GRT_KIND_AABB ? 1.0 : 0.0;
    }
  });
  for (size_t i = 0; i < nm; ++i) {
    const grt_material& m = s->mats[i];
    int flags = (m.transparent ? GRT_MAT_TRANSPARENT : 0) | (m.lambert ? GRT_MAT_LAMBERT : 0) | (m.normal ? GRT_MAT_NORMAL : 0);
    const int cls = m.normal ? GRT_CLS_NORMAL : m.emittance > 0 ? GRT_CLS_EMIT : m.lambert ? GRT_CLS_LAMBERT : m.reflectivity > 0 ? GRT_CLS_METAL
                    : m.transparent ? GRT_CLS_GLASS : GRT_CLS_DIFFUSE;                                       // scene.go:274-305
    const int vcls = m.reflectivity > 0 ? GRT_CLS_METAL : m.transparent ? GRT_CLS_GLASS : GRT_CLS_DIFFUSE;   // scene.go:207-228
    flags |= cls << 4 | vcls << 8;
    float w;
    memcpy(&w, &flags, 4);
    hm[GRT_MAT_STRIDE * i] = make_float4((float)m.color[0], (float)m.color[1], (float)m.color[2], (float)m.refr_index);
    hm[GRT_MAT_STRIDE * i + 1] = make_float4((float)m.reflectivity, (float)m.roughness, (float)m.emittance, w);
    // Refract's constants (vec3.go:134-159; refrRatio = 1/RefrIndex, scene.go:296-297) in float64, rounded once
    const double ratio = 1.0 / m.refr_index, flipped = 1.0 / ratio;
    const double r0a = (1.0 - ratio) / (1.0 + ratio), r0b = (1.0 - flipped) / (1.0 + flipped);
    hm[GRT_MAT_STRIDE * i + 2] = make_float4((float)ratio, (float)flipped, (float)(r0a * r0a), (float)(r0b * r0b));
  }
  for (size_t k = 0; k < nl; ++k) {
    const grt_prim& p = s->prims[s->lights[k]];
    double pos[3];
    for (int c = 0; c < 3; ++c) pos[c] = p.kind == GRT_KIND_AABB ? (p.b[c] + p.a[c]) * 0.5 : p.a[c];  // aabb.go:19, sphere.go:52
    const grt_material& m = s->mats[p.material];
    hl[2 * k] = make_float4((float)pos[0], (float)pos[1], (float)pos[2], (float)s->lights[k]);
    hl[2 * k + 1] = make_float4((float)m.color[0], (float)m.color[1], (float)m.color[2], __builtin_bit_cast(float, (int)s->lights[k]));
  }
  clk.lap("records (FP32 + FP64)");
  DeviceScene d;
  auto upload = [&]() -> int {
    GRT_HIP(hipMalloc(&d.prims, hp.size() * sizeof(float4)));
    GRT_HIP(hipMalloc(&d.mats, hm.size() * sizeof(float4)));
    GRT_HIP(hipMalloc(&d.lights, hl.size() * sizeof(float4)));
    GRT_HIP(hipMalloc(&d.prims64, h64.size() * sizeof(double)));
    GRT_HIP(hipMemcpy(d.prims, hp.data(), hp.size() * sizeof(float4), hipMemcpyHostToDevice));
    GRT_HIP(hipMemcpy(d.mats, hm.data(), hm.size() * sizeof(float4), hipMemcpyHostToDevice));
    GRT_HIP(hipMemcpy(d.lights, hl.data(), hl.size() * sizeof(float4), hipMemcpyHostToDevice));
    GRT_HIP(hipMemcpy(d.prims64, h64.data(), h64.size() * sizeof(double), hipMemcpyHostToDevice));
    clk.lap("upload");
    if (s->use_bvh) {
      int rc = stage_bvh(s, hp, d);
      if (rc) return rc;
    }
    clk.t = StageClock::now();
    const std::vector<uint8_t> live = cull_tiles(s);
    GRT_HIP(hipMalloc(&d.tile_live, live.size()));
    GRT_HIP(hipMemcpy(d.tile_live, live.data(), live.size(), hipMemcpyHostToDevice));
    clk.lap("tile culling");
    return GRT_OK;
  };
  int rc = upload();
  if (!rc) {  // set-up for grt_render_accum_noalloc_dev: this device's address of the library scratch
    void* unused = nullptr;
    rc = scratch_of_current_device(&unused);
  }
  if (rc) {  // nothing of a scene that could not be staged stays on the device
    free_device_scene(d);
    return rc;
  }
  s->dev[dev] = d;
  out = d;
  return GRT_OK;
}

int grt_scene_kernel_view(grt_scene* s, uint64_t seed, KScene* ks) {
  DeviceScene d;
  int rc = stage(s, d);
  if (rc) return rc;
  const grt_camera& c = s->d.cam;
  KScene k;
  k.cam.ox = (float)c.origin[0], k.cam.oy = (float)c.origin[1], k.cam.oz = (float)c.origin[2];
  k.cam.lx = (float)(c.lower_left[0] - c.origin[0]);
  k.cam.ly = (float)(c.lower_left[1] - c.origin[1]);
  k.cam.lz = (float)(c.lower_left[2] - c.origin[2]);
  k.cam.hx = (float)c.horizontal[0], k.cam.hy = (float)c.horizontal[1], k.cam.hz = (float)c.horizontal[2];
  k.cam.vx = (float)c.vertical[0], k.cam.vy = (float)c.vertical[1], k.cam.vz = (float)c.vertical[2];
  k.W = s->d.width, k.H = s->d.height, k.max_depth = s->d.max_depth, k.n_prims = (int)s->prims.size();
  k.bias = (float)s->d.bias;
  k.seed32 = grt_rng_seed32(seed);
  k.n_lights = (int)s->lights.size();
  k.box_mask = 0;
  for (size_t i = 0; i < s->prims.size() && i < 32; ++i)
    if (s->prims[i].kind == GRT_KIND_AABB) k.box_mask |= 1u << i;
  k.prims = d.prims, k.mats = d.mats, k.lights = d.lights;
  k.bvh_nodes = d.bvh_nodes, k.bvh_prims = d.bvh_prims, k.bvh_depth = d.bvh_depth;
#ifdef GRT_COUNT_DEPTH
  k.bvh_slot_depth = d.bvh_slot_depth;
#endif
  k.n_mats = (int)s->mats.size();
  k.bvh_grid.gmin = {d.bvh_grid[0], d.bvh_grid[1], d.bvh_grid[2]};
  k.bvh_grid.gs = {d.bvh_grid[3], d.bvh_grid[4], d.bvh_grid[5]};
  k.bvh_grid.grs = {d.bvh_grid[6], d.bvh_grid[7], d.bvh_grid[8]};
  k.tile_live = d.tile_live;
  *ks = k;
  return GRT_OK;
}

static int kernel_view64(grt_scene* s, KScene64* k64) {
  DeviceScene d;
  int rc = stage(s, d);
  if (rc) return rc;
  const grt_camera& c = s->d.cam;
  for (int i = 0; i < 3; ++i) {
    k64->cam[i] = c.origin[i];
    k64->cam[3 + i] = c.lower_left[i];
    k64->cam[6 + i] = c.horizontal[i];
    k64->cam[9 + i] = c.vertical[i];
  }
  k64->bias = s->d.bias;
  k64->prims = d.prims64;
  return GRT_OK;
}

// Scenes of up to GRT_FAST_PRIMS objects (the reference's own scene has 8, main.go:40-49) run
// the kernel instance that keeps the objects in registers; larger ones read them from HBM.
static bool scene_is_small(const grt_scene* s) {
  return !s->use_bvh && (int)s->prims.size() <= GRT_FAST_PRIMS && (int)s->mats.size() <= GRT_FAST_PRIMS && !getenv("GRT_FORCE_GENERIC");
}

// ---- work decomposition ----------------------------------------------------------------
GRT_API int32_t grt_shard_rows(int32_t height, int32_t shard, int32_t n_shards) {
  if (height <= 0 || n_shards <= 0 || shard < 0 || shard >= n_shards) return 0;
  const int bands = (height + GRT_BAND_ROWS - 1) / GRT_BAND_ROWS;
  int rows = 0;
  for (int b = shard; b < bands; b += n_shards) rows += std::min(GRT_BAND_ROWS, height - b * GRT_BAND_ROWS);
  return rows;
}

static int default_chunk_spp(int n_tiles, int samples, bool hierarchy) {
  // Enough (tile, chunk) wave-items that the tail of the grid is short next to its body — ~32 items per wave slot
  // (256 CUs x 32 waves; measured best at 1920x1080x1000: 8 chunks of 125 spp, 44.3 ms vs 48.8 ms unchunked) — but
  // chunks long enough that the wave's wait for its slowest pixel stays small next to the chunk (the end-of-item
  // drain): at least 64 samples for the small-scene kernel (a shard of an 8-way split: 4.57 ms with
  // 64-spp chunks, 4.72 with 16), 8 for the hierarchy kernel (its walks vary more than its path lengths).
  const int64_t want_items = 32 * 8192;
  int64_t n_chunks = std::max<int64_t>(1, want_items / std::max(1, n_tiles));
  int chunk = (int)((samples + n_chunks - 1) / n_chunks);
  chunk = std::max(hierarchy ? 8 : 64, chunk);
  // the schedule table holds GRT_MAX_CHUNKS chunks, ~8 of them for the shrinking tail
  chunk = std::max(chunk, (samples + (GRT_MAX_CHUNKS - 9)) / (GRT_MAX_CHUNKS - 8));
  return std::min(samples, chunk);
}

// chunk_start[] for `samples` samples.  Explicit size: equal chunks (frames are then bit-identical however the work
// is sharded).  Automatic: full chunks of `chunk`, and the last one to two chunks' worth of samples in halves — the
// launch hands out items chunk by chunk, so its last wave-items are the short ones and the grid's tail (waves idle
// while the last items finish) shrinks from half a full item to half a small one.
static int chunk_schedule(int samples, int chunk, bool shrinking_tail, int* start) {
  int n = 0, s = 0;
  start[0] = 0;
  if (!shrinking_tail) {
    while (s < samples) start[++n] = (s = std::min(samples, s + chunk));
    return n;
  }
  while (samples - s > 2 * chunk) start[++n] = (s += chunk);
  int r = samples - s;
  while (r > 16 && n < GRT_MAX_CHUNKS - 1) {
    const int take = std::max(8, r / 2);
    start[++n] = (s += take);
    r -= take;
  }
  if (r > 0) start[++n] = (s += r);
  return n;
}

static int make_job(const grt_scene* s, const grt_opts* o, int shard, int n_shards, int samples, RenderJob* j) {
  if (n_shards <= 0 || shard < 0 || shard >= n_shards) return grt_fail(GRT_EINVAL, "bad shard %d of %d", shard, n_shards);
  if (samples <= 0) return grt_fail(GRT_EINVAL, "samples must be positive (got %d)", samples);
  const int bands = (s->d.height + GRT_BAND_ROWS - 1) / GRT_BAND_ROWS;
  const int local_bands = shard < bands ? (bands - shard + n_shards - 1) / n_shards : 0;
  j->shard = shard, j->n_shards = n_shards;
  j->rows_local = grt_shard_rows(s->d.height, shard, n_shards);
  j->tiles_x = (s->d.width + 7) / 8;
  j->n_tiles = j->tiles_x * local_bands;
  const bool explicit_chunk = o && o->chunk_spp > 0;
  j->chunk_spp = explicit_chunk ? std::min(o->chunk_spp, samples) : default_chunk_spp(j->n_tiles, samples, s->use_bvh);
  if ((samples + j->chunk_spp - 1) / j->chunk_spp > GRT_MAX_CHUNKS)
    return grt_fail(GRT_ELIMIT, "chunk_spp %d cuts %d samples into more than %d chunks", j->chunk_spp, samples, GRT_MAX_CHUNKS);
  j->n_chunks = chunk_schedule(samples, j->chunk_spp, !explicit_chunk && samples > j->chunk_spp && !getenv("GRT_UNIFORM_CHUNKS"), j->chunk_start);
  j->spp = samples;
  j->s_first = 0;
  j->pm.pos = j->pm.nrm = j->pm.w = nullptr;
  j->pm.view = 0;
  j->pm_batch = 0;
  const char* e = getenv("GRT_BVH_SHADE_MIN");  // A/B knob; 64 = every level in lockstep
  // 32 with the wave queue (C5: 0.252 s at 10, 0.245 at 16, 0.240 at 32, 0.254 at 48); round 1's lane = pixel loops ran best at 10
  j->shade_min = e ? std::max(1, std::min(64, atoi(e))) : ((o && (o->flags & 15) == 8) ? 10 : 32);
  e = getenv("GRT_BVH_INNER_WEIGHT");
  j->inner_weight = e ? std::max(1, std::min(1024, atoi(e))) : 1;
  e = getenv("GRT_BVH_INNER_AGAIN");
  j->inner_again = e ? std::max(1, std::min(65, atoi(e))) : 1;
  e = getenv("GRT_BVH_INNER_RUN");
  j->inner_run = e ? std::max(1, std::min(64, atoi(e))) : 4;
  j->no_octants = (o && (o->flags & 15) == 7) ? 1 : 0;  // variant 7: the min/max slab test (A/B against the octant tables)
  // k_pathtrace_bvh sums a pixel's radiance in 64-bit fixed point: 2^e units per unit of radiance, e such that a chunk of
  // 2^20 samples of the brightest thing a path can return stays below 2^62 (a function of the scene alone, so frames
  // of different sample counts or shardings are made of the same numbers).
  int longest = 0;
  for (int c = 0; c < j->n_chunks; ++c) longest = std::max(longest, j->chunk_start[c + 1] - j->chunk_start[c]);
  if (s->use_bvh && longest > (1 << 20))
    return grt_fail(GRT_ELIMIT, "a chunk of %d samples is beyond the 2^20 the hierarchy kernel's fixed-point sums are sized for", longest);
  if (s->use_bvh && s->radiance_log2 == 1024)
    return grt_fail(GRT_ELIMIT, "the scene's brightest source x largest throughput is not finite: no fixed-point scale for the hierarchy kernel");
  // a black scene (INT32_MIN) takes any scale; the exponent stays inside double's range (|e2| <= 900: radiance down to 1e-250)
  const int e2 = s->radiance_log2 == INT32_MIN ? 0 : std::max(-900, std::min(900, 62 - 20 - s->radiance_log2));
  j->acc_scale = std::ldexp(1.0, e2), j->acc_inv = std::ldexp(1.0, -e2);
  return GRT_OK;
}

GRT_API uint64_t grt_accum_workspace_bytes(const grt_scene* s, const grt_opts* o, int32_t shard, int32_t n_shards,
                                           int32_t samples) {
  RenderJob j;
  if (!s || make_job(s, o, shard, n_shards, samples, &j)) return 0;
  if (j.n_chunks <= 1) return 0;
  return (uint64_t)j.n_chunks * (uint64_t)j.rows_local * (uint64_t)s->d.width * 3u * sizeof(float);
}

int grt_render_dev_internal(grt_scene* s, const grt_opts* o, int shard, int n_shards, int sample_first, int samples,
                            float* d_rgb_sum, void* d_workspace, uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream,
                            float* pm_pos, float* pm_nrm, float* pm_w, int pm_batch, int pm_view) {
  if (!s || !d_rgb_sum) return grt_fail(GRT_EINVAL, "grt_render_accum_dev: NULL scene or output");
  if (pm_view && !pm_pos) return grt_fail(GRT_EINVAL, "photon-map view without request arrays");
  if (sample_first < 0) return grt_fail(GRT_EINVAL, "first sample must not be negative (got %d)", sample_first);
  RenderJob j;
  int rc = make_job(s, o, shard, n_shards, samples, &j);
  if (rc) return rc;
  j.s_first = sample_first;
  j.pm.pos = pm_pos, j.pm.nrm = pm_nrm, j.pm.w = pm_w, j.pm_batch = pm_batch;
  j.pm.view = pm_view;
  if (pm_pos && pm_batch < samples) return grt_fail(GRT_EINVAL, "photon-mapped render: batch %d < samples %d", pm_batch, samples);
  if (j.rows_local == 0) return GRT_OK;
  const uint64_t need = grt_accum_workspace_bytes(s, o, shard, n_shards, samples);
  if (need > workspace_bytes || (need && !d_workspace))
    return grt_fail(GRT_EINVAL, "grt_render_accum_dev: workspace %llu bytes, need %llu", (unsigned long long)workspace_bytes,
                    (unsigned long long)need);
  KScene ks;
  rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)j.n_tiles * j.n_chunks;
  const unsigned grid = (unsigned)((items + 3) / 4);
  float* target = j.n_chunks > 1 ? (float*)d_workspace : d_rgb_sum;
  // opts->flags bits 0-3: kernel variant for A/B timing (0 = default)
  const int variant = o ? (o->flags & 15) : 0;
  unsigned long long* rcnt = (unsigned long long*)d_ray_count;
  if (pm_view && s->use_bvh)  // photon-map view (a debug mode: the lockstep state machine over the hierarchy / the linear loop)
    hipLaunchKernelGGL((k_pathtrace<-1, false, false, false, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (pm_view)
    hipLaunchKernelGGL((k_pathtrace<0, false, false, false, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (s->use_bvh && variant == 5)  // every level in lockstep through path_advance (the state machine of the small scenes)
    hipLaunchKernelGGL((k_pathtrace<-1, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (s->use_bvh && variant == 6) {  // profiling build: d_ray_count holds 3 words (rays, inner nodes visited, objects tested)
    if (!rcnt) return grt_fail(GRT_EINVAL, "render variant 6 counts into d_ray_count[0..15]: a NULL counter was passed");
    if (s->has_diffuse) return grt_fail(GRT_EINVAL, "render variant 6 (test counters) is built without the plain-diffuse branch");
#ifdef GRT_COUNT_MAP  // experiment build (tools/exp_c5_map.py): per-pixel work of the lane = pixel loops written into the frame
    hipLaunchKernelGGL((k_pathtrace_bvh<6, GRT_BVH_STACK, true, false, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
#else
    hipLaunchKernelGGL((k_pathtrace_bvh<6, GRT_BVH_STACK, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
#endif
  } else if (s->use_bvh) {
    const char* e = getenv("GRT_BVH_OCC");
    const int occ = e ? atoi(e) : 7;
    if (s->has_diffuse)
      hipLaunchKernelGGL((k_pathtrace_bvh<6, GRT_BVH_STACK, false, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
    else if (occ == 8 && ks.bvh_depth + 1 <= 20)
      hipLaunchKernelGGL((k_pathtrace_bvh<8, 20>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
    else if (variant == 8 && ks.bvh_depth + 1 <= 22)  // A/B: lane = pixel (round 1's loops) instead of the wave queue
      hipLaunchKernelGGL((k_pathtrace_bvh<7, 22, false, false, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
    else if (occ >= 7 && ks.bvh_depth + 1 <= 22)
      hipLaunchKernelGGL((k_pathtrace_bvh<7, 22>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
    else if (occ == 5)
      hipLaunchKernelGGL((k_pathtrace_bvh<5, GRT_BVH_STACK>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
    else
      hipLaunchKernelGGL((k_pathtrace_bvh<6, GRT_BVH_STACK>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  }
  else if (scene_is_small(s) && variant == 4)  // wave queue of (pixel, sample) pairs (92 VGPRs): 45.9 ms vs 44.1 ms
    hipLaunchKernelGGL((k_pathtrace<GRT_FAST_PRIMS, true, false, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (!scene_is_small(s) && variant != 1 && (int)s->prims.size() <= GRT_LDS_PRIMS && (int)s->mats.size() <= GRT_LDS_PRIMS)
    hipLaunchKernelGGL((k_pathtrace<0, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);  // objects in LDS, rolled loop
  else if (!scene_is_small(s) || variant == 1)
    hipLaunchKernelGGL((k_pathtrace<0, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (variant == 2)  // object reads hoisted out of the wave loop (115 VGPRs, 4 waves/SIMD): slower
    hipLaunchKernelGGL((k_pathtrace<GRT_FAST_PRIMS, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else if (variant == 3)  // next object record requested one object ahead (124 VGPRs): 48.2 ms vs 44.1 ms
    hipLaunchKernelGGL((k_pathtrace<GRT_FAST_PRIMS, true, true>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  else
    hipLaunchKernelGGL((k_pathtrace<GRT_FAST_PRIMS, true, false>), dim3(grid), dim3(256), 0, st, ks, j, target, rcnt);
  GRT_HIP(hipGetLastError());
  if (j.n_chunks > 1) {
    const size_t n = (size_t)j.rows_local * (size_t)s->d.width * 3;
    const unsigned g2 = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_sum_chunks, dim3(g2), dim3(256), 0, st, (const float*)d_workspace, j.n_chunks, n, d_rgb_sum);
    GRT_HIP(hipGetLastError());
  }
  return GRT_OK;
}

GRT_API int grt_render_accum_dev(grt_scene* s, const grt_opts* o, int32_t shard, int32_t n_shards, int32_t samples,
                                 float* d_rgb_sum, void* d_workspace, uint64_t workspace_bytes, uint64_t* d_ray_count,
                                 void* stream) try {
  return grt_render_dev_internal(s, o, shard, n_shards, 0, samples, d_rgb_sum, d_workspace, workspace_bytes, d_ray_count, stream,
                                 nullptr, nullptr, nullptr, 0);
} catch (...) {
  return grt_fail_exception("grt_render_accum_dev");
}

// Sample-range sharding: the whole frame, samples [sample_first, sample_first + samples) of every
// pixel.  The RNG is keyed by the absolute sample index, so the ranks' frames add up to the frame
// a single device renders (up to the order of the FP32 additions across ranks).
GRT_API int grt_render_accum_samples_dev(grt_scene* s, const grt_opts* o, int32_t sample_first, int32_t samples, float* d_rgb_sum,
                                         void* d_workspace, uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream) try {
  return grt_render_dev_internal(s, o, 0, 1, sample_first, samples, d_rgb_sum, d_workspace, workspace_bytes, d_ray_count, stream,
                                 nullptr, nullptr, nullptr, 0);
} catch (...) {
  return grt_fail_exception("grt_render_accum_samples_dev");
}

// grt_render_accum_dev for callers that cannot allocate (HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): no workspace argument, the per-chunk partial sums live in the library's `scratch`.
// Calls no hipMalloc / hipFree / allocating routine: the scene must already be staged on the current device
// (any earlier render or parity call on that device stages it), else GRT_EINVAL.  Chunking: with
// opts->chunk_spp == 0 the library's choice is coarsened until the slabs fit GRT_NOALLOC_SCRATCH_BYTES (one
// chunk needs no scratch at all); an explicit chunk_spp whose slabs do not fit is refused with GRT_ELIMIT.
// For the same chunk size the frame equals grt_render_accum_dev's bit for bit (same kernels, same order of
// the chunk sums).  Every pixel of every slab is written before k_sum_chunks reads it, so nothing of the
// previous call's contents has to be cleared.  All calls on one GPU share that GPU's array: calls that
// overlap there (two streams, two threads) race.
GRT_API int grt_render_accum_noalloc_dev(grt_scene* s, const grt_opts* o, int32_t shard, int32_t n_shards, int32_t samples,
                                         float* d_rgb_sum, uint64_t* d_ray_count, void* stream) try {
  if (!s || !d_rgb_sum) return grt_fail(GRT_EINVAL, "grt_render_accum_noalloc_dev: NULL scene or output");
  int dev = 0;
  GRT_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->dev.find(dev) == s->dev.end())
      return grt_fail(GRT_EINVAL, "grt_render_accum_noalloc_dev: the scene is not staged on device %d (staging allocates: render once "
                                  "with grt_render_accum_dev on this device first)", dev);
  }
  grt_opts opts = o ? *o : grt_opts{0, 0, 1, 0, 0};
  uint64_t need = grt_accum_workspace_bytes(s, &opts, shard, n_shards, samples);
  if (need > GRT_NOALLOC_SCRATCH_BYTES) {
    if (opts.chunk_spp > 0)
      return grt_fail(GRT_ELIMIT, "grt_render_accum_noalloc_dev: chunk_spp %d needs %llu bytes of partial sums, the library scratch holds %llu "
                                  "(raise chunk_spp, or pass 0 for a choice that fits)", opts.chunk_spp, (unsigned long long)need,
                      (unsigned long long)GRT_NOALLOC_SCRATCH_BYTES);
    const uint64_t slab = (uint64_t)grt_shard_rows(s->d.height, shard, n_shards) * (uint64_t)s->d.width * 3u * sizeof(float);
    const int fit = slab ? (int)(GRT_NOALLOC_SCRATCH_BYTES / slab) : 1;  // chunks whose slabs fit
    opts.chunk_spp = fit >= 2 ? (samples + fit - 1) / fit : samples;     // fit < 2: one chunk, written straight to d_rgb_sum
    need = grt_accum_workspace_bytes(s, &opts, shard, n_shards, samples);
    if (need > GRT_NOALLOC_SCRATCH_BYTES) return grt_fail(GRT_ELIMIT, "grt_render_accum_noalloc_dev: internal: %llu bytes do not fit", (unsigned long long)need);
  }
  void* ptr = nullptr;
  int rc = scratch_of_current_device(&ptr);  // resolved when the scene was staged: a table read
  if (rc) return rc;
  return grt_render_dev_internal(s, &opts, shard, n_shards, 0, samples, d_rgb_sum, ptr, GRT_NOALLOC_SCRATCH_BYTES, d_ray_count, stream,
                                 nullptr, nullptr, nullptr, 0);
} catch (...) {
  return grt_fail_exception("grt_render_accum_noalloc_dev");
}

// total (+)= frame + photon term of one batch (k_pm_accumulate); used by grt_photon.hip
int grt_pm_accumulate_internal(const float* frame, const float* qpos, const float* qw, const float* irr_g, const float* irr_c,
                               int batch, size_t n_pixels, float* total, int first_batch, void* stream) {
  hipLaunchKernelGGL(k_pm_accumulate, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frame, qpos, qw,
                     irr_g, irr_c, batch, n_pixels, total, first_batch);
  GRT_HIP(hipGetLastError());
  return GRT_OK;
}

GRT_API int grt_tonemap_dev(const float* d_rgb_sum, int32_t width, int32_t rows, int32_t samples, uint8_t* d_bgrx,
                            int32_t pitch, void* stream) try {
  if (!d_rgb_sum || !d_bgrx || width <= 0 || rows < 0 || samples <= 0) return grt_fail(GRT_EINVAL, "grt_tonemap_dev: bad argument");
  const int bpp = pitch / width;  // scene.go:85
  if (bpp < 3) return grt_fail(GRT_EINVAL, "grt_tonemap_dev: pitch %d gives %d bytes per pixel (<3)", pitch, bpp);
  if (rows == 0) return GRT_OK;
  const size_t n = (size_t)width * rows;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_tonemap, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_rgb_sum, width, rows,
                     1.0f / (float)samples, d_bgrx, pitch, bpp, 0, 1);
  GRT_HIP(hipGetLastError());
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_tonemap_dev");
}

// ---- host-buffer entry points ------------------------------------------------------------
struct ShardResult {
  int rc = GRT_OK;
  std::string err;
  uint64_t rays = 0;
  float ms = 0.0f;
};

// Grows a cached buffer (device or pinned host) to `need` bytes; the old contents are not kept.
template <class T>
static hipError_t grow_device(T** p, size_t* cap, size_t need) {
  if (need <= *cap) return hipSuccess;
  (void)hipFree(*p);
  *p = nullptr, *cap = 0;
  const hipError_t e = hipMalloc((void**)p, need);
  if (e == hipSuccess) *cap = need;
  return e;
}
template <class T>
static hipError_t grow_pinned(T** p, size_t* cap, size_t need) {
  if (need <= *cap) return hipSuccess;
  (void)hipHostFree(*p);
  *p = nullptr, *cap = 0;
  const hipError_t e = hipHostMalloc((void**)p, need, hipHostMallocDefault);
  if (e == hipSuccess) *cap = need;
  return e;
}

// Renders one shard on `device` into the caller's host buffers (the collector of scene.go:123-132).  Device
// buffers, pinned staging and events live in the scene's HostRenderCache: after the first frame a call allocates
// nothing.  pixels_mapped != NULL: the caller's pixel buffer is registered with HIP (render_host) and this is its
// device address — k_tonemap stores B,G,R straight into the caller's memory, rows at their frame positions, bytes
// beyond the third of a pixel never written (scene.go:71-73): no staging copy, no host scatter.
static void render_shard_host(grt_scene* s, const grt_opts* o, int device, int shard, int n_shards, int samples,
                              float* rgb_sum, uint8_t* pixels, uint8_t* pixels_mapped, int pitch, ShardResult* res) {
  auto fail = [&](int rc) {
    res->rc = rc;
    res->err = g_err;
  };
  if (hipSetDevice(device) != hipSuccess) return fail(grt_fail(GRT_ENODEV, "hipSetDevice(%d) failed", device));
  const int W = s->d.width, H = s->d.height;
  const int rows = grt_shard_rows(H, shard, n_shards);
  if (rows == 0) return;
  const size_t nfl = (size_t)rows * W * 3;
  const uint64_t ws_bytes = grt_accum_workspace_bytes(s, o, shard, n_shards, samples);
  HostRenderCache* cp;
  {
    std::lock_guard<std::mutex> lock(s->mu);
    cp = &s->rcache[device];
  }
  HostRenderCache& c = *cp;
  std::lock_guard<std::mutex> call_lock(c.mu);
  int rc = GRT_OK;
  auto H_ = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == GRT_OK) rc = grt_fail(GRT_EHIP, "%s: %s", what, hipGetErrorString(e));
    return e == hipSuccess;
  };
  const int bpp = pixels ? pitch / W : 0;
  do {
    if (!H_(grow_device(&c.d_sum, &c.sum_cap, nfl * sizeof(float)), "hipMalloc(accum)")) break;
    if (ws_bytes && !H_(grow_device(&c.d_ws, &c.ws_cap, ws_bytes), "hipMalloc(workspace)")) break;
    if (!c.d_rays && !H_(hipMalloc(&c.d_rays, 16 * sizeof(uint64_t)), "hipMalloc(ray counter)")) break;
    if (!c.e0 && (!H_(hipEventCreate(&c.e0), "hipEventCreate") || !H_(hipEventCreate(&c.e1), "hipEventCreate"))) break;
    if (!H_(hipMemsetAsync(c.d_rays, 0, 16 * sizeof(uint64_t), nullptr), "hipMemsetAsync")) break;
    H_(hipEventRecord(c.e0, nullptr), "hipEventRecord");
    rc = grt_render_accum_dev(s, o, shard, n_shards, samples, c.d_sum, c.d_ws, ws_bytes, c.d_rays, nullptr);
    if (rc) break;
    H_(hipEventRecord(c.e1, nullptr), "hipEventRecord");
    const unsigned tgrid = (unsigned)std::min<size_t>(((size_t)rows * W + 255) / 256, 8192);
    if (pixels && pixels_mapped) {
      hipLaunchKernelGGL(k_tonemap, dim3(tgrid), dim3(256), 0, nullptr, c.d_sum, W, rows, 1.0f / (float)samples, pixels_mapped, pitch, bpp,
                         shard, n_shards);
      if (!H_(hipGetLastError(), "k_tonemap")) break;
    } else if (pixels) {
      const size_t nb = (size_t)rows * W * 3;
      if (!H_(grow_device(&c.d_bgr, &c.bgr_cap, nb), "hipMalloc(bgr)")) break;
      if (!H_(grow_pinned(&c.h_bgr, &c.h_bgr_cap, nb), "hipHostMalloc(bgr)")) break;
      rc = grt_tonemap_dev(c.d_sum, W, rows, samples, c.d_bgr, W * 3, nullptr);
      if (rc) break;
      if (!H_(hipMemcpyAsync(c.h_bgr, c.d_bgr, nb, hipMemcpyDeviceToHost, nullptr), "hipMemcpy(bgr)")) break;
    }
    if (rgb_sum) {
      if (!H_(grow_pinned(&c.h_sum, &c.h_sum_cap, nfl * sizeof(float)), "hipHostMalloc(accum)")) break;
      if (!H_(hipMemcpyAsync(c.h_sum, c.d_sum, nfl * sizeof(float), hipMemcpyDeviceToHost, nullptr), "hipMemcpy(accum)")) break;
    }
    if (!H_(hipMemcpy(&res->rays, c.d_rays, sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy(rays)")) break;
    if (!H_(hipStreamSynchronize(nullptr), "hipStreamSynchronize")) break;
    H_(hipEventElapsedTime(&res->ms, c.e0, c.e1), "hipEventElapsedTime");
  } while (0);
  if (rc == GRT_OK && (rgb_sum || (pixels && !pixels_mapped))) {
    // the packed bands to their rows
    int lrow = 0;
    const int bands = (H + GRT_BAND_ROWS - 1) / GRT_BAND_ROWS;
    for (int b = shard; b < bands; b += n_shards)
      for (int y = b * GRT_BAND_ROWS; y < std::min(H, (b + 1) * GRT_BAND_ROWS); ++y, ++lrow) {
        if (rgb_sum) memcpy(rgb_sum + (size_t)y * W * 3, c.h_sum + (size_t)lrow * W * 3, (size_t)W * 3 * sizeof(float));
        if (pixels && !pixels_mapped) {
          const uint8_t* src = c.h_bgr + (size_t)lrow * W * 3;
          uint8_t* dst = pixels + (size_t)y * pitch;
          if (bpp == 3) {
            memcpy(dst, src, (size_t)W * 3);
          } else {
            for (int x = 0; x < W; ++x) {  // bytes beyond B,G,R stay untouched (scene.go:71-73)
              dst[(size_t)x * bpp + 0] = src[3 * x + 0];
              dst[(size_t)x * bpp + 1] = src[3 * x + 1];
              dst[(size_t)x * bpp + 2] = src[3 * x + 2];
            }
          }
        }
      }
  }
  if (rc) fail(rc);
}

static int render_host(grt_scene* s, const grt_opts* o, int samples, float* rgb_sum, uint8_t* pixels, int pitch) {
  grt_opts opts = o ? *o : grt_opts{0, 0, 1, 0, 0};
  const int G = std::max(1, opts.n_devices);
  const int ndev = grt_device_count();
  if (ndev <= 0) return grt_fail(GRT_ENODEV, "no HIP device visible");
  // Test seam (tests/test_gpu_render.py on a one-GPU box): all G shards on first_device, one host thread each as with G
  // devices.  They share that device's buffers, so they take turns; bands, copies and totals are those of a G-device run.
  const bool one_device = getenv("GRT_SHARDS_ON_ONE_DEVICE") != nullptr;
  auto device_of = [&](int g) { return opts.first_device + (one_device ? 0 : g); };
  if (opts.first_device < 0 || device_of(G - 1) >= ndev)
    return grt_fail(GRT_ENODEV, "devices %d..%d requested, %d visible", opts.first_device, opts.first_device + G - 1, ndev);
  if (samples <= 0) return grt_fail(GRT_EINVAL, "samples must be positive (got %d)", samples);
  int cur = 0;
  (void)hipGetDevice(&cur);
  std::vector<ShardResult> res(G);
  // Zero-copy output: the caller's pixel buffer — a Go slice (main.go:64) or SDL texture memory
  // (main.go:97) — is pinned and mapped for the length of the call and the tone-map kernels write into it.  Memory
  // that cannot be registered (or GRT_NO_ZERO_COPY=1) goes through the pinned staging buffer instead.
  std::vector<uint8_t*> mapped(G, nullptr);
  bool registered = false;
  if (pixels && !getenv("GRT_NO_ZERO_COPY")) {
    (void)hipSetDevice(opts.first_device);
    // exactly the bytes Render may touch: the last row ends after its last pixel (scene.go:128 copies bpp * W bytes per row)
    const size_t bpp_reg = (size_t)pitch / (size_t)s->d.width;
    const size_t reg_bytes = (size_t)(s->d.height - 1) * (size_t)pitch + (size_t)s->d.width * bpp_reg;
    if (hipHostRegister(pixels, reg_bytes, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
      registered = true;
      for (int g = 0; g < G && registered; ++g) {
        void* dp = nullptr;
        if (hipSetDevice(device_of(g)) != hipSuccess || hipHostGetDevicePointer(&dp, pixels, 0) != hipSuccess || !dp) {
          std::fill(mapped.begin(), mapped.end(), nullptr);
          break;
        }
        mapped[g] = (uint8_t*)dp;
      }
      if (!mapped[G - 1]) std::fill(mapped.begin(), mapped.end(), nullptr);
    } else {
      (void)hipGetLastError();  // not an error of the call: the staging path serves
    }
  }
  struct Unregister {
    uint8_t* p;
    ~Unregister() {
      if (p) (void)hipHostUnregister(p);
    }
  } unregister{registered ? pixels : nullptr};
  if (G == 1) {
    render_shard_host(s, &opts, opts.first_device, 0, 1, samples, rgb_sum, pixels, mapped[0], pitch, &res[0]);
  } else {
    // a thread that cannot start (std::system_error) must not leave joinable threads behind: their destructors
    // would call std::terminate, which no function-try-block catches
    std::vector<std::thread> th;
    th.reserve(G);
    try {
      for (int g = 0; g < G; ++g)
        th.emplace_back(render_shard_host, s, &opts, device_of(g), g, G, samples, rgb_sum, pixels, mapped[g], pitch, &res[g]);
    } catch (...) {
      for (auto& t : th) t.join();
      (void)hipSetDevice(cur);
      throw;
    }
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(cur);
  g_last_rays = 0;
  g_last_ms = 0.0;
  for (int g = 0; g < G; ++g) {
    if (res[g].rc) {
      g_err = res[g].err;
      return res[g].rc;
    }
    g_last_rays += res[g].rays;
    g_last_ms = std::max(g_last_ms, (double)res[g].ms);
  }
  return GRT_OK;
}

GRT_API int grt_render(grt_scene* s, const grt_opts* o, uint8_t* pixels, int32_t pitch, int32_t samples) try {
  if (!s || !pixels) return grt_fail(GRT_EINVAL, "grt_render: NULL scene or pixels");
  if (pitch / s->d.width < 3) return grt_fail(GRT_EINVAL, "grt_render: pitch %d gives %d bytes per pixel (<3)", pitch, pitch / s->d.width);
  return render_host(s, o, samples, nullptr, pixels, pitch);
} catch (...) {
  return grt_fail_exception("grt_render");
}

GRT_API int grt_render_accum(grt_scene* s, const grt_opts* o, float* rgb_sum, int32_t samples) try {
  if (!s || !rgb_sum) return grt_fail(GRT_EINVAL, "grt_render_accum: NULL scene or output");
  return render_host(s, o, samples, rgb_sum, nullptr, 0);
} catch (...) {
  return grt_fail_exception("grt_render_accum");
}

static int use_device(const grt_opts* o) {
  const int dev = o ? o->first_device : 0;
  const int ndev = grt_device_count();
  if (ndev <= 0) return grt_fail(GRT_ENODEV, "no HIP device visible");
  if (dev < 0 || dev >= ndev) return grt_fail(GRT_ENODEV, "device %d requested, %d visible", dev, ndev);
  GRT_HIP(hipSetDevice(dev));
  return GRT_OK;
}

GRT_API int grt_trace_samples(grt_scene* s, const grt_opts* o, const int32_t* pixel_xy, int32_t n_pixels, int32_t samples,
                              float* rgb, int32_t* n_rays) try {
  if (!s || !pixel_xy || !rgb || n_pixels < 0 || samples < 0) return grt_fail(GRT_EINVAL, "grt_trace_samples: bad argument");
  for (int i = 0; i < n_pixels; ++i)
    if (pixel_xy[2 * i] < 0 || pixel_xy[2 * i] >= s->d.width || pixel_xy[2 * i + 1] < 0 || pixel_xy[2 * i + 1] >= s->d.height)
      return grt_fail(GRT_EINVAL, "grt_trace_samples: pixel %d outside the image", i);
  const size_t n = (size_t)n_pixels * samples;
  if (n == 0) return GRT_OK;
  DeviceGuard guard;
  int rc = use_device(o);
  if (rc) return rc;
  KScene ks;
  rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  int* d_xy = nullptr;
  float* d_rgb = nullptr;
  int* d_nr = nullptr;
  rc = GRT_OK;
  do {
#define TRY(x)                                                               \
  {                                                                          \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      rc = grt_fail(GRT_EHIP, "%s: %s", #x, hipGetErrorString(e_));          \
      break;                                                                 \
    }                                                                        \
  }
    TRY(hipMalloc(&d_xy, (size_t)n_pixels * 2 * sizeof(int)));
    TRY(hipMalloc(&d_rgb, n * 3 * sizeof(float)));
    TRY(hipMalloc(&d_nr, n * sizeof(int)));
    TRY(hipMemcpy(d_xy, pixel_xy, (size_t)n_pixels * 2 * sizeof(int), hipMemcpyHostToDevice));
    if (s->use_bvh)
      hipLaunchKernelGGL(k_trace_samples<-1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, ks, d_xy, n_pixels,
                         samples, d_rgb, d_nr);
    else if (scene_is_small(s))
      hipLaunchKernelGGL(k_trace_samples<GRT_FAST_PRIMS>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, ks, d_xy,
                         n_pixels, samples, d_rgb, d_nr);
    else
      hipLaunchKernelGGL(k_trace_samples<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, ks, d_xy, n_pixels,
                         samples, d_rgb, d_nr);
    TRY(hipGetLastError());
    TRY(hipMemcpy(rgb, d_rgb, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (n_rays) TRY(hipMemcpy(n_rays, d_nr, n * sizeof(int), hipMemcpyDeviceToHost));
  } while (0);
  (void)hipFree(d_xy);
  (void)hipFree(d_rgb);
  (void)hipFree(d_nr);
  return rc;
} catch (...) {
  return grt_fail_exception("grt_trace_samples");
}

// flags bit 8..9 of opts->flags select the debug mode of the parity kernel:
// 0 = FP32 + FP64 re-test of ambiguous lanes (default), 1 = FP32 only, 2 = FP64 only.
GRT_API int grt_primary_hits(grt_scene* s, const grt_opts* o, int32_t jitter_mode, int32_t* id, float* t, float* normal_xyz,
                             uint64_t* fallback_lanes) try {
  if (!s || !id || !t || !normal_xyz) return grt_fail(GRT_EINVAL, "grt_primary_hits: NULL argument");
  if (jitter_mode != GRT_JITTER_CENTER && jitter_mode != GRT_JITTER_RNG)
    return grt_fail(GRT_EINVAL, "grt_primary_hits: unknown jitter mode %d", jitter_mode);
  DeviceGuard guard;
  int rc = use_device(o);
  if (rc) return rc;
  KScene ks;
  KScene64 k64;
  rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  rc = kernel_view64(s, &k64);
  if (rc) return rc;
  if (!s->bvh_forced) ks.bvh_nodes = nullptr;  // parity: the linear scan and its FP64 re-test whenever the scene allows it
  const size_t n = (size_t)ks.W * ks.H;
  const int force = o ? (o->flags >> 8) & 3 : 0;
  int* d_id = nullptr;
  float* d_t = nullptr;
  float* d_n = nullptr;
  unsigned long long* d_fb = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  rc = GRT_OK;
  do {
    TRY(hipMalloc(&d_id, n * sizeof(int)));
    TRY(hipMalloc(&d_t, n * sizeof(float)));
    TRY(hipMalloc(&d_n, n * 3 * sizeof(float)));
    TRY(hipMalloc(&d_fb, sizeof(unsigned long long)));
    TRY(hipMemset(d_fb, 0, sizeof(unsigned long long)));
    TRY(hipEventCreate(&e0));
    TRY(hipEventCreate(&e1));
    TRY(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_primary_hits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, ks, k64, jitter_mode, d_id, d_t,
                       d_n, d_fb, force);
    TRY(hipGetLastError());
    TRY(hipEventRecord(e1, nullptr));
    TRY(hipMemcpy(id, d_id, n * sizeof(int), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost));
    TRY(hipMemcpy(normal_xyz, d_n, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    unsigned long long fb = 0;
    TRY(hipMemcpy(&fb, d_fb, sizeof fb, hipMemcpyDeviceToHost));
    if (fallback_lanes) *fallback_lanes = fb;
    float ms = 0;
    TRY(hipEventElapsedTime(&ms, e0, e1));
    g_last_ms = ms;
    g_last_rays = n;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_id);
  (void)hipFree(d_t);
  (void)hipFree(d_n);
  (void)hipFree(d_fb);
  return rc;
} catch (...) {
  return grt_fail_exception("grt_primary_hits");
}

GRT_API int grt_primary_sweep_dev(grt_scene* s, const grt_opts* o, int32_t samples, uint32_t* d_checksum, void* stream) try {
  if (!s || !d_checksum || samples <= 0) return grt_fail(GRT_EINVAL, "grt_primary_sweep_dev: bad argument");
  KScene ks;
  int rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  const int tiles = ((ks.W + 7) / 8) * ((ks.H + 7) / 8);
  const int variant = o ? (o->flags & 15) : 0;
  // EYE_OUT of intersect_primary_rel: the eye outside every box (closed) in the float64 scene AND in the FP32 records
  bool eye_out = true;
  for (const grt_prim& p : s->prims) {
    if (p.kind != GRT_KIND_AABB) continue;
    bool out = false;
    for (int k = 0; k < 3; ++k) {
      const double lo = std::min(p.a[k], p.b[k]), hi = std::max(p.a[k], p.b[k]), e = s->d.cam.origin[k];
      out |= (e < lo && (float)e < (float)lo) || (e > hi && (float)e > (float)hi);
    }
    eye_out &= out;
  }
  const int mode = (variant == 7 ? 1 : 0) | (variant == 9 ? 2 : 0) | (eye_out ? 4 : 0);
  if (s->use_bvh)
    hipLaunchKernelGGL(k_primary_sweep<-1>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ks, samples,
                       d_checksum, 0);
  else if (scene_is_small(s) && variant == 3)
    hipLaunchKernelGGL((k_primary_sweep<GRT_FAST_PRIMS, true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       ks, samples, d_checksum, 0);
  else if (scene_is_small(s) && variant != 1)
    hipLaunchKernelGGL((k_primary_sweep<GRT_FAST_PRIMS, false>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       ks, samples, d_checksum, mode);
  else
    hipLaunchKernelGGL(k_primary_sweep<0>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ks, samples,
                       d_checksum, 0);
  GRT_HIP(hipGetLastError());
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_primary_sweep_dev");
}
