// grt_photon.hip — photon map emission, kd-tree and k-nearest-photon radiance gather (config C4).
//
//   k_photon_trace   mapPhotons' emission loop body + tracePhotons   scene.go:152-160, 354-424
//                    + Store's direction quantisation                photon_mapping.go:165-193
//   k_photon_scatter ordered compaction of the stored photons + ScalePhotonPower's scale
//                    (photon_mapping.go:197-206)
//   balance()        kdtree.New(list, true) (photon_mapping.go:207): a left-balanced kd-tree in
//                    implicit heap order (children of node i are 2i and 2i+1, no pointers)
//   k_knn_radius     NearestSet + NKeeper(k) (photon_mapping.go:131-133): squared distance of
//                    the k-th nearest photon, per query
//   k_gather         the loop of IrradianceEst (photon_mapping.go:138-161)
//
// PARITY UNPINNED for this path: the reference never runs it and the tree is
// gonum's.  The definitions used where the source leaves room are those of
// oracle/grt_oracle_photon.h (ties at the k-th distance all kept; r2 = largest accepted d^2).
//
// Emission is one thread per emitted photon, keyed by its emission index, so the stored set
// equals what the reference's sequential loop would store with the same uniforms: the map is
// the first `max` stores in emission order (flag -> exclusive scan -> scatter).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <vector>

#include "../../include/grt.h"
#include "grt_device.h"
#include "grt_internal.h"

#define GRT_API extern "C" __attribute__((visibility("default")))
#define PI_F 3.14159265358979323846f

// ---- photon record in HBM (leaf-bucket order, 0-based; 28 B per photon) --------------------
//   pos4[i]  = (x, y, z, bits)   bits: theta << 8 | phi << 16
//   pow3[3i] = power r, g, b     (already scaled by 1000/max)
//   nodes[i] = (plane, axis)     split planes, heap order, i in [1, 2^depth)
#define KD_LEAF 64
struct grt_pmap {
  int n = 0, max = 0, device = 0, depth = 0;
  uint64_t emitted = 0;
  float4* d_pos4 = nullptr;  // [n+1]
  float* d_pow3 = nullptr;   // [3(n+1)]
  float2* d_nodes = nullptr; // [2^depth]
  float* d_lut = nullptr;    // [4*256] costheta, sintheta, cosphi, sinphi (photon_mapping.go:105-111)
  std::vector<int> order;    // slot -> storage (emission) index, for grt_photon_read
};

// ---- emission ---------------------------------------------------------------------------------
// intersect over the transparent objects only (scene.tObjects, scene.go:48-50,360)
GRT_DEV bool hits_transparent(const KScene& S, const Ray32& r, bool careful) {
  float tbest = 3.402823466e38f;
  bool hit = false;
  for (int i = 0; i < S.n_prims; ++i) {
    float4 pa = S.prims[2 * i], pb = S.prims[2 * i + 1];
    int bits = __float_as_int(pb.w);
    if (!(bits & 4)) continue;
    float ht;
    if (bits & 1)
      ht = careful ? aabb_hit32<true>(pa, pb, r, S.bias, tbest) : aabb_hit32<false>(pa, pb, r, S.bias, tbest);
    else
      ht = sphere_hit32(pa, pb, r, S.bias, tbest);
    if (ht > 0.0f) {
      hit = true;
      tbest = ht;
    }
  }
  return hit;
}

// Store's quantisation (photon_mapping.go:177-191); Go's int() truncates toward zero.
GRT_DEV int quantize_dir(f3 d) {
  int theta = (int)(acosf(fminf(fmaxf(d.z, -1.0f), 1.0f)) * (256.0f / PI_F));
  theta = theta > 255 ? 255 : theta;
  int phi = (int)(atan2f(d.y, d.x) * (256.0f / (2.0f * PI_F)));
  phi = phi > 255 ? 255 : (phi < 0 ? phi + 256 : phi);
  return (theta << 8) | ((phi & 255) << 16);
}

__global__ __launch_bounds__(256) void k_photon_trace(KScene S, float4 light_pos, float3 power0, int caustics,
                                                      unsigned long long first, int n_emit, int* __restrict__ flag,
                                                      float4* __restrict__ rec_pos, float4* __restrict__ rec_pow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n_emit;
  const unsigned long long e = first + (unsigned long long)(valid ? i : 0);
  const uint32_t key = grt_rng_path_key(grt_rng_pixel_key(S.seed32, (uint32_t)e),
                                        (uint32_t)(caustics ? 1 : 0) + 2u * (uint32_t)(e >> 32));
  f3 O = mk3(light_pos.x, light_pos.y, light_pos.z);  // l.Pos(), scene.go:149
  f3 D = sample_hemi32(mk3(0.0f, -1.0f, 0.0f), grt_rng_draw(key, GRT_PDIM_EMIT_U), grt_rng_draw(key, GRT_PDIM_EMIT_V));
  f3 power = mk3(power0.x, power0.y, power0.z);
  bool alive = valid, stored = false;
  f3 sp = mk3(0, 0, 0), spow = mk3(0, 0, 0);
  int sbits = 0;
  for (int depth = 1; depth < S.max_depth; ++depth) {  // scene.go:355
    if (!__any(alive)) break;
    if (alive) {
      Ray32 r;
      ray_setup(r, O, D);
      const bool careful = __any(ray_has_zero(r));
      if (caustics && depth == 1 && !hits_transparent(S, r, careful)) alive = false;  // scene.go:359-363
      if (alive) {
        float t;
        int id = careful ? intersect32_t<true>(S, r, t) : intersect32_t<false>(S, r, t);  // scene.go:365
        if (id < 0) {
          alive = false;
        } else {
          f3 p, n;
          int m = surface32(S, id, r, t, p, n);
          float4 m0 = S.mats[2 * m], m1 = S.mats[2 * m + 1];
          int flags = __float_as_int(m1.w);
          f3 incident = r.D * rsq32(r.a);  // scene.go:371
          f3 f = mk3(m0.x, m0.y, m0.z);
          if (caustics && depth == 1 && !(flags & GRT_MAT_TRANSPARENT)) {  // scene.go:375
            alive = false;
          } else if ((flags & GRT_MAT_NORMAL) || m1.z > 0.0f || (flags & GRT_MAT_LAMBERT)) {
            alive = false;  // scene.go:391-396: Normal / light stop; Lambert drops the photon
          } else if (m1.x > 0.0f) {  // scene.go:397-402
            f3 on = dot3(n, incident) >= 0.0f ? -n : n;
            f3 refl = reflect32(incident, n);
            if (m1.y != 0.0f)
              refl = fma3(sample_hemi32(on, grt_rng_draw(key, GRT_PDIM_HIT(depth, 0)), grt_rng_draw(key, GRT_PDIM_HIT(depth, 1))),
                          m1.y, refl);
            O = p, D = refl, power = power * f;
          } else if (flags & GRT_MAT_TRANSPARENT) {  // scene.go:403-412
            f3 dir;
            if (!refract32(incident, n, rcp32(m0.w), grt_rng_draw(key, GRT_PDIM_HIT(depth, 2)), dir)) dir = reflect32(incident, n);
            O = p, D = dir;
          } else {
            const float rrp = (f.x + f.y + f.z) * (1.0f / 3.0f);  // scene.go:388
            if (grt_rng_draw(key, GRT_PDIM_HIT(depth, 3)) < rrp) {  // scene.go:414-417
              stored = true;
              sp = p;
              spow = (f * power) * (1.0f / (1.0f - rrp));
              sbits = quantize_dir(incident);
              alive = false;
            } else {  // scene.go:418-422
              O = p;
              D = sample_hemi32(n, grt_rng_draw(key, GRT_PDIM_HIT(depth, 0)), grt_rng_draw(key, GRT_PDIM_HIT(depth, 1)));
              power = (f * power) * (1.0f / rrp);
            }
          }
        }
      }
    }
  }
  if (valid) {
    flag[i] = stored ? 1 : 0;
    if (stored) {
      rec_pos[i] = make_float4(sp.x, sp.y, sp.z, __int_as_float(sbits));
      rec_pow[i] = make_float4(spow.x, spow.y, spow.z, 0.0f);
    }
  }
}

// Stores in emission order: photon i of the batch goes to slot base + scan[i] while the map has
// room (Store's cap, photon_mapping.go:166).  last_emit = emission index of the store that filled it.
__global__ __launch_bounds__(256) void k_photon_scatter(const int* __restrict__ flag, const int* __restrict__ scan, int n_emit,
                                                        int base, int max, float scale, const float4* __restrict__ rec_pos,
                                                        const float4* __restrict__ rec_pow, float4* __restrict__ out_pos,
                                                        float4* __restrict__ out_pow, int* __restrict__ last_emit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_emit || !flag[i]) return;
  const int slot = base + scan[i];
  if (slot >= max) return;
  float4 pw = rec_pow[i];
  out_pos[slot] = rec_pos[i];
  out_pow[slot] = make_float4(pw.x * scale, pw.y * scale, pw.z * scale, 0.0f);
  if (slot == max - 1) *last_emit = i;
}

// ---- bucket kd-tree (host build) ------------------------------------------------------------
// kdtree.New(list, true) (photon_mapping.go:207) builds a balanced tree; here the balanced tree is
// a complete binary tree of `depth` levels of split planes over the photons sorted into 2^depth
// leaf buckets of 33..64 photons each.  Leaf j owns the contiguous slots
// [j*n >> depth, (j+1)*n >> depth), so a wave reads a whole bucket with one coalesced load and
// neither ranges nor child pointers are stored.  nodes[i] (heap order, children 2i and 2i+1) =
// (plane, axis): every photon of the left subtree has coordinate <= plane, every photon of the
// right subtree >= plane.
struct HostPhoton {
  float p[3];
  float pw[3];
  int bits, src;
};
static inline int leaf_start(long long j, int n, int depth) { return (int)((j * (long long)n) >> depth); }

static void kd_split(const std::vector<HostPhoton>& ph, std::vector<int>& idx, int leaf_lo, int leaf_hi, size_t node, int n,
                     int depth, std::vector<float2>& nodes, int par_depth) {
  if (leaf_hi - leaf_lo <= 1) return;
  const int lo = leaf_start(leaf_lo, n, depth), hi = leaf_start(leaf_hi, n, depth);
  const int leaf_mid = (leaf_lo + leaf_hi) >> 1, mid = leaf_start(leaf_mid, n, depth);
  float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = lo; i < hi; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = ph[idx[i]].p[k];
      mn[k] = std::min(mn[k], v), mx[k] = std::max(mx[k], v);
    }
  int ax = 0;  // split axis: largest extent of the segment
  if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
  if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
  auto cmp = [&](int a, int b) {
    float x = ph[a].p[ax], y = ph[b].p[ax];
    return x < y || (x == y && a < b);
  };
  std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, cmp);
  float2 nd;
  nd.x = ph[idx[mid]].p[ax];  // smallest coordinate of the right part
  nd.y = __builtin_bit_cast(float, ax);
  nodes[node] = nd;
  if (par_depth > 0 && hi - lo > 8192) {
    auto fut = std::async(std::launch::async, [&, leaf_lo, leaf_mid, node, par_depth] {
      kd_split(ph, idx, leaf_lo, leaf_mid, 2 * node, n, depth, nodes, par_depth - 1);
    });
    kd_split(ph, idx, leaf_mid, leaf_hi, 2 * node + 1, n, depth, nodes, par_depth - 1);
    fut.get();
  } else {
    kd_split(ph, idx, leaf_lo, leaf_mid, 2 * node, n, depth, nodes, 0);
    kd_split(ph, idx, leaf_mid, leaf_hi, 2 * node + 1, n, depth, nodes, 0);
  }
}

static int upload_tree(std::vector<HostPhoton>& ph, grt_pmap* pm) {
  const int n = (int)ph.size();
  int depth = 0;
  while (((long long)KD_LEAF << depth) < n) depth++;
  std::vector<int> idx(n);
  std::vector<float2> nodes((size_t)1 << depth, make_float2(0.0f, 0.0f));
  for (int i = 0; i < n; ++i) idx[i] = i;
  if (n > 0) kd_split(ph, idx, 0, 1 << depth, 1, n, depth, nodes, 4);
  std::vector<float4> pos4((size_t)n + 1);
  std::vector<float> pow3(3 * ((size_t)n + 1), 0.0f);
  pm->order.assign((size_t)n, -1);
  for (int i = 0; i < n; ++i) {
    const HostPhoton& h = ph[idx[i]];
    pos4[i] = make_float4(h.p[0], h.p[1], h.p[2], __builtin_bit_cast(float, h.bits));
    pow3[3 * (size_t)i] = h.pw[0], pow3[3 * (size_t)i + 1] = h.pw[1], pow3[3 * (size_t)i + 2] = h.pw[2];
    pm->order[i] = h.src;
  }
  pos4[n] = make_float4(0, 0, 0, 0);
  std::vector<float> lut(4 * 256);
  for (int i = 0; i < 256; ++i) {  // photon_mapping.go:105-111
    double angle = (double)i * (1.0 / 256.0) * M_PI;
    lut[i] = (float)cos(angle), lut[256 + i] = (float)sin(angle);
    lut[512 + i] = (float)cos(2.0 * angle), lut[768 + i] = (float)sin(2.0 * angle);
  }
  pm->n = n;
  pm->depth = depth;
  GRT_HIP(hipMalloc(&pm->d_pos4, pos4.size() * sizeof(float4)));
  GRT_HIP(hipMalloc(&pm->d_pow3, pow3.size() * sizeof(float)));
  GRT_HIP(hipMalloc(&pm->d_lut, lut.size() * sizeof(float)));
  GRT_HIP(hipMalloc(&pm->d_nodes, nodes.size() * sizeof(float2)));
  GRT_HIP(hipMemcpy(pm->d_pos4, pos4.data(), pos4.size() * sizeof(float4), hipMemcpyHostToDevice));
  GRT_HIP(hipMemcpy(pm->d_pow3, pow3.data(), pow3.size() * sizeof(float), hipMemcpyHostToDevice));
  GRT_HIP(hipMemcpy(pm->d_lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
  GRT_HIP(hipMemcpy(pm->d_nodes, nodes.data(), nodes.size() * sizeof(float2), hipMemcpyHostToDevice));
  return GRT_OK;
}

static int pick_device(const grt_opts* o, int* dev) {
  int ndev = grt_device_count();
  if (ndev <= 0) return grt_fail(GRT_ENODEV, "no HIP device visible");
  *dev = o ? o->first_device : 0;
  if (*dev < 0 || *dev >= ndev) return grt_fail(GRT_ENODEV, "device %d requested, %d visible", *dev, ndev);
  GRT_HIP(hipSetDevice(*dev));
  return GRT_OK;
}

GRT_API int grt_photon_build(grt_scene* s, const grt_opts* o, int32_t max_photons, int32_t caustics, grt_pmap** out) {
  if (!s || !out || max_photons <= 0) return grt_fail(GRT_EINVAL, "grt_photon_build: bad argument");
  if (s->lights.empty()) return grt_fail(GRT_EINVAL, "grt_photon_build: the scene has no light (Emittance > 0)");
  int dev;
  int rc = pick_device(o, &dev);
  if (rc) return rc;
  KScene ks;
  rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  const int BATCH_MAX = 1 << 23;
  int *d_flag = nullptr, *d_scan = nullptr, *d_last = nullptr;
  float4 *d_rpos = nullptr, *d_rpow = nullptr, *d_opos = nullptr, *d_opow = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  std::vector<HostPhoton> ph;
  grt_pmap* pm = new grt_pmap;
  pm->max = max_photons;
  pm->device = dev;
  rc = GRT_OK;
#define PTRY(x)                                                                     \
  {                                                                                 \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      rc = grt_fail(GRT_EHIP, "%s: %s", #x, hipGetErrorString(e_));                 \
      break;                                                                        \
    }                                                                               \
  }
  do {
    PTRY(hipMalloc(&d_flag, BATCH_MAX * sizeof(int)));
    PTRY(hipMalloc(&d_scan, BATCH_MAX * sizeof(int)));
    PTRY(hipMalloc(&d_last, sizeof(int)));
    PTRY(hipMalloc(&d_rpos, (size_t)BATCH_MAX * sizeof(float4)));
    PTRY(hipMalloc(&d_rpow, (size_t)BATCH_MAX * sizeof(float4)));
    PTRY(hipMalloc(&d_opos, (size_t)max_photons * sizeof(float4)));
    PTRY(hipMalloc(&d_opow, (size_t)max_photons * sizeof(float4)));
    PTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag, d_scan, BATCH_MAX));
    PTRY(hipMalloc(&d_tmp, tmp_bytes));
    const float scale = (float)(1000.0 / (double)max_photons);  // scene.go:163-164
    const float3 power0 = caustics ? make_float3(1.0f, 1.0f, 1.0f) : make_float3(15.0f, 15.0f, 15.0f);
    int stored = 0;
    unsigned long long emitted = 0;
    bool failed = false;
    // scene.go:146-161: lights in order; a light's loop also ends when the map is full
    for (size_t li = 0; li < s->lights.size() && stored < max_photons && !failed; ++li) {
      const grt_prim& lp = s->prims[s->lights[li]];
      const grt_material& lm = s->mats[lp.material];
      const double area = lm.color[0] + lm.color[1] + lm.color[2];
      const long long target = std::min<long long>(max_photons, (long long)max_photons * (long long)(int)(s->light_area / area));
      double pos[3];
      for (int c = 0; c < 3; ++c) pos[c] = lp.kind == GRT_KIND_AABB ? (lp.b[c] + lp.a[c]) * 0.5 : lp.a[c];
      const float4 lpos = make_float4((float)pos[0], (float)pos[1], (float)pos[2], 0.0f);
      double yield = 0.0;
      int empty_batches = 0;
      while (stored < target) {
        const int remaining = (int)(target - stored);
        int batch = yield > 0.0 ? (int)std::min<double>(BATCH_MAX, remaining / yield * 1.05 + 4096) : std::min(BATCH_MAX, std::max(65536, 2 * remaining));
        batch = std::max(batch, 1024);
        const unsigned grid = (unsigned)((batch + 255) / 256);
        hipLaunchKernelGGL(k_photon_trace, dim3(grid), dim3(256), 0, nullptr, ks, lpos, power0, caustics, emitted, batch, d_flag,
                           d_rpos, d_rpow);
        if (hipGetLastError() != hipSuccess) { failed = true; break; }
        if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flag, d_scan, batch) != hipSuccess) { failed = true; break; }
        int minus1 = -1;
        if (hipMemcpy(d_last, &minus1, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { failed = true; break; }
        hipLaunchKernelGGL(k_photon_scatter, dim3(grid), dim3(256), 0, nullptr, d_flag, d_scan, batch, stored, (int)target, scale,
                           d_rpos, d_rpow, d_opos, d_opow, d_last);
        int last_scan = 0, last_flag = 0, last_emit = -1;
        if (hipMemcpy(&last_scan, d_scan + batch - 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&last_flag, d_flag + batch - 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&last_emit, d_last, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { failed = true; break; }
        const int got = last_scan + last_flag;
        if (stored + got >= target) {
          emitted += (unsigned long long)(last_emit >= 0 ? last_emit + 1 : batch);
          stored = (int)target;
        } else {
          emitted += (unsigned long long)batch;
          stored += got;
        }
        yield = std::max(1e-6, (double)got / batch);
        // A scene that can never store (e.g. Lambert walls) would spin forever in
        // the reference; give up after 64 fruitless batches.
        empty_batches = got ? 0 : empty_batches + 1;
        if (empty_batches >= 64) {
          rc = grt_fail(GRT_EINVAL, "grt_photon_build: no photon stored in %llu emissions (no DiffuseMaterial surface reachable?)",
                        emitted);
          break;
        }
      }
      if (rc) break;
    }
    if (rc) break;
    if (failed) {
      rc = grt_fail(GRT_EHIP, "grt_photon_build: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    pm->emitted = emitted;
    std::vector<float4> hp(stored), hw(stored);
    PTRY(hipMemcpy(hp.data(), d_opos, (size_t)stored * sizeof(float4), hipMemcpyDeviceToHost));
    PTRY(hipMemcpy(hw.data(), d_opow, (size_t)stored * sizeof(float4), hipMemcpyDeviceToHost));
    ph.resize(stored);
    for (int i = 0; i < stored; ++i) {
      ph[i].p[0] = hp[i].x, ph[i].p[1] = hp[i].y, ph[i].p[2] = hp[i].z;
      memcpy(&ph[i].bits, &hp[i].w, 4);
      ph[i].pw[0] = hw[i].x, ph[i].pw[1] = hw[i].y, ph[i].pw[2] = hw[i].z;
      ph[i].src = i;
    }
    rc = upload_tree(ph, pm);
  } while (0);
  (void)hipFree(d_flag), (void)hipFree(d_scan), (void)hipFree(d_last), (void)hipFree(d_rpos), (void)hipFree(d_rpow);
  (void)hipFree(d_opos), (void)hipFree(d_opow), (void)hipFree(d_tmp);
  if (rc) {
    grt_photon_free(pm);
    return rc;
  }
  *out = pm;
  return GRT_OK;
}

GRT_API int grt_photon_load(const grt_opts* o, const float* pos, const float* power, const uint8_t* theta_phi, int32_t n,
                            grt_pmap** out) {
  if (!pos || !power || !theta_phi || n < 0 || !out) return grt_fail(GRT_EINVAL, "grt_photon_load: bad argument");
  int dev;
  int rc = pick_device(o, &dev);
  if (rc) return rc;
  std::vector<HostPhoton> ph(n);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) ph[i].p[k] = pos[3 * (size_t)i + k], ph[i].pw[k] = power[3 * (size_t)i + k];
    ph[i].bits = ((int)theta_phi[2 * (size_t)i] << 8) | ((int)theta_phi[2 * (size_t)i + 1] << 16);
    ph[i].src = i;
  }
  grt_pmap* pm = new grt_pmap;
  pm->max = n;
  pm->device = dev;
  rc = upload_tree(ph, pm);
  if (rc) {
    grt_photon_free(pm);
    return rc;
  }
  *out = pm;
  return GRT_OK;
}

GRT_API int32_t grt_photon_count(const grt_pmap* pm) { return pm ? pm->n : 0; }
GRT_API uint64_t grt_photon_emitted(const grt_pmap* pm) { return pm ? pm->emitted : 0; }

GRT_API int grt_photon_read(const grt_pmap* pm, float* pos, float* power, uint8_t* theta_phi) {
  if (!pm) return grt_fail(GRT_EINVAL, "grt_photon_read: NULL map");
  GRT_HIP(hipSetDevice(pm->device));
  const size_t n = (size_t)pm->n;
  std::vector<float4> p4(n + 1);
  std::vector<float> w3(3 * (n + 1));
  GRT_HIP(hipMemcpy(p4.data(), pm->d_pos4, p4.size() * sizeof(float4), hipMemcpyDeviceToHost));
  GRT_HIP(hipMemcpy(w3.data(), pm->d_pow3, w3.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {  // back to storage (emission) order
    const size_t d = (size_t)pm->order[i];
    int bits;
    memcpy(&bits, &p4[i].w, 4);
    if (pos) pos[3 * d] = p4[i].x, pos[3 * d + 1] = p4[i].y, pos[3 * d + 2] = p4[i].z;
    if (power) power[3 * d] = w3[3 * i], power[3 * d + 1] = w3[3 * i + 1], power[3 * d + 2] = w3[3 * i + 2];
    if (theta_phi) theta_phi[2 * d] = (uint8_t)(bits >> 8), theta_phi[2 * d + 1] = (uint8_t)(bits >> 16);
  }
  return GRT_OK;
}

GRT_API void grt_photon_free(grt_pmap* pm) {
  if (!pm) return;
  if (hipSetDevice(pm->device) == hipSuccess) {
    (void)hipFree(pm->d_pos4);
    (void)hipFree(pm->d_pow3);
    (void)hipFree(pm->d_lut);
    (void)hipFree(pm->d_nodes);
  }
  delete pm;
}

// ---- k-nearest gather: one wave per query --------------------------------------------------
// IrradianceEst (photon_mapping.go:125-162).  A WAVE serves one query at a time:
//   - the walk down the split planes is wave-uniform (node index, stack and bound live in
//     SGPRs, the stack is two VGPRs indexed by lane: push = v_cndmask on lane == sp, pop = v_readlane); the top
//     GATHER_TOP_NODES planes are read from an LDS copy, deeper ones by scalar loads;
//   - a leaf bucket is ONE coalesced load: lane l tests photon start+l;
//   - photons within the current bound are appended to the wave's candidate pool in LDS
//     (ballot + mbcnt compaction).  NearestSet + NKeeper(k) (photon_mapping.go:131-133) becomes a
//     selection of the k-th smallest squared distance in the pool: 31 steps of bisection on the
//     float bits, each step one v_cmp + s_bcnt1 per pool register.  The selection runs when the
//     pool first holds k photons (the first finite bound), when the pool is full (then the pool
//     is compacted to the photons within the new bound) and once at the end;
//   - consecutive queries of a wave are neighbours when the caller passes them in image order,
//     and the k-th distance is 1-Lipschitz: (r_k(previous) + |q - previous|)^2 is a valid bound
//     from the first node on, so most queries run exactly one selection;
//   - the loop of photon_mapping.go:138-161 runs over the pool, one candidate per lane, and a
//     wave reduction produces the sums.
// Exact: every photon with d^2 <= (k-th smallest d^2) is accepted, as in the oracle.
#define GATHER_TOP_NODES 1024
#define GATHER_WAVES 4

GRT_DEV float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
GRT_DEV float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
GRT_DEV int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
GRT_DEV int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
GRT_DEV float sgprf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
GRT_DEV bool uniform_true(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

struct KdView {
  const float2* __restrict__ nodes;
  const float4* __restrict__ pos4;
  const float* __restrict__ pow3;
  const float* __restrict__ lut;
  int n, depth;
};

// k-th smallest (1-based) of the pool's squared distances, as float bits.  v[r] holds the bits
// of entry lane + 64 r (0xffffffff beyond the pool).  Squared distances are non-negative, so
// their bit patterns order like unsigned integers.
template <int RMAX>
GRT_DEV uint32_t select_kth(const uint32_t (&v)[RMAX], int k) {
  uint32_t T = 0;
#pragma unroll 1
  for (int bit = 30; bit >= 0; --bit) {
    const uint32_t cand = T | (1u << bit);
    int c = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(v[r] < cand));
    if (c < k) T = cand;
  }
  return T;
}

template <int RMAX>
__global__ __launch_bounds__(64 * GATHER_WAVES) void k_gather_knn(KdView T, const float* __restrict__ qpos,
                                                                  const float* __restrict__ qnrm, int n_q, int q_per_wave,
                                                                  float radius, int k, float* __restrict__ irrad,
                                                                  float* __restrict__ rk2_out) {
  constexpr int CAP = 64 * RMAX;
  __shared__ float2 top[GATHER_TOP_NODES];
  __shared__ float pool_d2[GATHER_WAVES][CAP];
  __shared__ int pool_ix[GATHER_WAVES][CAP];
  const int leaf_base = 1 << T.depth;
  const int lds_end = min(leaf_base, GATHER_TOP_NODES);
  for (int i = threadIdx.x; i < GATHER_TOP_NODES; i += blockDim.x) top[i] = i < leaf_base ? T.nodes[i] : make_float2(0.0f, 0.0f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = sgpr(threadIdx.x >> 6);
  float* pd2 = pool_d2[wave];
  int* pix = pool_ix[wave];
  const int q0 = (blockIdx.x * GATHER_WAVES + wave) * q_per_wave;
  const int q1 = min(q0 + q_per_wave, n_q);
  const float radius2 = radius * radius;
  float prev_rk = -1.0f, pqx = 0.0f, pqy = 0.0f, pqz = 0.0f;  // k-th distance of the previous query (if it had one)

  for (int qi = q0; qi < q1; ++qi) {
    const float qx = qpos[3 * (size_t)qi], qy = qpos[3 * (size_t)qi + 1], qz = qpos[3 * (size_t)qi + 2];
    float bound = radius > 0.0f ? radius2 : INFINITY;  // photons beyond the radius never count (:141)
    bool tight = false;                                // bound comes from k photons already seen
    if (prev_rk >= 0.0f) {
      const float dx = qx - pqx, dy = qy - pqy, dz = qz - pqz;
      const float r = prev_rk + sqrt32(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
      const float b = r * r * 1.0001f;
      if (b < bound) bound = b, tight = true;
    }
    int count = 0;
    uint32_t vb[RMAX];

    // refresh the bound from the pool; compact the pool when it is full
    auto reselect = [&](bool compact) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const int j = lane + 64 * r;
        vb[r] = j < count ? __float_as_uint(pd2[j]) : 0xffffffffu;
      }
      const uint32_t Tk = select_kth<RMAX>(vb, k);
      bound = __uint_as_float(Tk);
      tight = true;
      if (compact) {
        int nc = 0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const int j = lane + 64 * r;
          const bool keep = vb[r] <= Tk;
          const int ix = pix[j];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
          const int slot = nc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
          if (keep) {
            pd2[slot] = __uint_as_float(vb[r]);
            pix[slot] = ix;
          }
          nc += __builtin_popcountll(m);
        }
        count = nc;
      }
    };

    if (T.n > 0) {
      // wave-uniform walk: near child first, far child pushed with its plane distance
      int stk_node = 0;
      float stk_d2 = 0.0f;
      int sp = 0, node = 1;
      for (;;) {
        // descend: the index only grows, so the LDS-resident levels come first
#define GRT_KD_STEP(ND)                                                                   \
  {                                                                                       \
    const float2 nd = (ND);                                                               \
    const int ax = sgpr(__float_as_int(nd.y));                                            \
    const float qa = ax == 0 ? qx : (ax == 1 ? qy : qz);                                  \
    const float delta = qa - nd.x;                                                        \
    const int right = sgpr(__float_as_int(delta)) >= 0 ? 1 : 0;                           \
    const float dd = delta * delta;                                                       \
    if (uniform_true(dd <= bound)) {                                                      \
      const bool top_lane = lane == sp; /* lane sp of the two stack registers */          \
      stk_node = top_lane ? 2 * node + (right ^ 1) : stk_node;                            \
      stk_d2 = top_lane ? dd : stk_d2;                                                    \
      sp++;                                                                               \
    }                                                                                     \
    node = 2 * node + right;                                                              \
  }
        while (node < lds_end) GRT_KD_STEP(top[node]);
        while (node < leaf_base) GRT_KD_STEP(T.nodes[node]);
#undef GRT_KD_STEP
        {  // leaf bucket: one photon per lane
          const int leaf = node - leaf_base;
          const int a = (int)(((long long)leaf * (long long)T.n) >> T.depth);
          const int b = (int)(((long long)(leaf + 1) * (long long)T.n) >> T.depth);
          const int i = a + lane;
          const bool valid = i < b;
          const float4 p = T.pos4[valid ? i : a];
          const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
          const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
          const bool pass = valid & (d2 <= bound);
          const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
          if (m) {
            const int slot = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (pass) {
              pd2[slot] = d2;
              pix[slot] = i;
            }
            count += __builtin_popcountll(m);
            if (count > CAP - 64) reselect(true);
            else if (!tight && count >= k) reselect(false);
          }
        }
        bool more = false;
        while (sp > 0) {  // next far subtree still within the bound
          sp--;
          const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(stk_d2), sp));
          if (uniform_true(d <= bound)) {
            node = __builtin_amdgcn_readlane(stk_node, sp);
            more = true;
            break;
          }
        }
        if (!more) break;
      }
    }

    // final selection, then the loop of photon_mapping.go:138-161 over the pool
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int j = lane + 64 * r;
      vb[r] = j < count ? __float_as_uint(pd2[j]) : 0xffffffffu;
    }
    uint32_t Tk = 0x7f800000u;  // fewer than k photons: all of them
    if (count >= k) Tk = select_kth<RMAX>(vb, k);
    const float nx = qnrm[3 * (size_t)qi], ny = qnrm[3 * (size_t)qi + 1], nz = qnrm[3 * (size_t)qi + 2];
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, r2 = 0.0f;
    int found = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int j = lane + 64 * r;
      if (64 * r >= count) break;  // wave-uniform
      const float d2 = __uint_as_float(vb[r]);
      if (j < count && vb[r] <= Tk && (radius == 0.0f || d2 < radius2)) {  // :141
        const int ix = pix[j];
        const int bits = __float_as_int(T.pos4[ix].w);
        const int th = (bits >> 8) & 255, ph = (bits >> 16) & 255;
        const float st = T.lut[256 + th];
        const float dx = st * T.lut[512 + ph], dy = st * T.lut[768 + ph], dz = T.lut[th];  // PhotonDir, :117-122
        if (dx * nx + dy * ny + dz * nz < 0.0f) {                                       // :144
          sx += T.pow3[3 * (size_t)ix], sy += T.pow3[3 * (size_t)ix + 1], sz += T.pow3[3 * (size_t)ix + 2];
          r2 = fmaxf(r2, d2);
          found++;
        }
      }
    }
    sx = wave_sum_f(sx), sy = wave_sum_f(sy), sz = wave_sum_f(sz);
    r2 = wave_max_f(r2);
    found = wave_sum_i(found);
    if (found >= 8) {  // :154-161
      const float tmp = 1.0f / (PI_F * r2);
      sx *= tmp, sy *= tmp, sz *= tmp;
    }
    if (lane == 0) {
      irrad[3 * (size_t)qi] = sx, irrad[3 * (size_t)qi + 1] = sy, irrad[3 * (size_t)qi + 2] = sz;
      if (rk2_out) rk2_out[qi] = count >= k ? __uint_as_float(Tk) : (count ? INFINITY : -1.0f);
    }
    // The bound handed to the next query must be the true k-th distance over ALL photons: it is
    // when the selection saw k photons and was not cut short by the radius.
    if (count >= k) prev_rk = sqrt32(__uint_as_float(Tk)) * 1.00001f, pqx = qx, pqy = qy, pqz = qz;
    else prev_rk = -1.0f;
  }
}

#define GRT_GATHER_MAX_K 512

GRT_API uint64_t grt_photon_gather_workspace_bytes(int32_t n) { return n > 0 ? (uint64_t)n * sizeof(float) : 0; }

// Queries served back to back by one wave (they share the carried bound).  GRT_GATHER_QPW
// overrides the default for tuning runs (tools/bench_photon.py).
static int gather_qpw() {
  const char* e = getenv("GRT_GATHER_QPW");
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : 16;
}

static int gather_launch(const grt_pmap* pm, const float* d_pos, const float* d_nrm, int n, float radius, int k, float* d_irrad,
                         float* d_rk2, hipStream_t st) {
  const int qpw = gather_qpw();
  const unsigned grid = (unsigned)((n + GATHER_WAVES * qpw - 1) / (GATHER_WAVES * qpw));
  KdView T = {pm->d_nodes, pm->d_pos4, pm->d_pow3, pm->d_lut, pm->n, pm->depth};
  if (k <= 128)
    hipLaunchKernelGGL(k_gather_knn<5>, dim3(grid), dim3(64 * GATHER_WAVES), 0, st, T, d_pos, d_nrm, n, qpw, radius, k, d_irrad, d_rk2);
  else if (k <= 256)
    hipLaunchKernelGGL(k_gather_knn<9>, dim3(grid), dim3(64 * GATHER_WAVES), 0, st, T, d_pos, d_nrm, n, qpw, radius, k, d_irrad, d_rk2);
  else
    hipLaunchKernelGGL(k_gather_knn<17>, dim3(grid), dim3(64 * GATHER_WAVES), 0, st, T, d_pos, d_nrm, n, qpw, radius, k, d_irrad, d_rk2);
  GRT_HIP(hipGetLastError());
  return GRT_OK;
}

static int gather_check(const grt_pmap* pm, const void* a, const void* b, const void* c, int n, int k) {
  if (!pm || !a || !b || !c || n < 0) return grt_fail(GRT_EINVAL, "grt_photon_gather: bad argument");
  if (k <= 0 || k > GRT_GATHER_MAX_K) return grt_fail(GRT_ELIMIT, "grt_photon_gather: k = %d outside 1..%d", k, GRT_GATHER_MAX_K);
  return GRT_OK;
}

// d_workspace: optional, grt_photon_gather_workspace_bytes(n) bytes of device memory; when given
// it receives the squared distance of each query's k-th nearest photon (+inf: fewer than k
// photons in reach, -1: none).
GRT_API int grt_photon_gather_dev(const grt_pmap* pm, const float* d_pos, const float* d_nrm, int32_t n, float radius, int32_t k,
                                  float* d_irrad, void* d_workspace, void* stream) {
  int rc = gather_check(pm, d_pos, d_nrm, d_irrad, n, k);
  if (rc) return rc;
  if (n == 0) return GRT_OK;
  return gather_launch(pm, d_pos, d_nrm, n, radius, k, d_irrad, (float*)d_workspace, (hipStream_t)stream);
}

GRT_API int grt_photon_gather(const grt_pmap* pm, const float* pos, const float* nrm, int32_t n, float radius, int32_t k,
                              float* irrad) {
  int rc = gather_check(pm, pos, nrm, irrad, n, k);
  if (rc) return rc;
  if (n == 0) return GRT_OK;
  GRT_HIP(hipSetDevice(pm->device));
  float *d_pos = nullptr, *d_nrm = nullptr, *d_out = nullptr;
  const size_t b3 = (size_t)n * 3 * sizeof(float);
  rc = GRT_OK;
  do {
    PTRY(hipMalloc(&d_pos, b3));
    PTRY(hipMalloc(&d_nrm, b3));
    PTRY(hipMalloc(&d_out, b3));
    PTRY(hipMemcpy(d_pos, pos, b3, hipMemcpyHostToDevice));
    PTRY(hipMemcpy(d_nrm, nrm, b3, hipMemcpyHostToDevice));
    rc = gather_launch(pm, d_pos, d_nrm, n, radius, k, d_out, nullptr, nullptr);
    if (rc) break;
    PTRY(hipMemcpy(irrad, d_out, b3, hipMemcpyDeviceToHost));
  } while (0);
  (void)hipFree(d_pos), (void)hipFree(d_nrm), (void)hipFree(d_out);
  return rc;
}
