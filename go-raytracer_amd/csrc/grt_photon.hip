// grt_photon.hip — photon map emission, kd-tree and k-nearest-photon radiance gather (config C4).
//
//   k_photon_trace   mapPhotons' emission loop body + tracePhotons   scene.go:152-160, 354-424
//                    + Store's direction quantisation                photon_mapping.go:165-193
//   k_photon_scatter ordered compaction of the stored photons + ScalePhotonPower's scale
//                    (photon_mapping.go:197-206)
//   build_tree_device  kdtree.New(list, true) (photon_mapping.go:207): balanced median splits into
//                    leaf buckets (one radix sort per level), collapsed to a 64-ary tree of boxes
//   k_gather_knn     NearestSet + NKeeper(k) (photon_mapping.go:131-133) and the loop of
//                    IrradianceEst (photon_mapping.go:138-161), one wave per query
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
#include <cstdio>
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
// Bucket j holds photons [j n >> depth, (j + 1) n >> depth) of the sorted order, 33..64 of them, in SLOTS 64 j .. 64 j + 63
// (round 3: until then the buckets lay back to back and every leaf step worked out its bucket's bounds with two 64-bit
// scalar products before it could ask for the photons — 9 % of the gather); the slots beyond a bucket's photons hold NaN
// positions, which no bound accepts.
//   pos3[3i] = x, y, z           what the search reads: a leaf bucket is 64 x 12 B = six whole cache lines
//   pwb[i]   = (r, g, b, bits)   what an accepted photon adds: power (already scaled by
//                                1000/max) and direction bytes, bits = theta << 8 | phi << 16
//   boxes[i], boxes[n_boxes+i]  min / max corners of the 64-ary tree's node boxes (see build_tree_device)
#define KD_LEAF 64
#define KD_FAN_BITS 6
struct grt_pmap {
  int n = 0, max = 0, device = 0, depth = 0, levels = 1, b0 = 0, n_boxes = 0;
  uint64_t emitted = 0;
  size_t slots = 0;          // 64 << depth
  float* d_pos3 = nullptr;   // [3(slots+1)]
  float4* d_pwb = nullptr;   // [slots+1]
  float4* d_boxes = nullptr; // [2 * n_boxes]
  float* d_lut = nullptr;    // [4*256] costheta, sintheta, cosphi, sinphi (photon_mapping.go:105-111)
  int* d_order = nullptr;    // [n+1] sorted position -> storage (emission) index (grt_photon_read)
  float bmin[3] = {0, 0, 0}, bmax[3] = {1, 1, 1};  // bounds of the stored photons
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
  __shared__ int bvh_stack[GRT_BVH_STACK * 256];  // hierarchy scenes: [depth][thread]
  int* stack = bvh_stack + threadIdx.x;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n_emit;
  const unsigned long long e = first + (unsigned long long)(valid ? i : 0);
  const grt_key key = grt_rng_path_key(grt_rng_pixel_key(S.seed32, (uint32_t)e),
                                        (uint32_t)(caustics ? 1 : 0) + 2u * (uint32_t)(e >> 32));
  f3 O = mk3(light_pos.x, light_pos.y, light_pos.z);  // l.Pos(), scene.go:149
  // nl = (0, -1, 0) (scene.go:150): a signed axis, the basis of SampleHemisphereNormal is made of axes (same values, fewer instructions)
  f3 D = sample_hemi32_axis(mk3(0.0f, -1.0f, 0.0f), grt_rng_draw(key, GRT_PDIM_EMIT_U), grt_rng_draw(key, GRT_PDIM_EMIT_V));
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
      if (caustics && depth == 1) {  // scene.go:359-363: the photon must start towards a transparent object
        bool ht;
        if (S.bvh_nodes)
          ht = careful ? bvh_any_hit<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, 3.402823466e38f, -1, S.n_prims, 4, 0, true, stack, 256)
                       : bvh_any_hit<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, 3.402823466e38f, -1, S.n_prims, 4, 0, true, stack, 256);
        else
          ht = hits_transparent(S, r, careful);
        if (!ht) alive = false;
      }
      if (alive) {
        float t;
        int id;  // scene.go:365
        if (S.bvh_nodes)
          id = careful ? intersect_bvh_t<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, stack, 256)
                       : intersect_bvh_t<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, stack, 256);
        else
          id = careful ? intersect32_t<true>(S, r, t) : intersect32_t<false>(S, r, t);
        if (id < 0) {
          alive = false;
        } else {
          f3 p, n;
          int m = surface32(S, id, r, t, p, n);
          float4 m0 = S.mats[GRT_MAT_STRIDE * m], m1 = S.mats[GRT_MAT_STRIDE * m + 1];
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
            if (!refract32(incident, n, S.mats[GRT_MAT_STRIDE * m + 2], grt_rng_draw(key, GRT_PDIM_HIT(depth, 2)), dir)) dir = reflect32(incident, n);
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

#define PTRY(x)                                                                     \
  {                                                                                 \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      rc = grt_fail(GRT_EHIP, "%s: %s", #x, hipGetErrorString(e_));                 \
      break;                                                                        \
    }                                                                               \
  }
// ---- bucket kd-tree (device build) ----------------------------------------------------------
// kdtree.New(list, true) (photon_mapping.go:207) builds a balanced tree; here the balanced
// tree is `depth` levels of median splits (axis of largest extent) that sort the photons into
// 2^depth leaf buckets of 33..64 photons.  Leaf j owns the contiguous slots
// [j*n >> depth, (j+1)*n >> depth), so a wave reads a whole bucket with one coalesced load and
// neither ranges nor child pointers are stored.  The split planes themselves are not kept:
// the tree is collapsed to 64-ary nodes whose children carry their bounding boxes (a box is a
// tighter bound than the last split plane, and 64 boxes are tested by one wave instruction
// stream).  Level l holds 2^(b0 + 6 l) boxes, b0 = depth - 6 (levels - 1); the last level's
// boxes are the leaves'.  boxes[i] = (min.xyz, 0), boxes[n_boxes + i] = (max.xyz, 0), levels
// concatenated, so the first GATHER_LDS_BOXES records (the top levels) are one LDS copy.
//
// The photons never leave the device.  Level l of the build: (1) bounding box of each of the
// 2^l segments (wave reduction + one atomic min/max per wave and axis, on order-preserving
// integer images of the floats), (2) split axis per segment, (3) one radix sort of all photons
// by (segment, coordinate on the segment's axis): segments are contiguous and a segment's two
// halves are the next level's segments, so after `depth` sorts every bucket holds its photons.
// n = 10^7: 18 sorts of 64-bit keys with 4-byte values (hipCUB), tens of milliseconds.
GRT_DEV uint32_t ordered_bits(float f) {  // a < b  <=>  ordered_bits(a) < ordered_bits(b)
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
GRT_DEV float ordered_float(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); }
// bucket that owns slot p: the largest j with (j*n >> depth) <= p
GRT_DEV int kd_leaf_of(int p, int n, int depth) { return (int)(((((long long)p + 1) << depth) - 1) / (long long)n); }
static inline int leaf_start(long long j, int n, int depth) { return (int)((j * (long long)n) >> depth); }

__global__ __launch_bounds__(256) void k_kd_iota(int* __restrict__ perm, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) perm[p] = p;
}

// segbox[3 seg + c] = min, segbox[3 S + 3 seg + c] = max of coordinate c (ordered images).
// A wave owns KD_BBOX_RUN consecutive groups of 64 slots.  While the run lies inside one segment (the
// top levels, where every wave of the grid would otherwise hit the same six words) the lanes keep
// their min/max over the whole run and the wave issues ONE set of atomics; a run that crosses a
// segment border is handled group by group.
#define KD_BBOX_RUN 16
__global__ __launch_bounds__(256) void k_kd_bbox(const float4* __restrict__ pos, const int* __restrict__ perm, int n, int depth,
                                                 int level, int n_seg, uint32_t* __restrict__ segbox) {
  const int lane = threadIdx.x & 63;
  const long long first = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * KD_BBOX_RUN);  // wave-uniform
  if (first >= n) return;
  const int last = (int)min((long long)n, first + 64 * KD_BBOX_RUN) - 1;
  const int sh = depth - level;
  const bool one_segment = (kd_leaf_of((int)first, n, depth) >> sh) == (kd_leaf_of(last, n, depth) >> sh);
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  auto flush = [&](int seg) {  // wave reduction, then one atomic per word
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        lo[c] = min(lo[c], (uint32_t)__shfl_xor((int)lo[c], off, 64));
        hi[c] = max(hi[c], (uint32_t)__shfl_xor((int)hi[c], off, 64));
      }
    }
    if (lane == 0)
      for (int c = 0; c < 3; ++c) {
        atomicMin(segbox + 3 * seg + c, lo[c]);
        atomicMax(segbox + 3 * n_seg + 3 * seg + c, hi[c]);
      }
  };
  for (int g = 0; g < KD_BBOX_RUN; ++g) {
    const long long p = first + 64 * g + lane;
    if (first + 64 * g >= n) break;  // wave-uniform
    const bool valid = p < n;
    uint32_t v[3] = {0, 0, 0};
    if (valid) {
      const float4 q = pos[perm[p]];
      v[0] = ordered_bits(q.x), v[1] = ordered_bits(q.y), v[2] = ordered_bits(q.z);
    }
    if (one_segment) {
      if (valid)
        for (int c = 0; c < 3; ++c) lo[c] = min(lo[c], v[c]), hi[c] = max(hi[c], v[c]);
      continue;
    }
    const int seg = kd_leaf_of(valid ? (int)p : n - 1, n, depth) >> sh;
    const int seg0 = __builtin_amdgcn_readfirstlane(seg);
    if (__all(seg == seg0)) {
      for (int c = 0; c < 3; ++c) lo[c] = valid ? v[c] : 0xffffffffu, hi[c] = valid ? v[c] : 0u;
      flush(seg0);
    } else if (valid) {
      for (int c = 0; c < 3; ++c) {
        atomicMin(segbox + 3 * seg + c, v[c]);
        atomicMax(segbox + 3 * n_seg + 3 * seg + c, v[c]);
      }
    }
  }
  if (one_segment) flush(kd_leaf_of((int)first, n, depth) >> sh);
}

__global__ __launch_bounds__(256) void k_kd_keys(const float4* __restrict__ pos, const int* __restrict__ perm, int n, int depth,
                                                 int level, int n_seg, const uint32_t* __restrict__ segbox,
                                                 unsigned long long* __restrict__ keys) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int seg = kd_leaf_of(p, n, depth) >> (depth - level);
  float ext[3];
  for (int c = 0; c < 3; ++c) ext[c] = ordered_float(segbox[3 * n_seg + 3 * seg + c]) - ordered_float(segbox[3 * seg + c]);
  int ax = 0;  // split axis: largest extent of the segment
  if (ext[1] > ext[ax]) ax = 1;
  if (ext[2] > ext[ax]) ax = 2;
  const float4 q = pos[perm[p]];
  const float v = ax == 0 ? q.x : (ax == 1 ? q.y : q.z);
  keys[p] = ((unsigned long long)(uint32_t)seg << 32) | (unsigned long long)ordered_bits(v);
}

// Photon records in bucket order + the leaf boxes: one wave per bucket.
__global__ __launch_bounds__(256) void k_kd_finalize(const float4* __restrict__ pos, const float4* __restrict__ pow,
                                                     const int* __restrict__ perm, int n, int depth, float* __restrict__ pos3,
                                                     float4* __restrict__ pwb, float4* __restrict__ box_lo,
                                                     float4* __restrict__ box_hi) {
  const int leaf = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (leaf >= (1 << depth)) return;
  const int a = (int)(((long long)leaf * n) >> depth), b = (int)(((long long)(leaf + 1) * n) >> depth);
  const int p = a + lane;
  const size_t slot = (size_t)leaf * KD_LEAF + (size_t)lane;
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};  // an empty bucket is never within any bound
  if (p < b) {
    const int src = perm[p];
    const float4 q = pos[src], w = pow[src];
    pos3[3 * slot] = q.x, pos3[3 * slot + 1] = q.y, pos3[3 * slot + 2] = q.z;
    pwb[slot] = make_float4(w.x, w.y, w.z, q.w);
    lo[0] = hi[0] = q.x, lo[1] = hi[1] = q.y, lo[2] = hi[2] = q.z;
  } else {  // a slot beyond the bucket's photons: no distance to a NaN position is <= any bound
    const float nan = __int_as_float(0x7fc00000);
    pos3[3 * slot] = nan, pos3[3 * slot + 1] = nan, pos3[3 * slot + 2] = nan;
    pwb[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], off, 64));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off, 64));
    }
  }
  if (lane == 0) {
    box_lo[leaf] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    box_hi[leaf] = make_float4(hi[0], hi[1], hi[2], 0.0f);
  }
}

// Box i of a level = union of its 64 children on the level below: one wave per box.
__global__ __launch_bounds__(256) void k_kd_upbox(const float4* __restrict__ child_lo, const float4* __restrict__ child_hi,
                                                  int n_parent, float4* __restrict__ par_lo, float4* __restrict__ par_hi) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n_parent) return;
  const float4 a = child_lo[((size_t)i << KD_FAN_BITS) + lane], b = child_hi[((size_t)i << KD_FAN_BITS) + lane];
  float lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, b.z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], off, 64));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], off, 64));
    }
  }
  if (lane == 0) {
    par_lo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    par_hi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
  }
}

// d_pos[i] = (x, y, z, bits), d_pow[i] = (r, g, b, -) for the n photons in storage order (device).
#define GRT_PMAP_MAX_SLOTS 350000000  // k_gather_knn addresses a photon's 12-byte position with a 32-bit byte offset
static int build_tree_device(const float4* d_pos, const float4* d_pow, int n, grt_pmap* pm) {
  int depth = 0;
  while (((long long)KD_LEAF << depth) < n) depth++;
  const size_t slots = (size_t)KD_LEAF << depth;  // n <= slots < 2 n
  if (slots > GRT_PMAP_MAX_SLOTS)
    return grt_fail(GRT_ELIMIT, "photon map of %d photons takes %zu bucket slots; at most %d (2^28 = 268 435 456 photons always fit)", n, slots,
                    GRT_PMAP_MAX_SLOTS);
  pm->slots = slots;
  const int levels = std::max(1, (depth + KD_FAN_BITS - 1) / KD_FAN_BITS);
  const int b0 = depth - KD_FAN_BITS * (levels - 1);
  std::vector<size_t> off(levels);
  size_t total = 0;
  for (int l = 0; l < levels; ++l) off[l] = total, total += (size_t)1 << (b0 + KD_FAN_BITS * l);
  pm->n = n, pm->depth = depth, pm->levels = levels, pm->b0 = b0, pm->n_boxes = (int)total;
  std::vector<float> lut(4 * 256);
  for (int i = 0; i < 256; ++i) {  // photon_mapping.go:105-111
    double angle = (double)i * (1.0 / 256.0) * M_PI;
    lut[i] = (float)cos(angle), lut[256 + i] = (float)sin(angle);
    lut[512 + i] = (float)cos(2.0 * angle), lut[768 + i] = (float)sin(2.0 * angle);
  }
  GRT_HIP(hipMalloc(&pm->d_pos3, 3 * (slots + 1) * sizeof(float)));
  GRT_HIP(hipMalloc(&pm->d_pwb, (slots + 1) * sizeof(float4)));
  GRT_HIP(hipMalloc(&pm->d_lut, lut.size() * sizeof(float)));
  GRT_HIP(hipMalloc(&pm->d_boxes, 2 * total * sizeof(float4)));
  GRT_HIP(hipMalloc(&pm->d_order, ((size_t)n + 1) * sizeof(int)));
  GRT_HIP(hipMemcpy(pm->d_lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));

  GRT_HIP(hipMemset(pm->d_boxes, 0, 2 * total * sizeof(float4)));
  if (n == 0) return GRT_OK;
  unsigned long long *d_k0 = nullptr, *d_k1 = nullptr;
  int* d_p1 = nullptr;
  uint32_t* d_segbox = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  int rc = GRT_OK;
  const unsigned grid = (unsigned)((n + 255) / 256);
  do {
    PTRY(hipMalloc(&d_k0, (size_t)n * sizeof(unsigned long long)));
    PTRY(hipMalloc(&d_k1, (size_t)n * sizeof(unsigned long long)));
    PTRY(hipMalloc(&d_p1, (size_t)n * sizeof(int)));
    PTRY(hipMalloc(&d_segbox, 6 * ((size_t)1 << std::max(0, depth - 1)) * sizeof(uint32_t)));
    int* perm_in = pm->d_order;
    int* perm_out = d_p1;
    PTRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k0, d_k1, perm_in, perm_out, n, 0, 64));
    PTRY(hipMalloc(&d_tmp, tmp_bytes));
    hipLaunchKernelGGL(k_kd_iota, dim3(grid), dim3(256), 0, nullptr, perm_in, n);
    for (int level = 0; level < depth; ++level) {
      const int n_seg = 1 << level;
      PTRY(hipMemsetAsync(d_segbox, 0xff, 3 * (size_t)n_seg * sizeof(uint32_t), nullptr));
      PTRY(hipMemsetAsync(d_segbox + 3 * (size_t)n_seg, 0, 3 * (size_t)n_seg * sizeof(uint32_t), nullptr));
      hipLaunchKernelGGL(k_kd_bbox, dim3((unsigned)((n + 256 * KD_BBOX_RUN - 1) / (256 * KD_BBOX_RUN))), dim3(256), 0, nullptr, d_pos, perm_in,
                         n, depth, level, n_seg, d_segbox);
      hipLaunchKernelGGL(k_kd_keys, dim3(grid), dim3(256), 0, nullptr, d_pos, perm_in, n, depth, level, n_seg, d_segbox, d_k0);
      PTRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_k0, d_k1, perm_in, perm_out, n, 0, 32 + level));
      std::swap(perm_in, perm_out);
    }
    if (rc) break;
    if (perm_in != pm->d_order) PTRY(hipMemcpyAsync(pm->d_order, perm_in, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, nullptr));
    hipLaunchKernelGGL(k_kd_finalize, dim3((unsigned)(((1 << depth) + 3) / 4)), dim3(256), 0, nullptr, d_pos, d_pow, pm->d_order, n,
                       depth, pm->d_pos3, pm->d_pwb, pm->d_boxes + off[levels - 1], pm->d_boxes + total + off[levels - 1]);
    for (int l = levels - 2; l >= 0; --l) {
      const int n_parent = 1 << (b0 + KD_FAN_BITS * l);
      hipLaunchKernelGGL(k_kd_upbox, dim3((unsigned)((n_parent + 3) / 4)), dim3(256), 0, nullptr, pm->d_boxes + off[l + 1],
                         pm->d_boxes + total + off[l + 1], n_parent, pm->d_boxes + off[l], pm->d_boxes + total + off[l]);
    }
    PTRY(hipGetLastError());
    PTRY(hipDeviceSynchronize());
    // the union of the top-level boxes
    const int n_top = 1 << b0;
    std::vector<float4> top(2 * (size_t)n_top);
    PTRY(hipMemcpy(top.data(), pm->d_boxes, (size_t)n_top * sizeof(float4), hipMemcpyDeviceToHost));
    PTRY(hipMemcpy(top.data() + n_top, pm->d_boxes + total, (size_t)n_top * sizeof(float4), hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; ++c) pm->bmin[c] = INFINITY, pm->bmax[c] = -INFINITY;
    for (int i = 0; i < n_top; ++i) {
      const float lo[3] = {top[i].x, top[i].y, top[i].z}, hi[3] = {top[n_top + i].x, top[n_top + i].y, top[n_top + i].z};
      for (int c = 0; c < 3; ++c) pm->bmin[c] = std::min(pm->bmin[c], lo[c]), pm->bmax[c] = std::max(pm->bmax[c], hi[c]);
    }
  } while (0);
  (void)hipFree(d_k0), (void)hipFree(d_k1), (void)hipFree(d_p1), (void)hipFree(d_segbox), (void)hipFree(d_tmp);
  return rc;
}

static int pick_device(const grt_opts* o, int* dev) {
  int ndev = grt_device_count();
  if (ndev <= 0) return grt_fail(GRT_ENODEV, "no HIP device visible");
  *dev = o ? o->first_device : 0;
  if (*dev < 0 || *dev >= ndev) return grt_fail(GRT_ENODEV, "device %d requested, %d visible", *dev, ndev);
  GRT_HIP(hipSetDevice(*dev));
  return GRT_OK;
}

GRT_API int grt_photon_build(grt_scene* s, const grt_opts* o, int32_t max_photons, int32_t caustics, grt_pmap** out) try {
  if (!s || !out || max_photons <= 0) return grt_fail(GRT_EINVAL, "grt_photon_build: bad argument");
  if (s->lights.empty()) return grt_fail(GRT_EINVAL, "grt_photon_build: the scene has no light (Emittance > 0)");
  DeviceGuard guard;
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
  grt_pmap* pm = new grt_pmap;
  pm->max = max_photons;
  pm->device = dev;
  rc = GRT_OK;
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
    rc = build_tree_device(d_opos, d_opow, stored, pm);
  } while (0);
  (void)hipFree(d_flag), (void)hipFree(d_scan), (void)hipFree(d_last), (void)hipFree(d_rpos), (void)hipFree(d_rpow);
  (void)hipFree(d_opos), (void)hipFree(d_opow), (void)hipFree(d_tmp);
  if (rc) {
    grt_photon_free(pm);
    return rc;
  }
  *out = pm;
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_photon_build");
}

GRT_API int grt_photon_load(const grt_opts* o, const float* pos, const float* power, const uint8_t* theta_phi, int32_t n,
                            grt_pmap** out) try {
  if (!pos || !power || !theta_phi || n < 0 || !out) return grt_fail(GRT_EINVAL, "grt_photon_load: bad argument");
  DeviceGuard guard;
  int dev;
  int rc = pick_device(o, &dev);
  if (rc) return rc;
  std::vector<float4> hp(std::max(n, 1)), hw(std::max(n, 1));
  for (int i = 0; i < n; ++i) {
    const int bits = ((int)theta_phi[2 * (size_t)i] << 8) | ((int)theta_phi[2 * (size_t)i + 1] << 16);
    hp[i] = make_float4(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], __builtin_bit_cast(float, bits));
    hw[i] = make_float4(power[3 * (size_t)i], power[3 * (size_t)i + 1], power[3 * (size_t)i + 2], 0.0f);
  }
  grt_pmap* pm = new grt_pmap;
  pm->max = n;
  pm->device = dev;
  float4 *d_p = nullptr, *d_w = nullptr;
  rc = GRT_OK;
  do {
    PTRY(hipMalloc(&d_p, hp.size() * sizeof(float4)));
    PTRY(hipMalloc(&d_w, hw.size() * sizeof(float4)));
    PTRY(hipMemcpy(d_p, hp.data(), hp.size() * sizeof(float4), hipMemcpyHostToDevice));
    PTRY(hipMemcpy(d_w, hw.data(), hw.size() * sizeof(float4), hipMemcpyHostToDevice));
    rc = build_tree_device(d_p, d_w, n, pm);
  } while (0);
  (void)hipFree(d_p), (void)hipFree(d_w);
  if (rc) {
    grt_photon_free(pm);
    return rc;
  }
  *out = pm;
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_photon_load");
}

// ---- emission slices + build from device arrays (photon-build sharding) -------
// A rank traces the emission indices [first, first + count) of one light and gets its stored
// photons, compacted in emission order; the ranks' slices are concatenated in rank order (an
// all-gather, shard.py) and every rank builds the identical tree with grt_photon_load_dev.
__global__ __launch_bounds__(256) void k_photon_compact(const int* __restrict__ flag, const int* __restrict__ scan, int n_emit,
                                                        const float4* __restrict__ rec_pos, const float4* __restrict__ rec_pow,
                                                        float4* __restrict__ out_pos, float4* __restrict__ out_pow,
                                                        int* __restrict__ n_stored) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_emit) return;
  if (flag[i]) {
    const float4 pw = rec_pow[i];
    out_pos[scan[i]] = rec_pos[i];
    out_pow[scan[i]] = make_float4(pw.x, pw.y, pw.z, __int_as_float(i));  // .w: offset of its emission inside the slice
  }
  if (i == n_emit - 1) *n_stored = scan[i] + flag[i];
}
__global__ __launch_bounds__(256) void k_photon_scale(const float4* __restrict__ in, float scale, int n, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_float4(in[i].x * scale, in[i].y * scale, in[i].z * scale, 0.0f);
}

static size_t emit_scan_bytes(int count) {
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (int*)nullptr, (int*)nullptr, count);
  return (b + 255) & ~(size_t)255;
}
GRT_API uint64_t grt_photon_emit_workspace_bytes(int32_t count) {
  if (count <= 0) return 0;
  const size_t c = ((size_t)count + 63) & ~(size_t)63;
  return 2 * c * sizeof(int) + 2 * c * sizeof(float4) + emit_scan_bytes(count);
}
GRT_API int32_t grt_scene_light_count(const grt_scene* s) { return s ? (int32_t)s->lights.size() : 0; }
// int(lightArea / area) of light k: how many times `max` photons that light's loop aims for (scene.go:152,157).
GRT_API int32_t grt_scene_light_share(const grt_scene* s, int32_t k) {
  if (!s || k < 0 || k >= (int)s->lights.size()) return 0;
  const grt_material& m = s->mats[s->prims[s->lights[k]].material];
  return (int32_t)(s->light_area / (m.color[0] + m.color[1] + m.color[2]));
}

GRT_API int grt_photon_emit_dev(grt_scene* s, const grt_opts* o, int32_t caustics, int32_t light, uint64_t first, int32_t count,
                                float* d_pos4, float* d_pow4, int32_t* d_stored, void* d_workspace, uint64_t workspace_bytes,
                                void* stream) try {
  if (!s || !d_pos4 || !d_pow4 || !d_stored || count <= 0) return grt_fail(GRT_EINVAL, "grt_photon_emit_dev: bad argument");
  if (light < 0 || light >= (int)s->lights.size()) return grt_fail(GRT_EINVAL, "grt_photon_emit_dev: light %d of %zu", light, s->lights.size());
  if (!d_workspace || workspace_bytes < grt_photon_emit_workspace_bytes(count))
    return grt_fail(GRT_EINVAL, "grt_photon_emit_dev: workspace %llu bytes, need %llu", (unsigned long long)workspace_bytes,
                    (unsigned long long)grt_photon_emit_workspace_bytes(count));
  KScene ks;
  int rc = grt_scene_kernel_view(s, o ? o->seed : 0, &ks);
  if (rc) return rc;
  const size_t c = ((size_t)count + 63) & ~(size_t)63;
  int* d_flag = (int*)d_workspace;
  int* d_scan = d_flag + c;
  float4* d_rpos = (float4*)(d_scan + c);
  float4* d_rpow = d_rpos + c;
  void* d_tmp = (void*)(d_rpow + c);
  size_t tmp_bytes = emit_scan_bytes(count);
  const grt_prim& lp = s->prims[s->lights[light]];
  double pos[3];
  for (int k = 0; k < 3; ++k) pos[k] = lp.kind == GRT_KIND_AABB ? (lp.b[k] + lp.a[k]) * 0.5 : lp.a[k];
  const float4 lpos = make_float4((float)pos[0], (float)pos[1], (float)pos[2], 0.0f);
  const float3 power0 = caustics ? make_float3(1.0f, 1.0f, 1.0f) : make_float3(15.0f, 15.0f, 15.0f);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((count + 255) / 256);
  hipLaunchKernelGGL(k_photon_trace, dim3(grid), dim3(256), 0, st, ks, lpos, power0, caustics, (unsigned long long)first, count, d_flag,
                     d_rpos, d_rpow);
  GRT_HIP(hipGetLastError());
  GRT_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flag, d_scan, count, st));
  hipLaunchKernelGGL(k_photon_compact, dim3(grid), dim3(256), 0, st, d_flag, d_scan, count, d_rpos, d_rpow, (float4*)d_pos4,
                     (float4*)d_pow4, d_stored);
  GRT_HIP(hipGetLastError());
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_photon_emit_dev");
}

GRT_API int grt_photon_load_dev(const grt_opts* o, const float* d_pos4, const float* d_pow4, int32_t n, float power_scale,
                                uint64_t emitted, grt_pmap** out) try {
  if (n < 0 || !out || (n > 0 && (!d_pos4 || !d_pow4))) return grt_fail(GRT_EINVAL, "grt_photon_load_dev: bad argument");
  // A "_dev" entry point: the arrays live on the CURRENT device and the map is built there.  opts->first_device
  // is not consulted and the current device is not switched (one process per GPU: rank r's current device is
  // GPU r, and a map built on GPU 0 from GPU r's pointers would fault).
  (void)o;
  int dev = 0;
  GRT_HIP(hipGetDevice(&dev));
  int rc = GRT_OK;
  grt_pmap* pm = new grt_pmap;
  pm->max = n;
  pm->device = dev;
  pm->emitted = emitted;
  float4* d_w = nullptr;
  rc = GRT_OK;
  do {
    PTRY(hipMalloc(&d_w, (size_t)std::max(n, 1) * sizeof(float4)));
    if (n > 0) hipLaunchKernelGGL(k_photon_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const float4*)d_pow4, power_scale, n, d_w);
    PTRY(hipGetLastError());
    rc = build_tree_device((const float4*)d_pos4, d_w, n, pm);
  } while (0);
  (void)hipFree(d_w);
  if (rc) {
    grt_photon_free(pm);
    return rc;
  }
  *out = pm;
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_photon_load_dev");
}

GRT_API int32_t grt_photon_count(const grt_pmap* pm) { return pm ? pm->n : 0; }
GRT_API int32_t grt_photon_device(const grt_pmap* pm) { return pm ? pm->device : -1; }
GRT_API uint64_t grt_photon_emitted(const grt_pmap* pm) { return pm ? pm->emitted : 0; }

GRT_API int grt_photon_read(const grt_pmap* pm, float* pos, float* power, uint8_t* theta_phi) try {
  if (!pm) return grt_fail(GRT_EINVAL, "grt_photon_read: NULL map");
  DeviceGuard guard;
  GRT_HIP(hipSetDevice(pm->device));
  const size_t n = (size_t)pm->n;
  if (n == 0) return GRT_OK;
  std::vector<float> p3(3 * pm->slots);
  std::vector<float4> w4(pm->slots);
  GRT_HIP(hipMemcpy(p3.data(), pm->d_pos3, p3.size() * sizeof(float), hipMemcpyDeviceToHost));
  GRT_HIP(hipMemcpy(w4.data(), pm->d_pwb, w4.size() * sizeof(float4), hipMemcpyDeviceToHost));
  std::vector<int> order(n + 1);
  GRT_HIP(hipMemcpy(order.data(), pm->d_order, order.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t p = 0; p < n; ++p) {  // back to storage (emission) order: sorted position p lies in slot 64 leaf + (p - leaf's first)
    const size_t d = (size_t)order[p];
    const long long leaf = ((((long long)p + 1) << pm->depth) - 1) / (long long)n;  // kd_leaf_of
    const size_t i = (size_t)leaf * KD_LEAF + (p - (size_t)leaf_start(leaf, pm->n, pm->depth));
    int bits;
    memcpy(&bits, &w4[i].w, 4);
    if (pos) pos[3 * d] = p3[3 * i], pos[3 * d + 1] = p3[3 * i + 1], pos[3 * d + 2] = p3[3 * i + 2];
    if (power) power[3 * d] = w4[i].x, power[3 * d + 1] = w4[i].y, power[3 * d + 2] = w4[i].z;
    if (theta_phi) theta_phi[2 * d] = (uint8_t)(bits >> 8), theta_phi[2 * d + 1] = (uint8_t)(bits >> 16);
  }
  return GRT_OK;
} catch (...) {
  return grt_fail_exception("grt_photon_read");
}

GRT_API void grt_photon_free(grt_pmap* pm) {
  if (!pm) return;
  DeviceGuard guard;
  if (hipSetDevice(pm->device) == hipSuccess) {
    (void)hipFree(pm->d_pos3);
    (void)hipFree(pm->d_pwb);
    (void)hipFree(pm->d_lut);
    (void)hipFree(pm->d_boxes);
    (void)hipFree(pm->d_order);
  }
  delete pm;
}

// ---- k-nearest gather: one wave per query --------------------------------------------------
// IrradianceEst (photon_mapping.go:125-162).  A WAVE serves one query at a time:
//   - the tree walk is wave-uniform: one step tests the (up to) 64 child boxes of a node, one
//     box per lane (squared distance from the query to the box against the current bound),
//     and a ballot gives the children to visit; the pending-children masks of the <= 5 levels
//     are the only stack.  The first GATHER_LDS_BOXES boxes (the top levels) are read from an
//     LDS copy, deeper ones from HBM/L2 by one coalesced load;
//   - a leaf bucket is ONE coalesced load: lane l tests photon start+l;
//   - photons within the current bound are appended to the wave's candidate pool in LDS
//     (ballot + mbcnt compaction).  NearestSet + NKeeper(k) (photon_mapping.go:131-133) becomes
//     the selection of the k-th smallest squared distance in the pool: a 256-bin histogram of
//     d^2 / max (LDS atomics), a wave prefix sum to find the bin holding the k-th, and an exact
//     ranking of that bin's few members (bisection on the float bits if the bin is crowded).
//     The selection runs when the pool first holds k photons without a finite bound, when the
//     pool is full (then it is compacted to the photons within the new bound) and at the end;
//   - consecutive queries of a wave are neighbours when the caller passes them in image order,
//     and the k-th distance is 1-Lipschitz: (r_k(previous) + |q - previous|)^2 bounds the
//     search from the first step on.  A query without a predecessor first walks nearest-child-
//     first to the nearest leaves and takes the largest distance of k photons found there;
//   - the loop of photon_mapping.go:138-161 runs over the pool, one candidate per lane, and
//     DPP wave reductions produce the sums.
// Exact: every photon with d^2 <= (k-th smallest d^2) is accepted, as in the oracle.  Pruning
// is safe in floating point: a box distance is computed with the same operation order as a
// photon distance and rounding is monotone, so box d^2 <= d^2 of every photon inside.
#define GATHER_LDS_BOXES 320
#define GATHER_BINS 256

template <int CTRL, int ROWMASK>
GRT_DEV int dpp_i(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWMASK, 0xf, false);
}
// Inclusive scans / totals over the 64 lanes: row_shr 1,2,4,8 inside each row of 16, then
// row_bcast15 into rows 1,3 and row_bcast31 into rows 2,3.  Lane 63 ends with the total.
#define GRT_WAVE_SCAN(v, OP, ID)                \
  v = OP(v, dpp_i<0x111, 0xf>(ID, v));          \
  v = OP(v, dpp_i<0x112, 0xf>(ID, v));          \
  v = OP(v, dpp_i<0x114, 0xf>(ID, v));          \
  v = OP(v, dpp_i<0x118, 0xf>(ID, v));          \
  v = OP(v, dpp_i<0x142, 0xa>(ID, v));          \
  v = OP(v, dpp_i<0x143, 0xc>(ID, v));
GRT_DEV int op_add_i(int a, int b) { return a + b; }
GRT_DEV int op_add_f(int a, int b) { return __float_as_int(__int_as_float(a) + __int_as_float(b)); }
GRT_DEV int op_max_u(int a, int b) { return (int)max((uint32_t)a, (uint32_t)b); }
GRT_DEV int op_min_u(int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); }
GRT_DEV int wave_scan_add_i(int v) {
  GRT_WAVE_SCAN(v, op_add_i, 0)
  return v;
}
GRT_DEV int wave_total_add_i(int v) {
  GRT_WAVE_SCAN(v, op_add_i, 0)
  return __builtin_amdgcn_readlane(v, 63);
}
GRT_DEV float wave_total_add_f(float f) {
  int v = __float_as_int(f);
  GRT_WAVE_SCAN(v, op_add_f, 0)
  return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
GRT_DEV uint32_t wave_total_max_u(uint32_t u) {
  int v = (int)u;
  GRT_WAVE_SCAN(v, op_max_u, 0)
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
GRT_DEV uint32_t wave_total_min_u(uint32_t u) {
  int v = (int)u;
  GRT_WAVE_SCAN(v, op_min_u, -1)
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
GRT_DEV int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
GRT_DEV int popc64(unsigned long long m) { return __builtin_popcountll(m); }
GRT_DEV int mbcnt64(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

struct P3 {
  float x, y, z;
};
struct KdView {
  const float4* __restrict__ boxes;
  const float* __restrict__ pos3;
  const float4* __restrict__ pwb;
  const float* __restrict__ lut;
  int n, depth, levels, b0, n_boxes;
};

// k-th smallest (1-based) of the values v[r] (float bits of non-negative numbers, which order
// like unsigned integers; 0xffffffff = no entry) by bisection on the bits.
template <int RMAX>
GRT_DEV uint32_t select_kth_bits(const uint32_t (&v)[RMAX], int k) {
  uint32_t T = 0;
#pragma unroll 1
  for (int bit = 30; bit >= 0; --bit) {
    const uint32_t cand = T | (1u << bit);
    int c = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) c += popc64(__builtin_amdgcn_ballot_w64(v[r] < cand));
    if (c < k) T = cand;
  }
  return T;
}

// The same through a histogram.  v[r] = entry lane + 64 r of a pool of `count` >= k entries.
template <int RMAX>
GRT_DEV uint32_t select_kth(const uint32_t (&v)[RMAX], int count, int k, int lane, int* hist) {
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) m = max(m, v[r] == 0xffffffffu ? 0u : v[r]);
  const uint32_t bmax = wave_total_max_u(m);
  if (bmax == 0u) return 0u;  // every distance is zero
  const float scale = 255.5f * rcp32(__uint_as_float(bmax));
  int key[RMAX];
  *(int4*)(hist + 4 * lane) = make_int4(0, 0, 0, 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    key[r] = -1;
    if (64 * r < count) {  // wave-uniform
      if (v[r] != 0xffffffffu) {
        key[r] = min(GATHER_BINS - 1, (int)(__uint_as_float(v[r]) * scale));  // monotone in the value
        atomicAdd(hist + key[r], 1);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int4 h = *(const int4*)(hist + 4 * lane);
  const int s = h.x + h.y + h.z + h.w;
  const int inc = wave_scan_add_i(s);
  const int L = __builtin_ctzll(__builtin_amdgcn_ballot_w64(inc >= k));  // lane whose 4 bins hold the k-th
  const int excl = inc - s, c1 = excl + h.x, c2 = c1 + h.y, c3 = c2 + h.z;
  const int bl = c1 >= k ? 0 : (c2 >= k ? 1 : (c3 >= k ? 2 : 3));
  const int below_l = bl == 0 ? excl : (bl == 1 ? c1 : (bl == 2 ? c2 : c3));
  const int b = __builtin_amdgcn_readlane(4 * lane + bl, L);
  const int j = k - __builtin_amdgcn_readlane(below_l, L);  // 1-based rank inside bin b
  uint32_t mv[RMAX];
  int members = 0;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    mv[r] = key[r] == b ? v[r] : 0xffffffffu;
    members += popc64(__builtin_amdgcn_ballot_w64(key[r] == b));
  }
  if (members == 1) {  // the usual case: the k-th is alone in its bin
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(key[r] == b);
      if (mk) return (uint32_t)__builtin_amdgcn_readlane((int)v[r], __builtin_ctzll(mk));
    }
  }
  if (members <= 8) {  // rank the members one at a time
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      unsigned long long mk = __builtin_amdgcn_ballot_w64(key[r] == b);
      while (mk) {
        const int l = __builtin_ctzll(mk);
        mk &= mk - 1;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v[r], l);
        int less = 0, leq = 0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
          less += popc64(__builtin_amdgcn_ballot_w64(mv[q] < x));
          leq += popc64(__builtin_amdgcn_ballot_w64(mv[q] <= x));
        }
        if (less < j && j <= leq) return x;
      }
    }
  }
  return select_kth_bits<RMAX>(mv, j);
}

// Work distribution: wave w of the grid serves queries [w q_per_wave, (w + 1) q_per_wave) — consecutive queries are
// neighbours in image order and hand the carried bound on.  Round 3 measured a device-wide chunk queue instead
// (resident waves take chunks from an atomic counter until none is left): 1.71 ms against 1.62 for the global map, and
// 0.82 against 0.53 ms for the caustic map, whose 8-query chunks make 113 000 atomics on one address.  The kernel does
// not lack resident waves: with 2 / 4 / 6 / 8 waves per SIMD the same launch takes 3.34 / 1.99 / 1.85 / 1.70 ms — it is
// bound by instruction issue (625 VALU + 621 SALU per query; the CU's scalar unit 58 % busy), DESIGN.md §6.
template <int RMAX, int GATHER_WAVES>
__global__ __launch_bounds__(64 * GATHER_WAVES) __attribute__((amdgpu_num_sgpr(80))) void k_gather_knn(KdView T, const float* __restrict__ qpos,
                                                                  const float* __restrict__ qnrm, int n_q, int q_per_wave,
                                                                  float radius, int k, float* __restrict__ irrad,
                                                                  float* __restrict__ rk2_out,
                                                                  unsigned long long* __restrict__ stats, float loose) {
  constexpr int CAP = 64 * RMAX;
  __shared__ float4 top_lo[GATHER_LDS_BOXES], top_hi[GATHER_LDS_BOXES];
  __shared__ float pool_d2[GATHER_WAVES][CAP];
  __shared__ int pool_ix[GATHER_WAVES][CAP];
  __shared__ __attribute__((aligned(16))) int hist_all[GATHER_WAVES][GATHER_BINS];
  __shared__ float lut[4 * 256];  // costheta, sintheta, cosphi, sinphi (photon_mapping.go:90-93)
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) lut[i] = T.lut[i];
  for (int i = threadIdx.x; i < min(GATHER_LDS_BOXES, T.n_boxes); i += blockDim.x) {
    top_lo[i] = T.boxes[i];
    top_hi[i] = T.boxes[T.n_boxes + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = sgpr(threadIdx.x >> 6);
  float* pd2 = pool_d2[wave];
  int* pix = pool_ix[wave];
  int* hist = hist_all[wave];
  int count = 0;
  const float radius2 = radius * radius;
  const int last = T.levels - 1;
  const int q0 = (blockIdx.x * GATHER_WAVES + wave) * q_per_wave;
  const int q1 = min(q0 + q_per_wave, n_q);
  float prev_rk = -1.0f, pqx = 0.0f, pqy = 0.0f, pqz = 0.0f;  // k-th distance of the previous query (if it had one)

  float nqx = 0.0f, nqy = 0.0f, nqz = 0.0f;  // the next query's position is requested one query ahead
  if (q0 < q1) nqx = qpos[3 * (size_t)q0], nqy = qpos[3 * (size_t)q0 + 1], nqz = qpos[3 * (size_t)q0 + 2];
  for (int qi = q0; qi < q1; ++qi) {
    const float qx = nqx, qy = nqy, qz = nqz;
    if (qi + 1 < q1) nqx = qpos[3 * (size_t)qi + 3], nqy = qpos[3 * (size_t)qi + 4], nqz = qpos[3 * (size_t)qi + 5];
    if (!(qx == qx)) {  // NaN position = no query in this slot (photon-mapped render): zero, and no bound to hand on
      if (lane == 0) {
        irrad[3 * (size_t)qi] = 0.0f, irrad[3 * (size_t)qi + 1] = 0.0f, irrad[3 * (size_t)qi + 2] = 0.0f;
        if (rk2_out) rk2_out[qi] = -1.0f;
      }
      prev_rk = -1.0f;
      continue;
    }
    float bound = radius > 0.0f ? radius2 : INFINITY;  // photons beyond the radius never count (:141)
    bool tight = false;                                // the bound encloses k photons
    if (prev_rk >= 0.0f) {
      const float dx = qx - pqx, dy = qy - pqy, dz = qz - pqz;
      const float r = prev_rk + sqrt32(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
      const float b = r * r * 1.0001f;
      // a bound far above the previous k-th distance (queries further apart than r_k) still culls, but is
      // worth tightening as soon as k photons are in the pool
      if (b < bound) bound = b, tight = b <= loose * prev_rk * prev_rk;
    }
    uint32_t vb[RMAX];
    int st_nodes = 0, st_leaves = 0, st_sel = 0;  // tuning counters (stats != NULL only)
    uint32_t Tk = 0x7f800000u;  // fewer than k photons: all of them
    count = 0;

    // one step of the walk: the children of node `prefix` at level lvl (boxes off .. ), one per lane
    auto test_group = [&](int lvl, int off, int prefix, float bnd, float& d2box) -> unsigned long long {
      const int nb = lvl == 0 ? T.b0 : KD_FAN_BITS;
      const int base = off + (prefix << nb);
      const bool act = lane < (1 << nb);
      const int rec = base + (act ? lane : 0);
      float4 lo, hi;
      if (base + 64 <= GATHER_LDS_BOXES) lo = top_lo[rec], hi = top_hi[rec];
      else lo = T.boxes[rec], hi = T.boxes[T.n_boxes + rec];
      const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.0f);
      const float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.0f);
      const float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.0f);
      d2box = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      st_nodes++;
      return __builtin_amdgcn_ballot_w64(act & (d2box <= bnd));
    };
    // one leaf bucket: photons within bnd go to the pool
    auto append_leaf = [&](int leaf, float bnd) {
      // bucket `leaf` = slots 64 leaf .. 64 leaf + 63 (the slots beyond its photons hold NaN: never within a bound)
      const uint32_t i = ((uint32_t)leaf << 6) + (uint32_t)lane;
      // 32-bit byte offset from the scalar base (12 x slots < 2^32: checked when the map is built)
      const P3 p = *(const P3*)((const char*)T.pos3 + 12u * i);  // one 12-byte load per lane, six whole lines per bucket
      const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
      const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      const unsigned long long m = __builtin_amdgcn_ballot_w64(d2 <= bnd);
      const int slot = count + mbcnt64(m);
      if ((m >> lane) & 1ull) {
        pd2[slot] = d2;
        pix[slot] = (int)i;
      }
      count += popc64(m);
      st_leaves++;
    };
    // which pending child next: in index order under a tight bound, nearest box first otherwise (so that
    // the pool reaches k photons, and the bound its final size, after as few buckets as possible)
    auto pick_child = [&](unsigned long long pending, float d2b) -> int {
      if (tight) return __builtin_ctzll(pending);
      const bool in = (pending >> lane) & 1ull;
      const uint32_t dmin = wave_total_min_u(in ? __float_as_uint(d2b) : 0xffffffffu);
      return __builtin_ctzll(__builtin_amdgcn_ballot_w64(in & (__float_as_uint(d2b) == dmin)));
    };
    auto load_pool = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const int j = lane + 64 * r;
        vb[r] = j < count ? __float_as_uint(pd2[j]) : 0xffffffffu;
      }
    };
    // refresh the bound from the pool; compact the pool when it is full
    auto reselect = [&](bool compact) {
      st_sel++;
      load_pool();
      const uint32_t Tk = select_kth<RMAX>(vb, count, k, lane, hist);
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
          const int slot = nc + mbcnt64(m);
          if (keep) {
            pd2[slot] = __uint_as_float(vb[r]);
            pix[slot] = ix;
          }
          nc += popc64(m);
        }
        count = nc;
      }
    };

    if (T.n > 0) {
      // A query with no bound yet (the first of a wave's range, or after a query that saw fewer than k photons) starts the walk
      // below with bound = inf: nearest child first down to the nearest buckets, whose photons all enter the pool; as soon as k
      // are in, their k-th distance becomes the bound, the buckets still pending are tested against it and every level's pending
      // mask is tested again when the walk returns to it (`stale`).  (Until round 3 a separate descent found that first bound and
      // the walk then started over: the nearest buckets were read twice.)
      // depth-first walk over the boxes within the bound
      int stk_lo = 0, stk_hi = 0;  // pending-children masks, level l in lane l
      int lvl = 0, off = 0, nl = 1 << T.b0, prefix = 0;
      uint32_t stale = 0u;         // bit l: the bound has shrunk since level l's pending mask was made
      float d2box;
      unsigned long long mask = test_group(0, 0, 0, bound, d2box);
      for (;;) {
        if (lvl == last) {  // the children are leaf buckets
          const int pb = prefix << (lvl == 0 ? T.b0 : KD_FAN_BITS);
          while (mask) {
            const int l = pick_child(mask, d2box);
            mask &= ~(1ull << l);
            append_leaf(pb | l, bound);
            const bool full = count > CAP - 64;
            if (full || (!tight && count >= k)) {
              const float was = bound;
              reselect(full);
              if (bound < was) {  // the buckets still pending were chosen under the old bound (d2box is this group's)
                mask &= __builtin_amdgcn_ballot_w64(d2box <= bound);
                stale = (1u << lvl) - 1u;
              }
            }
          }
        }
        if (mask == 0ull) {  // back to the parent's pending children
          if (lvl == 0) break;
          lvl--;
          nl >>= KD_FAN_BITS;
          off -= nl;
          prefix >>= (lvl == 0 ? T.b0 : KD_FAN_BITS);
          mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(stk_hi, lvl) << 32) |
                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(stk_lo, lvl);
          if (mask && ((stale >> lvl) & 1u)) {  // test this level's group again under the bound of now (one step; the distances are back too)
            stale &= ~(1u << lvl);
            mask &= test_group(lvl, off, prefix, bound, d2box);
          }
          continue;
        }
        const int l = pick_child(mask, d2box);  // (a mask popped from the stack has lost its distances: index order)
        mask &= ~(1ull << l);
        const bool me = lane == lvl;
        stk_lo = me ? (int)(uint32_t)mask : stk_lo;
        stk_hi = me ? (int)(uint32_t)(mask >> 32) : stk_hi;
        prefix = (prefix << (lvl == 0 ? T.b0 : KD_FAN_BITS)) | l;
        off += nl;
        nl <<= KD_FAN_BITS;
        lvl++;
        mask = test_group(lvl, off, prefix, bound, d2box);
      }
    }

    // final selection, then the loop of photon_mapping.go:138-161 over the pool
    load_pool();
    if (count >= k) Tk = select_kth<RMAX>(vb, count, k, lane, hist);
    const float nx = qnrm[3 * (size_t)qi], ny = qnrm[3 * (size_t)qi + 1], nz = qnrm[3 * (size_t)qi + 2];
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    uint32_t r2 = 0u;
    int found = 0;
    bool found_any = false;  // some photon lies within reach (the radius, or anywhere)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int j = lane + 64 * r;
      if (64 * r < count) {  // wave-uniform
        const float d2 = __uint_as_float(vb[r]);
        bool acc = false;
        found_any |= __builtin_amdgcn_ballot_w64(j < count && (radius == 0.0f || d2 <= radius2)) != 0ull;
        if (j < count && vb[r] <= Tk && (radius == 0.0f || d2 < radius2)) {  // :141
          const float4 pw = T.pwb[pix[j]];
          const int bits = __float_as_int(pw.w);
          const int th = (bits >> 8) & 255, ph = (bits >> 16) & 255;
          const float st = lut[256 + th];
          const float dx = st * lut[512 + ph], dy = st * lut[768 + ph], dz = lut[th];  // PhotonDir, :117-122
          if (dx * nx + dy * ny + dz * nz < 0.0f) {                                   // :144
            sx += pw.x, sy += pw.y, sz += pw.z;
            r2 = max(r2, vb[r]);
            acc = true;
          }
        }
        found += popc64(__builtin_amdgcn_ballot_w64(acc));
      }
    }
    if (found > 0) {  // wave-uniform
      sx = wave_total_add_f(sx), sy = wave_total_add_f(sy), sz = wave_total_add_f(sz);
      if (found >= 8) {  // :154-161
        const float tmp = rcp32(PI_F * __uint_as_float(wave_total_max_u(r2)));
        sx *= tmp, sy *= tmp, sz *= tmp;
      }
    }
    if (stats && lane == 0) {
      atomicAdd(stats + 0, (unsigned long long)st_nodes);
      atomicAdd(stats + 1, (unsigned long long)st_leaves);
      atomicAdd(stats + 2, (unsigned long long)st_sel + (count >= k ? 1 : 0));
      atomicAdd(stats + 3, (unsigned long long)count);
    }
    // The k-th distance is known when the selection saw k photons inside the radius (a pool that reaches beyond the
    // radius may hold k photons of which fewer than k count)
    const bool kth_known = count >= k && (radius == 0.0f || __uint_as_float(Tk) <= radius2);
    if (lane == 0) {
      irrad[3 * (size_t)qi] = sx, irrad[3 * (size_t)qi + 1] = sy, irrad[3 * (size_t)qi + 2] = sz;
      if (rk2_out) rk2_out[qi] = kth_known ? __uint_as_float(Tk) : (found_any ? INFINITY : -1.0f);
    }
    // The bound handed to the next query must be the true k-th distance over ALL photons.
    if (kth_known) prev_rk = sqrt32(__uint_as_float(Tk)) * 1.00001f, pqx = qx, pqy = qy, pqz = qz;
    else prev_rk = -1.0f;
  }
}

#define GRT_GATHER_MAX_K 512

// Workspace of a gather of n queries: the k-th squared distances (n floats, what the caller reads).
GRT_API uint64_t grt_photon_gather_workspace_bytes(int32_t n) { return n > 0 ? (uint64_t)n * 4 : 0; }

// Tuning knobs (tools/tune_gather.sh), read ONCE when the library first gathers:
//   GRT_GATHER_QPW    queries served back to back by one wave (they share the carried bound); default 24 for the
//                     unbounded search, 8 with a radius (queries near the caustic cost far more than the rest)
//   GRT_GATHER_WAVES  waves per workgroup of the k <= 128 kernel (4, 8, 16)
//   GRT_GATHER_LOOSE  carried bound / previous r_k^2 above which the bound counts as loose
//   GRT_GATHER_STATS  per-launch totals on stderr (allocates and synchronises: tuning runs only)
struct GatherTune {
  int qpw = 0, waves = 16;
  float loose = 6.0f;
  bool stats = false;
  GatherTune() {
    if (const char* e = getenv("GRT_GATHER_QPW")) qpw = atoi(e);
    if (const char* e = getenv("GRT_GATHER_WAVES")) waves = atoi(e);
    if (const char* e = getenv("GRT_GATHER_LOOSE")) loose = (float)atof(e);
    stats = getenv("GRT_GATHER_STATS") != nullptr;
  }
};
static const GatherTune& gather_tune() {
  static const GatherTune t;
  return t;
}

static int gather_launch(const grt_pmap* pm, const float* d_pos, const float* d_nrm, int n, float radius, int k, float* d_irrad,
                         float* d_rk2, hipStream_t st) {
  const GatherTune& tune = gather_tune();
  const int qpw = tune.qpw > 0 ? tune.qpw : (radius > 0.0f ? 8 : 24);
  KdView T = {pm->d_boxes, pm->d_pos3, pm->d_pwb, pm->d_lut, pm->n, pm->depth, pm->levels, pm->b0, pm->n_boxes};
  unsigned long long* d_stats = nullptr;
  if (tune.stats) {
    GRT_HIP(hipMalloc(&d_stats, 16 * sizeof(unsigned long long)));
    GRT_HIP(hipMemsetAsync(d_stats, 0, 16 * sizeof(unsigned long long), st));
  }
  // waves per workgroup by pool size, so that the LDS pools of a CU's 32 waves fit in 160 KB
#define GRT_GATHER_LAUNCH(RMAX, WAVES)                                                                                                  \
  hipLaunchKernelGGL((k_gather_knn<RMAX, WAVES>), dim3((unsigned)((n + WAVES * qpw - 1) / (WAVES * qpw))), dim3(64 * WAVES), 0, st, T, d_pos, \
                     d_nrm, n, qpw, radius, k, d_irrad, d_rk2, d_stats, tune.loose)
  if (k <= 128 && tune.waves == 4) GRT_GATHER_LAUNCH(5, 4);
  else if (k <= 128 && tune.waves == 8) GRT_GATHER_LAUNCH(5, 8);
  else if (k <= 128) GRT_GATHER_LAUNCH(5, 16);
  else if (k <= 256) GRT_GATHER_LAUNCH(9, 8);
  else GRT_GATHER_LAUNCH(17, 4);
#undef GRT_GATHER_LAUNCH
  GRT_HIP(hipGetLastError());
  if (d_stats) {
    unsigned long long h[16];
    GRT_HIP(hipStreamSynchronize(st));
    GRT_HIP(hipMemcpy(h, d_stats, sizeof h, hipMemcpyDeviceToHost));
    (void)hipFree(d_stats);
    fprintf(stderr, "[gather stats] n=%d k=%d radius=%g qpw=%d: node steps %.1f, leaf buckets %.2f, selections %.2f, pool %.1f per query\n",
            n, k, (double)radius, qpw, (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n);
  }
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
                                  float* d_irrad, void* d_workspace, void* stream) try {
  int rc = gather_check(pm, d_pos, d_nrm, d_irrad, n, k);
  if (rc) return rc;
  if (n == 0) return GRT_OK;
  return gather_launch(pm, d_pos, d_nrm, n, radius, k, d_irrad, (float*)d_workspace, (hipStream_t)stream);
} catch (...) {
  return grt_fail_exception("grt_photon_gather_dev");
}

GRT_API int grt_photon_gather(const grt_pmap* pm, const float* pos, const float* nrm, int32_t n, float radius, int32_t k,
                              float* irrad) try {
  int rc = gather_check(pm, pos, nrm, irrad, n, k);
  if (rc) return rc;
  if (n == 0) return GRT_OK;
  DeviceGuard guard;
  GRT_HIP(hipSetDevice(pm->device));
  float *d_pos = nullptr, *d_nrm = nullptr, *d_out = nullptr;
  void* d_ws = nullptr;
  const size_t b3 = (size_t)n * 3 * sizeof(float);
  rc = GRT_OK;
  do {
    PTRY(hipMalloc(&d_pos, b3));
    PTRY(hipMalloc(&d_nrm, b3));
    PTRY(hipMalloc(&d_out, b3));
    PTRY(hipMalloc(&d_ws, grt_photon_gather_workspace_bytes(n)));
    PTRY(hipMemcpy(d_pos, pos, b3, hipMemcpyHostToDevice));
    PTRY(hipMemcpy(d_nrm, nrm, b3, hipMemcpyHostToDevice));
    rc = grt_photon_gather_dev(pm, d_pos, d_nrm, n, radius, k, d_out, d_ws, nullptr);
    if (rc) break;
    PTRY(hipMemcpy(irrad, d_out, b3, hipMemcpyDeviceToHost));
  } while (0);
  (void)hipFree(d_pos), (void)hipFree(d_nrm), (void)hipFree(d_out), (void)hipFree(d_ws);
  return rc;
} catch (...) {
  return grt_fail_exception("grt_photon_gather");
}

// ---- photon-mapped trace() (scene.go:308-320, commented out in the reference) -----------------
// trace() with the photon block enabled: a path that ends on a plain diffuse surface adds, besides
// the direct light, m.Color * (caustic.IrradianceEst(p,n,1,100) + global.IrradianceEst(p,n,0,100)) / pi.
// Wavefront form: samples are rendered in batches; the path-trace launch leaves one gather request
// per (pixel, sample) that ended on a diffuse surface, k_gather_knn serves them for both maps, and
// k_pm_accumulate adds weight * irradiance to the frame in a fixed order.
#define GRT_PM_BATCH 8
#define GRT_PM_K 100          // scene.go:313,316
#define GRT_PM_CAUSTIC_R 1.0f // scene.go:313
#define GRT_PMVIEW_DEBUG_K 50 // scene.go:262,268

static int pm_batch_of(int samples) { return std::min(samples, GRT_PM_BATCH); }

GRT_API uint64_t grt_render_pm_workspace_bytes(const grt_scene* s, const grt_opts* o, int32_t shard, int32_t n_shards,
                                               int32_t samples) {
  if (!s || samples <= 0) return 0;
  const int B = pm_batch_of(samples);
  const uint64_t npix = (uint64_t)grt_shard_rows(grt_scene_desc_of(s)->height, shard, n_shards) * (uint64_t)grt_scene_desc_of(s)->width;
  const uint64_t slabs = (grt_accum_workspace_bytes(s, o, shard, n_shards, B) + 255) & ~(uint64_t)255;
  const uint64_t nq = npix * (uint64_t)B;
  return slabs + npix * 3 * sizeof(float) + 5 * nq * 3 * sizeof(float) + 1024;
}

// view == 0: trace() with both maps (scene.go:308-320).  view == 1 / 2: the photon-map view of ONE map (gmap; cmap
// NULL): Scene.irradiance (scene.go:184-232, k = 100) / trace()'s debugging block (scene.go:261-271, k = 50).
static int render_pm_dev(grt_scene* s, const grt_opts* o, const grt_pmap* gmap, const grt_pmap* cmap, int view, int32_t shard,
                         int32_t n_shards, int32_t samples, float* d_rgb_sum, void* d_workspace, uint64_t workspace_bytes,
                         uint64_t* d_ray_count, void* stream) {
  if (!s || !gmap || (!cmap && !view) || !d_rgb_sum || samples <= 0) return grt_fail(GRT_EINVAL, "grt_render_accum_pm_dev: bad argument");
  const uint64_t need = grt_render_pm_workspace_bytes(s, o, shard, n_shards, samples);
  if (!d_workspace || workspace_bytes < need)
    return grt_fail(GRT_EINVAL, "grt_render_accum_pm_dev: workspace %llu bytes, need %llu", (unsigned long long)workspace_bytes,
                    (unsigned long long)need);
  const int B = pm_batch_of(samples);
  const int W = grt_scene_desc_of(s)->width;
  const size_t npix = (size_t)grt_shard_rows(grt_scene_desc_of(s)->height, shard, n_shards) * (size_t)W;
  if (npix == 0) return GRT_OK;
  const uint64_t slab_bytes = (grt_accum_workspace_bytes(s, o, shard, n_shards, B) + 255) & ~(uint64_t)255;
  char* base = (char*)d_workspace;
  float* frame = (float*)(base + slab_bytes);
  float* qpos = frame + npix * 3;
  const size_t nq = npix * (size_t)B;
  float *qnrm = qpos + nq * 3, *qw = qnrm + nq * 3, *irr_g = qw + nq * 3, *irr_c = irr_g + nq * 3;
  hipStream_t st = (hipStream_t)stream;
  if (nq > (size_t)0x7fffffff) return grt_fail(GRT_ELIMIT, "grt_render_accum_pm_dev: %zu gather requests per batch", nq);
  for (int s0 = 0, first = 1; s0 < samples; s0 += B, first = 0) {
    const int b = std::min(B, samples - s0);
    GRT_HIP(hipMemsetAsync(qpos, 0xff, nq * 3 * sizeof(float), st));  // all bits set = NaN = no request
    int rc = grt_render_dev_internal(s, o, shard, n_shards, s0, b, frame, base, slab_bytes, d_ray_count, stream, qpos, qnrm, qw, B, view);
    if (rc) return rc;
    rc = gather_launch(gmap, qpos, qnrm, (int)nq, 0.0f, view == 2 ? GRT_PMVIEW_DEBUG_K : GRT_PM_K, irr_g, nullptr, st);  // scene.go:316 | :228 | :262,268
    if (rc) return rc;
    if (cmap) rc = gather_launch(cmap, qpos, qnrm, (int)nq, GRT_PM_CAUSTIC_R, GRT_PM_K, irr_c, nullptr, st);  // scene.go:313
    if (rc) return rc;
    rc = grt_pm_accumulate_internal(frame, qpos, qw, irr_g, cmap ? irr_c : nullptr, B, npix, d_rgb_sum, first, stream);
    if (rc) return rc;
  }
  return GRT_OK;
}

GRT_API int grt_render_accum_pm_dev(grt_scene* s, const grt_opts* o, const grt_pmap* gmap, const grt_pmap* cmap, int32_t shard,
                                    int32_t n_shards, int32_t samples, float* d_rgb_sum, void* d_workspace,
                                    uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream) try {
  if (!cmap) return grt_fail(GRT_EINVAL, "grt_render_accum_pm_dev: bad argument");
  return render_pm_dev(s, o, gmap, cmap, 0, shard, n_shards, samples, d_rgb_sum, d_workspace, workspace_bytes, d_ray_count, stream);
} catch (...) {
  return grt_fail_exception("grt_render_accum_pm_dev");
}

// Host-buffer form on one device: linear RGB sums [H*W*3] (rgb_sum) and / or tone-mapped B,G,R bytes (pixels).
static int render_pm_host(grt_scene* s, const grt_opts* o, const grt_pmap* gmap, const grt_pmap* cmap, int samples, float* rgb_sum,
                          uint8_t* pixels, int pitch, int view = 0) {
  if (!s || !gmap || (!cmap && !view) || samples <= 0) return grt_fail(GRT_EINVAL, "grt_render_pm: bad argument");
  if (o && o->n_devices > 1) return grt_fail(GRT_ELIMIT, "grt_render_pm: one device per call (shard with grt_render_accum_pm_dev)");
  DeviceGuard guard;
  int dev;
  int rc = pick_device(o, &dev);
  if (!rc && (gmap->device != dev || (cmap && cmap->device != dev)))
    rc = grt_fail(GRT_EINVAL, "grt_render_pm: the photon maps live on devices %d and %d, the render was asked for device %d", gmap->device,
                  cmap ? cmap->device : gmap->device, dev);
  if (rc) return rc;
  const int W = grt_scene_desc_of(s)->width, H = grt_scene_desc_of(s)->height;
  const size_t nfl = (size_t)W * H * 3;
  const uint64_t ws_bytes = grt_render_pm_workspace_bytes(s, o, 0, 1, samples);
  float* d_sum = nullptr;
  void* d_ws = nullptr;
  uint8_t* d_bgr = nullptr;
  rc = GRT_OK;
  do {
    PTRY(hipMalloc(&d_sum, nfl * sizeof(float)));
    PTRY(hipMalloc(&d_ws, ws_bytes));
    rc = render_pm_dev(s, o, gmap, cmap, view, 0, 1, samples, d_sum, d_ws, ws_bytes, nullptr, nullptr);
    if (rc) break;
    if (rgb_sum) PTRY(hipMemcpy(rgb_sum, d_sum, nfl * sizeof(float), hipMemcpyDeviceToHost));
    if (pixels) {
      const int bpp = pitch / W;
      std::vector<uint8_t> h((size_t)W * H * 3);
      PTRY(hipMalloc(&d_bgr, h.size()));
      rc = grt_tonemap_dev(d_sum, W, H, samples, d_bgr, W * 3, nullptr);
      if (rc) break;
      PTRY(hipMemcpy(h.data(), d_bgr, h.size(), hipMemcpyDeviceToHost));
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          for (int c = 0; c < 3; ++c) pixels[(size_t)y * pitch + (size_t)x * bpp + c] = h[((size_t)y * W + x) * 3 + c];
    }
  } while (0);
  (void)hipFree(d_sum), (void)hipFree(d_ws), (void)hipFree(d_bgr);
  return rc;
}
GRT_API int grt_render_pm(grt_scene* s, const grt_opts* o, const grt_pmap* gmap, const grt_pmap* cmap, uint8_t* pixels, int32_t pitch,
                          int32_t samples) try {
  if (!pixels || !s || pitch / grt_scene_desc_of(s)->width < 3) return grt_fail(GRT_EINVAL, "grt_render_pm: NULL pixels or pitch too small");
  return render_pm_host(s, o, gmap, cmap, samples, nullptr, pixels, pitch);
} catch (...) {
  return grt_fail_exception("grt_render_pm");
}
GRT_API int grt_render_accum_pm(grt_scene* s, const grt_opts* o, const grt_pmap* gmap, const grt_pmap* cmap, float* rgb_sum,
                                int32_t samples) try {
  if (!rgb_sum) return grt_fail(GRT_EINVAL, "grt_render_accum_pm: NULL output");
  return render_pm_host(s, o, gmap, cmap, samples, rgb_sum, nullptr, 0);
} catch (...) {
  return grt_fail_exception("grt_render_accum_pm");
}

// ---- photon-map view ------------------------------------------------------
// Scene.irradiance (scene.go:184-232) has no caller upstream; it is what Render shows when it stands in for
// trace() at scene.go:96: mirrors and glass are followed (weights f, 1), every other surface shows
// pmap.IrradianceEst(p, n, 0, 100).  GRT_PMVIEW_FIRST_HIT is the debugging block inside trace()
// (scene.go:261-271): IrradianceEst(p, n, 0, 50) at the first hit, whatever its material.
static int pmview_mode(int32_t mode, int* view) {
  if (mode != GRT_PMVIEW_IRRADIANCE && mode != GRT_PMVIEW_FIRST_HIT) return grt_fail(GRT_EINVAL, "grt_render_pmview: unknown mode %d", mode);
  *view = mode == GRT_PMVIEW_IRRADIANCE ? 1 : 2;
  return GRT_OK;
}
GRT_API int grt_render_pmview(grt_scene* s, const grt_opts* o, const grt_pmap* pmap, int32_t mode, uint8_t* pixels, int32_t pitch,
                              int32_t samples) try {
  if (!pixels || !s || pitch / grt_scene_desc_of(s)->width < 3) return grt_fail(GRT_EINVAL, "grt_render_pmview: NULL pixels or pitch too small");
  int view;
  const int rc = pmview_mode(mode, &view);
  return rc ? rc : render_pm_host(s, o, pmap, nullptr, samples, nullptr, pixels, pitch, view);
} catch (...) {
  return grt_fail_exception("grt_render_pmview");
}
GRT_API int grt_render_accum_pmview(grt_scene* s, const grt_opts* o, const grt_pmap* pmap, int32_t mode, float* rgb_sum,
                                    int32_t samples) try {
  if (!rgb_sum) return grt_fail(GRT_EINVAL, "grt_render_accum_pmview: NULL output");
  int view;
  const int rc = pmview_mode(mode, &view);
  return rc ? rc : render_pm_host(s, o, pmap, nullptr, samples, rgb_sum, nullptr, 0, view);
} catch (...) {
  return grt_fail_exception("grt_render_accum_pmview");
}
