// grt_bvh.h — Scene.intersect (pkg/tracer/scene.go:169-181) for scenes too large for the
// reference's linear scan (config C5: 10^6 spheres).  The reference HAS no acceleration
// structure (it calls every object's Hit for every ray, scene.go:173); this is the
// order-free restatement of that loop over a bounding-volume hierarchy.
//
// What the linear loop computes, independent of visiting order (derivation in DESIGN.md §5):
//   h(o) = the distance object o reports against tMax = +inf with the accept test `ht > 0`:
//          sphere: first root > tMin (sphere.go:36-43); box: t0 when t1 > max(tMin, t0) and
//          t0 > 0 (aabb.go:33-44, scene.go:174);
//   Q    = objects with 0 < h <= tMin (a box entered within the bias, aabb.go:44 returns t0
//          not max(tMin,t0)): the lowest-index member of Q wins outright, because once it is
//          accepted tMax <= tMin and every later `hi > lo` test fails;
//   else   the smallest h wins, the lowest index among equal h (later objects replace the
//          best only when strictly closer).
//
// Layout (HBM): a binary tree, 32 B per node = the boxes of BOTH children, so one fetch
// decides two subtrees.  The boxes live on a 16-bit grid over the scene's bounds (world =
// gmin + q * gs per axis), rounded outward by one cell, so a node is two 16-byte words:
//   nodes[2n+c] = (lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, ref)   c = child 0, 1
//   ref >= 0: inner node index; ref < 0: leaf, ~ref = first object slot << 3 | count (<= 7;
//   1 per leaf walks fastest at C5, deeper scenes take 2, 4 or 7 to fit the traversal stack).
//   The inner children of a node are neighbours in memory (a 64-byte pair on an even slot, followed
//   by the pair of the first child's children): a 128-byte line brings the far child with the near one.
// With 64-byte FP32 nodes the texture addresser was busy 76 % of the kernel at C5: the vector
// memory path moves bytes per lane, and halving the node halved that traffic.
// A ray is taken into grid coordinates once (BvhRay): t = q * (gs / D) - (O - gmin) / D.
// Objects are re-ordered into leaf order (bprims: the 2 x float4 records of KScene::prims with
// the last word replaced by object index << 3 | transparent << 2 | emissive << 1 | is_box, so a
// leaf needs no second table).
// Object boxes are padded by a few ulps before the unions are taken and one grid cell after,
// so a subtree is never culled by a rounding difference between the slab arithmetic here and
// the object arithmetic of sphere_hit32 / aabb_hit32.
// Traversal: one ray per lane, nearer child first, the other child on a per-lane stack in LDS
// (stack[depth][lane]: conflict-free).  The same tree serves the reference's two other object loops
// through bvh_any_hit (the shadow loop of the diffuse branch, the caustic filter of tracePhotons).
#pragma once
// (included from the middle of grt_device.h, after the per-object hit functions)

#define GRT_BVH_LEAF 1
#define GRT_BVH_STACK 24
#define GRT_BVH_DONE 0x7fffffff

struct BvhRay {
  f3 invD, OinvD;  // gs / D, (O - gmin) / D
};
template <class RAY>
GRT_DEV void bvh_ray_setup(BvhRay& g, const RAY& r, const BvhGrid& G) {
  const f3 invD = mk3(rcp32(r.D.x), rcp32(r.D.y), rcp32(r.D.z));
  g.invD = G.gs * invD;
  g.OinvD = mk3(r.O.x - G.gmin.x, r.O.y - G.gmin.y, r.O.z - G.gmin.z) * invD;
}

template <bool CAREFUL>
GRT_DEV bool bvh_box(uint4 q, const BvhRay& r, f3 Og, float tbest, float& tn) {
#ifdef GRT_NOCULL  // experiment (tools/exp_c5_pixels.py): no box is ever culled, the walk becomes the linear scan in tree order
  tn = 0.0f;
  return true;
#endif
  const float lx = (float)(q.x & 0xffffu), ly = (float)(q.x >> 16), lz = (float)(q.y & 0xffffu);
  const float hx = (float)(q.y >> 16), hy = (float)(q.z & 0xffffu), hz = (float)(q.z >> 16);
  float x0, x1, y0, y1, z0, z1;
  if (!CAREFUL) {
    x0 = fmaf(lx, r.invD.x, -r.OinvD.x), x1 = fmaf(hx, r.invD.x, -r.OinvD.x);
    y0 = fmaf(ly, r.invD.y, -r.OinvD.y), y1 = fmaf(hy, r.invD.y, -r.OinvD.y);
    z0 = fmaf(lz, r.invD.z, -r.OinvD.z), z1 = fmaf(hz, r.invD.z, -r.OinvD.z);
  } else {  // a zero direction component: (q - O) * (+-inf); 0 * inf = NaN is dropped by min/max = miss,
            // which is right because every object lies strictly inside its padded box
    x0 = (lx - Og.x) * r.invD.x, x1 = (hx - Og.x) * r.invD.x;
    y0 = (ly - Og.y) * r.invD.y, y1 = (hy - Og.y) * r.invD.y;
    z0 = (lz - Og.z) * r.invD.z, z1 = (hz - Og.z) * r.invD.z;
  }
  tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
  // a hit distance h in (0, tbest] of an object inside the box lies in [tn, tf]
  return (tn <= tf) & (tf > 0.0f) & (tn <= tbest);
}

struct BvhHit {
  float tbest;  // smallest h so far (objects outside Q)
  int best;
  float qt;     // Q: lowest-index object entered within the bias
  int qbest;
};

template <bool CAREFUL, class RAY>
GRT_DEV void bvh_object(float4 pa, float4 pb, const RAY& r, float bias, BvhHit& H) {
  float h = -1.0f;
  const int id = (int)((uint32_t)__float_as_int(pb.w) >> 3);
  if (__float_as_int(pb.w) & 1) {
    h = aabb_hit32<CAREFUL>(pa, pb, full_ray(r), bias, 3.402823466e38f);  // t0, or -1
  } else {
    const float inva = ray_inva(r);
    const f3 oc = mk3(r.O.x - pa.x, r.O.y - pa.y, r.O.z - pa.z);
    const float k = dot3(oc, r.D) * inva;
    const f3 l = fma3(r.D, -k, oc);
    const float sq = sqrt32((pb.x - dot3(l, l)) * inva);  // NaN when there is no real root
    const float t1 = -k - sq, t2 = -k + sq;
    h = t1 > bias ? t1 : (t2 > bias ? t2 : -1.0f);
  }
  if (!(h > 0.0f)) return;  // scene.go:174
  if (h <= bias) {
    if (id < H.qbest) H.qbest = id, H.qt = h;
  } else if (h < H.tbest || (h == H.tbest && id < H.best)) {
    H.tbest = h, H.best = id;
  }
}

// A walk that can be suspended between rounds: the reference in hand, the stack height (the stack
// itself is the lane's column of the LDS stack, entry d at stack[d * stride]) and the hit so far.
struct BvhWalk {
  int ref, sp;
  BvhHit H;
};
GRT_DEV void bvh_walk_begin(BvhWalk& w) {
  w.H.tbest = 3.402823466e38f, w.H.best = -1, w.H.qt = 0.0f, w.H.qbest = 0x7fffffff;
  w.sp = 0;
  w.ref = 0;  // the root is an inner node
}
GRT_DEV int bvh_walk_result(const BvhWalk& w, float& t_out) {
  if (w.H.qbest != 0x7fffffff) {
    t_out = w.H.qt;
    return w.H.qbest;
  }
  t_out = w.H.tbest;
  return w.H.best;
}
// Stack entry d of this lane.  stride > 0: `stack` is the lane's own column, entry d at
// stack[d * stride].  stride == 0: `stack` is the wave's [depth][64] block and the lane index is
// taken afresh at every access (two v_mbcnt) — a lane offset held in a VGPR across the walk is the
// first thing the register allocator spills at 7 waves per SIMD, and a scratch reload per stack
// access costs far more than recomputing it.
GRT_DEV int* bvh_slot(int* stack, int stride, int d) {
  if (stride > 0) return stack + d * stride;
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return stack + d * 64 + lane;
}

// One inner node for the lanes that call it (w.ref is an inner node): both children's boxes, the
// nearer hit child next, the other on the stack.
template <bool CAREFUL, class RAY>
GRT_DEV void bvh_inner_apply(uint4 c0, uint4 c1, const BvhGrid& G, const RAY& r, const BvhRay& g, BvhWalk& w, int* stack, int stride) {
  f3 Og = mk3(0.0f, 0.0f, 0.0f);  // (O - gmin) / gs: only rays with a zero direction component read it
  if (CAREFUL) Og = mk3(r.O.x - G.gmin.x, r.O.y - G.gmin.y, r.O.z - G.gmin.z) * G.grs;
  float tn0, tn1;
  const bool h0 = bvh_box<CAREFUL>(c0, g, Og, w.H.tbest, tn0);
  const bool h1 = bvh_box<CAREFUL>(c1, g, Og, w.H.tbest, tn1);
  const int r0 = (int)c0.w, r1 = (int)c1.w;
  // Straight-line selects, two exec-masked LDS accesses: the lane masks h0, h1 stay in SGPR pairs (as branches of an
  // if / else-if chain the compiler turned them into 0/1 VGPRs and compared those again: 8 VALU per node)
  const bool first0 = tn0 <= tn1;
  const bool both = h0 & h1, none = !(h0 | h1);
  const int near = (h0 & (first0 | !h1)) ? r0 : r1;  // the only child hit, or the nearer of two
  const int far = first0 ? r1 : r0;
  const bool pop = none & (w.sp > 0);
  const int top = w.sp - (pop ? 1 : 0);               // entry read (pop) or written (both)
  int* slot = bvh_slot(stack, stride, top);
  int next = GRT_BVH_DONE;
  if (pop) next = *slot;
  if (both) *slot = far;
  w.sp = top + (both ? 1 : 0);
  w.ref = none ? next : near;
}
template <bool CAREFUL, class RAY>
GRT_DEV void bvh_inner_step(const uint4* __restrict__ nodes, const BvhGrid& G, const RAY& r, const BvhRay& g, BvhWalk& w, int* stack,
                            int stride) {
  const uint4 c0 = nodes[2 * (size_t)w.ref], c1 = nodes[2 * (size_t)w.ref + 1];
  bvh_inner_apply<CAREFUL>(c0, c1, G, r, g, w, stack, stride);
}
// One leaf for the lanes that call it (w.ref is a leaf): its objects, then the next reference.
template <bool CAREFUL, class RAY>
GRT_DEV void bvh_leaf_step(const float4* __restrict__ bprims, const RAY& r, float bias, BvhWalk& w, int* stack, int stride) {
  const int first = (~w.ref) >> 3, cnt = (~w.ref) & 7;
  for (int j = 0; j < cnt; ++j) {
    const float4 pa = bprims[2 * (size_t)(first + j)], pb = bprims[2 * (size_t)(first + j) + 1];
    bvh_object<CAREFUL>(pa, pb, r, bias, w.H);
  }
  w.ref = w.sp > 0 ? *bvh_slot(stack, stride, --w.sp) : GRT_BVH_DONE;
}
// The first object of the leaf in hand has been fetched and tested: the rest of the leaf, or the
// next reference when that was its last object.
GRT_DEV void bvh_leaf_advance(BvhWalk& w, int* stack, int stride) {
  const int first = (~w.ref) >> 3, cnt = (~w.ref) & 7;
  if (cnt > 1) w.ref = ~(((first + 1) << 3) | (cnt - 1));
  else w.ref = w.sp > 0 ? *bvh_slot(stack, stride, --w.sp) : GRT_BVH_DONE;
}
GRT_DEV bool bvh_at_inner(const BvhWalk& w) { return w.ref >= 0 && w.ref != GRT_BVH_DONE; }
GRT_DEV bool bvh_at_leaf(const BvhWalk& w) { return w.ref < 0; }

// Every lane of the wave traces one ray to the end (while-while: inner nodes until every walking
// lane holds a leaf or has finished, then the leaves).
template <bool CAREFUL>
GRT_DEV int intersect_bvh_t(const uint4* __restrict__ nodes, const float4* __restrict__ bprims, const BvhGrid& G, const Ray32& r,
                            float bias, float& t_out, int* stack, int stride) {
  BvhWalk w;
  BvhRay g;
  bvh_walk_begin(w);
  bvh_ray_setup(g, r, G);
  while (__any(w.ref != GRT_BVH_DONE)) {
    while (__any(bvh_at_inner(w)))
      if (bvh_at_inner(w)) bvh_inner_step<CAREFUL>(nodes, G, r, g, w, stack, stride);
    if (bvh_at_leaf(w)) bvh_leaf_step<CAREFUL>(bprims, r, bias, w, stack, stride);
  }
  return bvh_walk_result(w, t_out);
}
// Any-hit query: is there an object with index in (idx_lo, idx_hi), every bit of `require` and no
// bit of `reject` set in its flags (1 box, 2 emissive, 4 transparent), whose Hit against the window
// (bias, cap) — the same aabb_hit32 / sphere_hit32 calls the linear loops make — is accepted
// (ht > 0)?  This is the order-free core of the reference's two "does anything ... " loops:
//   * the caustic filter of tracePhotons (scene.go:359-363): any transparent object, cap = +inf;
//   * the shadow loop of trace() (scene.go:336-344), see direct_light_bvh in grt_device.h.
// Wave-cooperative (while-while); a lane with active == false takes no part, a lane that has
// found an object stops walking.
template <bool CAREFUL>
GRT_DEV bool bvh_any_hit(const uint4* __restrict__ nodes, const float4* __restrict__ bprims, const BvhGrid& G, const Ray32& r,
                         float bias, float cap, int idx_lo, int idx_hi, int require, int reject, bool active, int* stack,
                         int stride) {
  BvhRay g;
  bvh_ray_setup(g, r, G);
  f3 Og = mk3(0.0f, 0.0f, 0.0f);
  if (CAREFUL) Og = mk3(r.O.x - G.gmin.x, r.O.y - G.gmin.y, r.O.z - G.gmin.z) * G.grs;
  int ref = active ? 0 : GRT_BVH_DONE, sp = 0;
  bool found = false;
  while (__any(ref != GRT_BVH_DONE)) {
    while (__any(ref >= 0 && ref != GRT_BVH_DONE)) {
      if (ref >= 0 && ref != GRT_BVH_DONE) {
        const uint4 c0 = nodes[2 * (size_t)ref], c1 = nodes[2 * (size_t)ref + 1];
        float tn0, tn1;
        const bool h0 = bvh_box<CAREFUL>(c0, g, Og, cap, tn0);
        const bool h1 = bvh_box<CAREFUL>(c1, g, Og, cap, tn1);
        if (h0 & h1) {
          *bvh_slot(stack, stride, sp) = (int)c1.w;
          sp++;
          ref = (int)c0.w;
        } else if (h0 | h1) {
          ref = h0 ? (int)c0.w : (int)c1.w;
        } else {
          ref = sp > 0 ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;
        }
      }
    }
    if (ref < 0) {
      const int first = (~ref) >> 3, cnt = (~ref) & 7;
      for (int j = 0; j < cnt; ++j) {
        const float4 pa = bprims[2 * (size_t)(first + j)], pb = bprims[2 * (size_t)(first + j) + 1];
        const int bits = __float_as_int(pb.w), id = (int)((uint32_t)bits >> 3);
        if ((id > idx_lo) & (id < idx_hi) & ((bits & require) == require) & ((bits & reject) == 0)) {
          const float ht = (bits & 1) ? aabb_hit32<CAREFUL>(pa, pb, r, bias, cap) : sphere_hit32(pa, pb, r, bias, cap);
          found |= ht > 0.0f;
        }
      }
      ref = (!found && sp > 0) ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;
    }
  }
  return found;
}
static_assert(GRT_BVH_LEAF <= 7, "a leaf reference holds its object count in 3 bits");
// Measured alternative (C5, 16 spp): the same tree folded into 4-ary nodes (128 B, four slab tests and
// a sorting network per step, half the dependent fetches) took 92.6 ms against 60.7 ms for this
// binary layout: the walk is bound by divergence (20 % lane utilisation), not by fetch latency alone.
