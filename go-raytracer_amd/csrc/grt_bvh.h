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
// Layout (HBM): a binary tree, 64 B per node = the boxes of BOTH children, so one fetch
// decides two subtrees:
//   nodes[4n+0] = (lo0.xyz, ref0)  nodes[4n+1] = (hi0.xyz, 0)
//   nodes[4n+2] = (lo1.xyz, ref1)  nodes[4n+3] = (hi1.xyz, 0)
//   ref >= 0: inner node index; ref < 0: leaf, ~ref = first object slot << 3 | count (<= 7;
//   2 per leaf walks fastest at C5, deeper scenes take 4 or 7 to fit the traversal stack).
// Objects are re-ordered into leaf order (bprims: the 2 x float4 records of KScene::prims;
// bids: slot -> index into Scene.Objects).  Object boxes are padded by a few ulps before the
// unions are taken, so a subtree is never culled by a rounding difference between the slab
// arithmetic here and the sphere arithmetic of sphere_hit32.
// Traversal: one ray per lane, nearer child first, the other child on a per-lane stack in LDS
// (stack[depth][lane]: conflict-free).
#pragma once
// (included from the middle of grt_device.h, after the per-object hit functions)

#define GRT_BVH_LEAF 2
#define GRT_BVH_STACK 24
#define GRT_BVH_DONE 0x7fffffff

template <bool CAREFUL>
GRT_DEV bool bvh_box(float4 lo, float4 hi, const Ray32& r, float tbest, float& tn) {
  float x0, x1, y0, y1, z0, z1;
  if (!CAREFUL) {
    x0 = fmaf(lo.x, r.invD.x, -r.OinvD.x), x1 = fmaf(hi.x, r.invD.x, -r.OinvD.x);
    y0 = fmaf(lo.y, r.invD.y, -r.OinvD.y), y1 = fmaf(hi.y, r.invD.y, -r.OinvD.y);
    z0 = fmaf(lo.z, r.invD.z, -r.OinvD.z), z1 = fmaf(hi.z, r.invD.z, -r.OinvD.z);
  } else {  // a zero direction component: (b - O) * (+-inf); 0 * inf = NaN is dropped by min/max = miss,
            // which is right because every object lies strictly inside its padded box
    x0 = (lo.x - r.O.x) * r.invD.x, x1 = (hi.x - r.O.x) * r.invD.x;
    y0 = (lo.y - r.O.y) * r.invD.y, y1 = (hi.y - r.O.y) * r.invD.y;
    z0 = (lo.z - r.O.z) * r.invD.z, z1 = (hi.z - r.O.z) * r.invD.z;
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

template <bool CAREFUL>
GRT_DEV void bvh_object(float4 pa, float4 pb, int id, const Ray32& r, float bias, BvhHit& H) {
  float h = -1.0f;
  if (__float_as_int(pb.w) & 1) {
    h = aabb_hit32<CAREFUL>(pa, pb, r, bias, 3.402823466e38f);  // t0, or -1
  } else {
    const f3 oc = mk3(r.O.x - pa.x, r.O.y - pa.y, r.O.z - pa.z);
    const float k = dot3(oc, r.D) * r.inva;
    const f3 l = fma3(r.D, -k, oc);
    const float sq = sqrt32((pb.x - dot3(l, l)) * r.inva);  // NaN when there is no real root
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
// One round for the lanes that call it (the walking lanes of the wave): inner nodes until every one
// of them holds a leaf or has finished (lanes at a leaf wait), then the leaves.
template <bool CAREFUL>
GRT_DEV void bvh_walk_round(const float4* __restrict__ nodes, const float4* __restrict__ bprims, const int* __restrict__ bids,
                            const Ray32& r, float bias, BvhWalk& w, int* stack, int stride) {
  int ref = w.ref, sp = w.sp;
  while (__any(ref >= 0 && ref != GRT_BVH_DONE)) {
    if (ref >= 0 && ref != GRT_BVH_DONE) {
      const float4 a0 = nodes[4 * (size_t)ref], a1 = nodes[4 * (size_t)ref + 1];
      const float4 b0 = nodes[4 * (size_t)ref + 2], b1 = nodes[4 * (size_t)ref + 3];
      float tn0, tn1;
      const bool h0 = bvh_box<CAREFUL>(a0, a1, r, w.H.tbest, tn0);
      const bool h1 = bvh_box<CAREFUL>(b0, b1, r, w.H.tbest, tn1);
      const int r0 = __float_as_int(a0.w), r1 = __float_as_int(b0.w);
      if (h0 & h1) {
        const bool first0 = tn0 <= tn1;
        stack[sp * stride] = first0 ? r1 : r0;
        sp++;
        ref = first0 ? r0 : r1;
      } else if (h0 | h1) {
        ref = h0 ? r0 : r1;
      } else if (sp > 0) {
        ref = stack[--sp * stride];
      } else {
        ref = GRT_BVH_DONE;
      }
    }
  }
  if (ref != GRT_BVH_DONE) {  // a leaf
    const int first = (~ref) >> 3, cnt = (~ref) & 7;
    for (int j = 0; j < cnt; ++j) {
      const float4 pa = bprims[2 * (size_t)(first + j)], pb = bprims[2 * (size_t)(first + j) + 1];
      bvh_object<CAREFUL>(pa, pb, bids[first + j], r, bias, w.H);
    }
    ref = sp > 0 ? stack[--sp * stride] : GRT_BVH_DONE;
  }
  w.ref = ref, w.sp = sp;
}

template <bool CAREFUL>
GRT_DEV int intersect_bvh_t(const float4* __restrict__ nodes, const float4* __restrict__ bprims, const int* __restrict__ bids,
                            const Ray32& r, float bias, float& t_out, int* stack, int stride) {
  BvhWalk w;
  bvh_walk_begin(w);
  while (__any(w.ref != GRT_BVH_DONE))
    if (w.ref != GRT_BVH_DONE) bvh_walk_round<CAREFUL>(nodes, bprims, bids, r, bias, w, stack, stride);
  return bvh_walk_result(w, t_out);
}
static_assert(GRT_BVH_LEAF <= 7, "a leaf reference holds its object count in 3 bits");
// Measured alternative (C5, 16 spp): the same tree folded into 4-ary nodes (128 B, four slab tests and
// a sorting network per step, half the dependent fetches) took 92.6 ms against 60.7 ms for this
// binary layout: the walk is bound by divergence (20 % lane utilisation), not by fetch latency alone.
