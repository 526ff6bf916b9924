// grt_parity.h — the primary-ray parity kernel's walks of the bounding-volume hierarchy (grt_bvh.h).
//
// Scene.intersect (pkg/tracer/scene.go:169-181) on hierarchy scenes, to the north-star bar: hit/miss
// and object id bit-exact against the float64 reference, t and normals to 1e-5.  Three steps, as on
// the linear-scan path (ambiguous32 / intersect64 in grt_device.h):
//   1. intersect_bvh_margin: the FP32 walk with a first-order error bound on every object test it runs
//      (hit_margin).  It certifies the winner when every test was decidable, no object was entered
//      within the bias, and the winner's interval [t - e, t + e] lies below every other accepted
//      object's.  Subtrees are culled against tbest + its bound, so every object the walk skipped lies
//      beyond the winner's interval as well.
//   2. a certified winner whose FP32 t or normal may miss 1e-5 (hit_accuracy_doubtful: at C5 a sphere
//      of radius 1 seen from 1000 units away) gets its Hit and Surface re-evaluated in float64 — one
//      object, no walk;
//   3. intersect_bvh64: lanes step 1 could not certify walk again with every object test in float64,
//      the reference's operation order (prim_hit64), and the order-free winner rule of grt_bvh.h on
//      doubles.  Boxes are still tested in FP32 on the 16-bit grid: they are padded by a whole grid cell
//      (>= 1e-6 of the scene's reach), a thousand times the distance between the float64 ray and its
//      FP32 rounding, so an object the float64 ray hits is never culled.
#pragma once
// (included from grt_kernels.h, after grt_device.h)

template <bool CAREFUL>
GRT_DEV int intersect_bvh_margin(const KScene& S, const Ray32& r, float& t_out, float& e_out, bool& amb_out, int* stack, int stride) {
  const float BIG = 3.402823466e38f;
  const float Dlen = sqrt32(r.a), Olen = sqrt32(dot3(r.O, r.O));
  BvhRay g;
  bvh_ray_setup(g, r, S.bvh_grid);
  f3 Og = mk3(0.0f, 0.0f, 0.0f);
  if (CAREFUL) Og = mk3(r.O.x - S.bvh_grid.gmin.x, r.O.y - S.bvh_grid.gmin.y, r.O.z - S.bvh_grid.gmin.z) * S.bvh_grid.grs;
  bool amb = ray_has_zero(r);  // Go's NaN / Inf rules (aabb.go:29-31): float64 decides
  int ref = amb ? GRT_BVH_DONE : 0, sp = 0, best = -1;
  float tb = BIG, eb = 0.0f;
  while (__any(ref != GRT_BVH_DONE)) {
    while (__any(ref >= 0 && ref != GRT_BVH_DONE)) {
      if (ref >= 0 && ref != GRT_BVH_DONE) {
        const uint4 c0 = S.bvh_nodes[2 * (size_t)ref], c1 = S.bvh_nodes[2 * (size_t)ref + 1];
        const float cut = tb + eb + 2e-6f * tb;  // +inf while nothing is hit
        float tn0, tn1;
        const bool h0 = bvh_box<CAREFUL>(c0, g, Og, cut, tn0);
        const bool h1 = bvh_box<CAREFUL>(c1, g, Og, cut, tn1);
        const int r0 = (int)c0.w, r1 = (int)c1.w;
        if (h0 & h1) {
          const bool first0 = tn0 <= tn1;
          *bvh_slot(stack, stride, sp) = first0 ? r1 : r0;
          sp++;
          ref = first0 ? r0 : r1;
        } else if (h0 | h1) {
          ref = h0 ? r0 : r1;
        } else {
          ref = sp > 0 ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;
        }
      }
    }
    if (ref < 0) {
      const int first = (~ref) >> 3, cnt = (~ref) & 7;
      for (int j = 0; j < cnt; ++j) {
        const float4 pa = S.bvh_prims[2 * (size_t)(first + j)], pb = S.bvh_prims[2 * (size_t)(first + j) + 1];
        const int id = (int)((uint32_t)__float_as_int(pb.w) >> 3);
        float h, e;
        const int st = hit_margin(pa, pb, r, S.bias, Dlen, Olen, BIG, BIG, h, e);
        if (st == 2) amb = true;
        if (st == 1) {
          if (!(h - e > S.bias)) amb = true;  // entered within the bias (the set Q of grt_bvh.h), or too close to tell
          else if (h + e < tb - eb) tb = h, eb = e, best = id;
          else if (!(h - e > tb + eb)) amb = true;  // order of the two distances (or, tied, of the indices) undecided
        }
      }
      ref = (!amb && sp > 0) ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;  // an uncertified lane walks again in float64
    }
  }
  t_out = tb, e_out = eb, amb_out = amb;
  return best;
}

// The reference's loop result over the hierarchy with every Hit in float64 (prim_hit64 against
// tMax = +inf, accept test ht > 0, winner by the order-free rule of grt_bvh.h).  Lanes with
// active == false take no part.
template <bool CAREFUL>
GRT_DEV int intersect_bvh64(const KScene& S, const KScene64& S64, const Ray32& r, d3 O, d3 D, bool active, double& t_out, int* stack,
                            int stride) {
  BvhRay g;
  bvh_ray_setup(g, r, S.bvh_grid);
  f3 Og = mk3(0.0f, 0.0f, 0.0f);
  if (CAREFUL) Og = mk3(r.O.x - S.bvh_grid.gmin.x, r.O.y - S.bvh_grid.gmin.y, r.O.z - S.bvh_grid.gmin.z) * S.bvh_grid.grs;
  int ref = active ? 0 : GRT_BVH_DONE, sp = 0, best = -1, qbest = 0x7fffffff;
  double tb = 1.7976931348623157e308, qt = 0.0;
  while (__any(ref != GRT_BVH_DONE)) {
    while (__any(ref >= 0 && ref != GRT_BVH_DONE)) {
      if (ref >= 0 && ref != GRT_BVH_DONE) {
        const uint4 c0 = S.bvh_nodes[2 * (size_t)ref], c1 = S.bvh_nodes[2 * (size_t)ref + 1];
        float cut = (float)tb;  // rounds to nearest: widen by two ulps so the bound never falls below tb
        cut += 2.4e-7f * cut;
        float tn0, tn1;
        const bool h0 = bvh_box<CAREFUL>(c0, g, Og, cut, tn0);
        const bool h1 = bvh_box<CAREFUL>(c1, g, Og, cut, tn1);
        const int r0 = (int)c0.w, r1 = (int)c1.w;
        if (h0 & h1) {
          const bool first0 = tn0 <= tn1;
          *bvh_slot(stack, stride, sp) = first0 ? r1 : r0;
          sp++;
          ref = first0 ? r0 : r1;
        } else if (h0 | h1) {
          ref = h0 ? r0 : r1;
        } else {
          ref = sp > 0 ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;
        }
      }
    }
    if (ref < 0) {
      const int first = (~ref) >> 3, cnt = (~ref) & 7;
      for (int j = 0; j < cnt; ++j) {
        const int id = (int)((uint32_t)__float_as_int(S.bvh_prims[2 * (size_t)(first + j) + 1].w) >> 3);
        const double h = prim_hit64(S64.prims + 8 * (size_t)id, O, D, S64.bias, 1.7976931348623157e308);
        if (h > 0.0) {  // scene.go:174
          if (h <= S64.bias) {
            if (id < qbest) qbest = id, qt = h;
          } else if (h < tb || (h == tb && id < best)) {
            tb = h, best = id;
          }
        }
      }
      ref = sp > 0 ? *bvh_slot(stack, stride, --sp) : GRT_BVH_DONE;
    }
  }
  if (qbest != 0x7fffffff) {
    t_out = qt;
    return qbest;
  }
  t_out = tb;
  return best;
}
