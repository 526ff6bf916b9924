// grt_bvh8.h — the render kernel's walk over an 8-WIDE hierarchy (round 3; config C5).
//
// Why: the binary walk of grt_bvh.h moves a 128-byte cache line from L2 for every 32-byte node it tests and runs at
// 76-85 % of the rate at which the L2 can serve gathered lines (profiles/r03b_c5_pmc.json: 33.5 lines per ray).  The
// lever is LINES PER RAY.  Here a node is one whole line and holds the boxes of up to EIGHT children on an 8-bit grid
// local to the node, so a walk takes a third of the steps.  The order-free form of Scene.intersect's result
// (grt_bvh.h: the lowest-index object entered within the bias, else the nearest, lowest index among equal distances)
// does not care in which order objects are met, so any conservative cull gives the reference's winner.
//
// Built on the host from the binary tree (grt_api.hip: wide_from_binary): a wide node takes a binary node's two
// children and keeps replacing the inner child of largest surface by ITS two children until it has eight (leaves are
// single objects).  Children sit in slot order along the node's longest axis; a ray visits them in ascending or
// descending slot order by the sign of its direction on that axis (near to far, roughly).
//
// Node (128-byte stride, 80 bytes used):
//   q0 = (o.x, o.y, o.z, s.x)   origin (the node box's lower corner) and the grid's cell size per axis
//   q1 = (s.y, s.z, w0, w1)     w0 = child_base | axis << 22 | n_children << 24; w1 = leaf_base | inner_mask << 24
//   q2 = lo.x[8] lo.y[8]        one byte per child
//   q3 = lo.z[8] hi.x[8]
//   q4 = hi.y[8] hi.z[8]
//   inner child k is wide node child_base + popcount(inner_mask below k); leaf child k is object slot
//   leaf_base + popcount(~inner_mask below k) of the object array in wide-leaf order (records as grt_bvh.h's).
// Child boxes are rounded outward and padded by one cell (1/253 of the node's extent per axis): the slab arithmetic
// below and the object arithmetic of sphere_hit32 / aabb_hit32 round differently, a subtree must never be culled by that.
// A zero direction component needs no separate form: |1/D| is capped (bvh8_ray_setup), so the axis culls like any other;
// where a distance still comes out NaN (inf - inf at absurd coordinates), v_min / v_max drop it: the axis then counts
// as overlapping, which is conservative.
//
// Walk state: the GROUP in hand = a visited node's children still to be entered (w0 with the hit mask in bits 24-31,
// w1), and a stack of such groups in LDS (two words per entry; at most one push per level of the wide tree).
#pragma once

#define GRT_BVH8_NODE_QUADS 8  // uint4 per node (128-byte stride)
#define GRT_BVH8_STACK 12

struct Bvh8Ray {
  f3 invD, OinvD;  // 1 / D, O / D
};
template <class RAY>
GRT_DEV void bvh8_ray_setup(Bvh8Ray& g, const RAY& r) {
  // |1/D| capped at 2^80: a zero (or vanishing) component then still culls on its axis (the ray stays within the padded
  // box of anything it can hit: it moves less than 2^-79 of the scene's reach along that axis, the padding is 2e-6 of it),
  // where +-inf would turn both slab distances into NaN and let the ray visit every node that overlaps on the other axes
  const float cap = 1.2089258196146292e24f;
  g.invD = mk3(fminf(fmaxf(rcp32(r.D.x), -cap), cap), fminf(fmaxf(rcp32(r.D.y), -cap), cap), fminf(fmaxf(rcp32(r.D.z), -cap), cap));
  g.OinvD = r.O * g.invD;
}
struct Bvh8Walk {
  uint32_t g0, g1;  // the group in hand: child_base | axis << 22 | hit mask << 24, leaf_base | inner_mask << 24
  int sp;
  BvhHit H;
};
GRT_DEV void bvh8_walk_begin(Bvh8Walk& w) {
  w.H.tbest = 3.402823466e38f, w.H.best = -1, w.H.qt = 0.0f, w.H.qbest = 0x7fffffff;
  w.sp = 0;
  w.g0 = 1u << 24, w.g1 = 1u << 24;  // a group of one inner child: the root (node 0)
}
GRT_DEV void bvh8_walk_end(Bvh8Walk& w) { w.g0 = 0u, w.sp = 0; }
GRT_DEV bool bvh8_busy(const Bvh8Walk& w) { return (w.g0 >> 24) != 0u; }
// the child the walk enters next: lowest pending slot for a ray that runs up the node's axis, highest otherwise
GRT_DEV int bvh8_next_child(const Bvh8Walk& w, f3 D) {
  const uint32_t hit = w.g0 >> 24, axis = (w.g0 >> 22) & 3u;
  const float d = axis == 0u ? D.x : (axis == 1u ? D.y : D.z);
  return d >= 0.0f ? __builtin_ctz(hit | 0x100u) : 31 - __builtin_clz(hit | 1u);
}
GRT_DEV bool bvh8_at_inner(const Bvh8Walk& w, f3 D) { return bvh8_busy(w) && (((w.g1 >> 24) >> bvh8_next_child(w, D)) & 1u); }
GRT_DEV bool bvh8_at_leaf(const Bvh8Walk& w, f3 D) { return bvh8_busy(w) && !(((w.g1 >> 24) >> bvh8_next_child(w, D)) & 1u); }

// stack entry d of this lane: the wave's [depth][64] blocks of the two words (lane index taken afresh, see bvh_slot)
GRT_DEV void bvh8_pop(Bvh8Walk& w, int* stack) {
  if (w.sp > 0) {
    --w.sp;
    w.g0 = (uint32_t)*bvh_slot(stack, 0, 2 * w.sp);
    w.g1 = (uint32_t)*bvh_slot(stack, 0, 2 * w.sp + 1);
  } else {
    w.g0 = 0u;
  }
}
GRT_DEV float ubyte_f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 255u); }  // v_cvt_f32_ubyteK

// One inner child: leave it in the group's mask no more, keep the rest of the group on the stack, fetch the child's
// node and test its (up to) eight children against the hit so far.
template <class RAY>
GRT_DEV void bvh8_inner_step(const uint4* __restrict__ nodes, const RAY& r, const Bvh8Ray& g, Bvh8Walk& w, int* stack) {
  const int k = bvh8_next_child(w, r.D);
  const uint32_t inner = w.g1 >> 24;
  const uint32_t node = (w.g0 & 0x3fffffu) + (uint32_t)__builtin_popcount(inner & ((1u << k) - 1u));
  w.g0 &= ~(1u << (24 + k));
  if (w.g0 >> 24) {
    *bvh_slot(stack, 0, 2 * w.sp) = (int)w.g0;
    *bvh_slot(stack, 0, 2 * w.sp + 1) = (int)w.g1;
    w.sp++;
  }
  const uint4* __restrict__ n = nodes + (size_t)GRT_BVH8_NODE_QUADS * node;
  const uint4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3], q4 = n[4];
  const f3 o = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
  const f3 s = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
  const f3 a = mk3(fmaf(o.x, g.invD.x, -g.OinvD.x), fmaf(o.y, g.invD.y, -g.OinvD.y), fmaf(o.z, g.invD.z, -g.OinvD.z));
  const f3 sv = s * g.invD;
  const uint32_t nch = (q1.z >> 24) & 15u;
  uint32_t hit = 0u;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint32_t wlx = c < 4 ? q2.x : q2.y, wly = c < 4 ? q2.z : q2.w, wlz = c < 4 ? q3.x : q3.y;
    const uint32_t whx = c < 4 ? q3.z : q3.w, why = c < 4 ? q4.x : q4.y, whz = c < 4 ? q4.z : q4.w;
    const float x0 = fmaf(ubyte_f(wlx, c & 3), sv.x, a.x), x1 = fmaf(ubyte_f(whx, c & 3), sv.x, a.x);
    const float y0 = fmaf(ubyte_f(wly, c & 3), sv.y, a.y), y1 = fmaf(ubyte_f(why, c & 3), sv.y, a.y);
    const float z0 = fmaf(ubyte_f(wlz, c & 3), sv.z, a.z), z1 = fmaf(ubyte_f(whz, c & 3), sv.z, a.z);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    // a hit distance h in (0, tbest] of an object inside the box lies in [tn, tf]
    const bool h = (tn <= tf) & (tf > 0.0f) & (tn <= w.H.tbest);
    hit |= h ? (1u << c) : 0u;
  }
  hit &= (1u << nch) - 1u;
  if (hit) {
    w.g0 = (q1.z & 0xffffffu) | (hit << 24);
    w.g1 = q1.w;
  } else {
    bvh8_pop(w, stack);
  }
}
// One leaf child: its object (the linear loop's own Hit), then the rest of the group or the next group of the stack.
template <class RAY>
GRT_DEV void bvh8_leaf_step(const float4* __restrict__ bprims, const RAY& r, float bias, Bvh8Walk& w, int* stack) {
  const int k = bvh8_next_child(w, r.D);
  const uint32_t leaves = ~(w.g1 >> 24) & 255u;
  const uint32_t slot = (w.g1 & 0xffffffu) + (uint32_t)__builtin_popcount(leaves & ((1u << k) - 1u));
  w.g0 &= ~(1u << (24 + k));
  const float4 pa = bprims[2 * (size_t)slot], pb = bprims[2 * (size_t)slot + 1];
  bvh_object<true>(pa, pb, r, bias, w.H);  // (the careful box form is chosen per lane inside aabb_hit32)
  if (!(w.g0 >> 24)) bvh8_pop(w, stack);
}
GRT_DEV int bvh8_walk_result(const Bvh8Walk& w, float& t_out) {
  if (w.H.qbest != 0x7fffffff) {
    t_out = w.H.qt;
    return w.H.qbest;
  }
  t_out = w.H.tbest;
  return w.H.best;
}
