// grt_gather_group.h — IrradianceEst (photon_mapping.go:125-162) for a GROUP of neighbouring queries per wave.
//
// k_gather_knn (one wave per query) spends a tree walk, ~9 bucket loads, a selection and a reduction per query and
// waits on their dependent loads half of the time.  Neighbouring queries want nearly the same photons.  Here a wave
// takes 64 queries that lie close together (the caller's order, or the Morton order gather_launch sorts them into), one
// query per lane:
//   1. centre c and radius rho of the group; an upper bound B on the k-th distance at c (k-th smallest of the photons
//      in the leaves nearest to c);
//   2. ONE walk collects every photon within P = min(B + rho, radius) + rho of c into an LDS pool (position + slot).
//      The k-th distance is 1-Lipschitz, so r_k(q) <= B + rho for every query of the group and the pool holds every
//      photon within min(r_k(q), radius) of q: it is complete within P - |q - c| of q;
//   3. every lane finds ITS k-th smallest squared distance over the pool: counting passes (three thresholds per pass, a
//      broadcast LDS read per pooled photon, no cross-lane traffic) that bracket the rank-k value by interpolation plus
//      a bit-space midpoint (so a bracket at least halves per pass), then one pass that fetches the value;
//   4. one more pass runs the loop of photon_mapping.go:138-161 per lane; a pooled photon's power and direction
//      bytes are wave-uniform loads.
// A pool that would overflow splits the group in two halves (smaller rho), down to single queries, where the bound
// is tightened from the pool itself as in k_gather_knn.  Same definitions as k_gather_knn and the oracle: every
// photon with d^2 <= the k-th smallest d^2 is accepted, r2 = the largest accepted d^2, found < 8 -> unscaled sum.
// Squared distances use the operation order of k_gather_knn, so both kernels accept the same photons; a lane adds
// them in pool order (k_gather_knn: a wave reduction), so sums agree to FP32 rounding.
//
// STATUS (round 2): exact (tests/test_gpu_photon.py pass with GRT_GATHER_GROUP=1) and MEASURED SLOWER at C4 — global map
// 42 ms against 1.64 ms, caustic map 7.9 against 0.52, photon-mapped render 104 against 41 ms — so it is an opt-in
// experiment, not the default.  Why (counters of GRT_GATHER_STATS=1): at one query per pixel the queries are 0.6 units apart
// and r_k is 7, so a group of 64 has rho ~ 3-5 and its pool P = r_k + 2 rho holds 3-4 x the photons a query accepts (every
// lane scans all of them, 4-5 passes); Morton runs of 64 are less compact than that, 18 groups per 64 queries after
// splitting; a group pays a whole k-nearest search at its centre (40 us at 3 waves per SIMD) before it serves anyone;
// and where photons are far denser than queries (the caustic under the glass sphere) every group splits down to single
// queries, which ONE wave then serves one after the other (64 searches in a row: the launch waits for that wave).
#pragma once
// (included from grt_photon.hip after k_gather_knn's helpers: KdView, P3, select_kth, wave_total_*, mbcnt64 ...)

#define GG_SLACK 1.0001f  // relative slack on squared radii: covers the FP32 rounding of distances and of the triangle inequality

template <int CAP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gather_group(KdView T, const float* __restrict__ qpos, const float* __restrict__ qnrm,
                                                             const int* __restrict__ perm, int n_q, float radius, int k,
                                                             float* __restrict__ irrad, float* __restrict__ rk2_out,
                                                             unsigned long long* __restrict__ stats) {
  constexpr int RP = CAP / 64;
  __shared__ float4 pool_all[WAVES][CAP];  // x, y, z, photon slot (int bits)
  __shared__ __attribute__((aligned(16))) int hist_all[WAVES][GATHER_BINS];
  __shared__ unsigned long long todo_all[WAVES][8];
  __shared__ float lut[4 * 256];  // costheta, sintheta, cosphi, sinphi (photon_mapping.go:90-93)
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) lut[i] = T.lut[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = sgpr(threadIdx.x >> 6);
  float4* pool = pool_all[wave];
  int* hist = hist_all[wave];
  unsigned long long* todo = todo_all[wave];
  const int slot = (blockIdx.x * WAVES + wave) * 64 + lane;
  const bool in_range = slot < n_q;
  const int qi = in_range ? (perm ? perm[slot] : slot) : 0;
  float qx = 0.0f, qy = 0.0f, qz = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f;
  if (in_range) {
    qx = qpos[3 * (size_t)qi], qy = qpos[3 * (size_t)qi + 1], qz = qpos[3 * (size_t)qi + 2];
    nx = qnrm[3 * (size_t)qi], ny = qnrm[3 * (size_t)qi + 1], nz = qnrm[3 * (size_t)qi + 2];
  }
  const bool valid = in_range & (qx == qx);  // NaN position = no query in this slot (photon-mapped render)
  if (in_range & !valid) {
    irrad[3 * (size_t)qi] = 0.0f, irrad[3 * (size_t)qi + 1] = 0.0f, irrad[3 * (size_t)qi + 2] = 0.0f;
    if (rk2_out) rk2_out[qi] = -1.0f;
  }
  const float radius2 = radius * radius;
  const int last = T.levels - 1;
  int st_groups = 0, st_pool = 0, st_passes = 0, st_splits = 0;
  float st_rho = 0.0f, st_B = 0.0f, st_P = 0.0f;
  int st_nodes = 0, st_leaves = 0, st_tight = 0;

  int sp = 0;
  {
    const unsigned long long all = __builtin_amdgcn_ballot_w64(valid);
    if (all) {
      if (lane == 0) todo[0] = all;
      sp = 1;
    }
  }
  while (sp > 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned long long Mv = todo[--sp];
    const unsigned long long M = ((unsigned long long)(uint32_t)sgpr((int)(Mv >> 32)) << 32) | (uint32_t)sgpr((int)(uint32_t)Mv);
    const bool in = (M >> lane) & 1ull;
    const int cnt = popc64(M);
    st_groups++;

    // ---- 1. centre, radius of the group, bound on the k-th distance at the centre ----
    const float inv_cnt = 1.0f / (float)cnt;
    const float cx = wave_total_add_f(in ? qx : 0.0f) * inv_cnt, cy = wave_total_add_f(in ? qy : 0.0f) * inv_cnt,
                cz = wave_total_add_f(in ? qz : 0.0f) * inv_cnt;
    float rho_l;  // |q - c| of this lane's query
    {
      const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
      rho_l = in ? sqrt32(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) : 0.0f;
    }
    const float rho = __uint_as_float(wave_total_max_u(__float_as_uint(rho_l))) * GG_SLACK;
    // photons beyond radius + rho of c are beyond the radius of every query of the group
    const float prune2 = radius > 0.0f ? (radius + rho) * (radius + rho) * GG_SLACK : INFINITY;

    auto test_group = [&](int lvl, int off, int prefix, float bnd, float& d2box) -> unsigned long long {
      const int nb = lvl == 0 ? T.b0 : KD_FAN_BITS;
      const int base = off + (prefix << nb);
      const bool act = lane < (1 << nb);
      const int rec = base + (act ? lane : 0);
      const float4 lo = T.boxes[rec], hi = T.boxes[T.n_boxes + rec];
      const float dx = fmaxf(fmaxf(lo.x - cx, cx - hi.x), 0.0f);
      const float dy = fmaxf(fmaxf(lo.y - cy, cy - hi.y), 0.0f);
      const float dz = fmaxf(fmaxf(lo.z - cz, cz - hi.z), 0.0f);
      d2box = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      st_nodes++;
      return __builtin_amdgcn_ballot_w64(act & (d2box <= bnd));
    };
    int count = 0;
    // one leaf bucket: photons within bnd of c go to the pool (KEEP_POS: with their position and slot, else only d^2)
    auto append_leaf = [&](int leaf, float bnd) {
      const int a = (int)(((long long)leaf * (long long)T.n) >> T.depth);
      const int b = (int)(((long long)(leaf + 1) * (long long)T.n) >> T.depth);
      const int i = a + lane;
      const bool ok = i < b;
      const P3 p = *(const P3*)(T.pos3 + 3 * (size_t)(ok ? i : a));  // one 12-byte load per lane
      const float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
      const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      const bool pass = ok & (d2 <= bnd);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
      const int s = count + mbcnt64(m);
      if (pass) pool[s] = make_float4(p.x, p.y, p.z, __int_as_float(i));
      count += popc64(m);
      st_leaves++;
    };

    // A depth-first walk over the boxes within bound2 of c; photons within bound2 go to the pool.
    //   SEARCH: the k-nearest search of k_gather_knn around c — nearest box first while the bound is loose, the bound
    //   tightened to the k-th smallest of the pool as soon as k photons are in it and whenever the pool is full (the
    //   pool is then cut down to what lies within); ends with bound2 = the k-th squared distance at c.
    //   !SEARCH: a fixed bound, boxes in index order; a full pool reports overflow.
    float bound2 = prune2;
    bool tight = false;  // bound2 encloses k photons
    bool overflow = false;
    auto tighten = [&]() {  // bound2 = the k-th smallest squared distance of the pool (count >= k); the pool keeps what lies within
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      uint32_t vb[RP];
      float4 e[RP];
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const int j = lane + 64 * r;
        e[r] = pool[j < count ? j : 0];
        const float dx = e[r].x - cx, dy = e[r].y - cy, dz = e[r].z - cz;
        vb[r] = j < count ? __float_as_uint(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) : 0xffffffffu;
      }
      const uint32_t Tk = select_kth<RP>(vb, count, k, lane, hist);
      bound2 = __uint_as_float(Tk);
      tight = true;
      st_tight++;
      int nc = 0;
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const bool keep = vb[r] <= Tk;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        const int s = nc + mbcnt64(m);
        if (keep) pool[s] = e[r];
        nc += popc64(m);
      }
      count = nc;
    };
    auto pick_child = [&](unsigned long long pending, float d2b) -> int {
      if (tight) return __builtin_ctzll(pending);
      const bool inp = (pending >> lane) & 1ull;
      const uint32_t dmin = wave_total_min_u(inp ? __float_as_uint(d2b) : 0xffffffffu);
      return __builtin_ctzll(__builtin_amdgcn_ballot_w64(inp & (__float_as_uint(d2b) == dmin)));
    };
    auto walk = [&](bool search) {
      int stk_lo = 0, stk_hi = 0;  // pending-children masks, level l in lane l
      int lvl = 0, off = 0, nl = 1 << T.b0, prefix = 0;
      float d2box;
      unsigned long long mask = test_group(0, 0, 0, bound2, d2box);
      for (;;) {
        if (lvl == last) {  // the children are leaf buckets
          const int pb = prefix << (lvl == 0 ? T.b0 : KD_FAN_BITS);
          while (mask) {
            const int l = pick_child(mask, d2box);
            mask &= ~(1ull << l);
            if (count > CAP - 64) {
              if (search && count >= k) tighten();
              if (count > CAP - 64) {
                overflow = true;
                return;
              }
            }
            append_leaf(pb | l, bound2);
            if (search && !tight && count >= k) tighten();
          }
        }
        if (mask == 0ull) {  // back to the parent's pending children
          if (lvl == 0) return;
          lvl--;
          nl >>= KD_FAN_BITS;
          off -= nl;
          prefix >>= (lvl == 0 ? T.b0 : KD_FAN_BITS);
          mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(stk_hi, lvl) << 32) |
                 (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(stk_lo, lvl);
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
        mask = test_group(lvl, off, prefix, bound2, d2box);
      }
    };

    float B2 = INFINITY;  // the k-th squared distance at c (INFINITY: fewer than k photons within reach)
    bool exact_bound = false;  // single query: the pool is its own search result and bound2 its k-th squared distance
    if (T.n > 0) {
      if (bound2 == INFINITY) {
        // No bound yet: nearest child first down to the nearest leaves until k photons are seen; their k-th smallest
        // distance bounds the k-th.  The pool is then emptied and the search starts over with that bound.
        int off = 0, nl = 1 << T.b0, prefix = 0;
        for (int lvl = 0; lvl <= last; ++lvl) {
          float d2box;
          unsigned long long rem = test_group(lvl, off, prefix, bound2, d2box);
          const int nb = lvl == 0 ? T.b0 : KD_FAN_BITS;
          if (rem == 0ull) break;
          while (rem) {
            const int l = pick_child(rem, d2box);
            rem &= ~(1ull << l);
            if (lvl < last) {
              prefix = (prefix << nb) | l;
              break;
            }
            append_leaf((prefix << nb) | l, bound2);
            if (count >= k) break;
          }
          off += nl;
          nl <<= KD_FAN_BITS;
        }
        if (count >= k) tighten();
        count = 0;
      }
      walk(true);
      if (!overflow && count >= k) {
        tighten();
        B2 = bound2;
      }
    }
    float P2 = bound2;
    if (cnt == 1) {
      exact_bound = B2 < INFINITY;  // (rho = 0: what the search kept is all the query can accept)
    } else if (T.n > 0 && !overflow) {
      // ---- 2. the pool: every photon within P of c ----
      float reach = sqrt32(B2) * GG_SLACK + rho;          // >= r_k(q) for every query of the group
      if (radius > 0.0f) reach = fminf(reach, radius);    // photons beyond the radius never count (:141)
      const float P = (reach + rho) * GG_SLACK;
      P2 = bound2 = P * P;
      tight = true;
      count = 0;
      walk(false);
    }
    st_rho += rho, st_B += B2 < INFINITY ? sqrt32(B2) : 0.0f, st_P += P2 < INFINITY ? sqrt32(P2) : 1000.0f;
    if (overflow && cnt > 1) {  // two halves, each with a smaller rho (the queries are in spatial order)
      const int rank = mbcnt64(M);
      const unsigned long long lower = __builtin_amdgcn_ballot_w64(in & (rank < (cnt + 1) / 2));
      if (lane == 0) todo[sp] = M & ~lower, todo[sp + 1] = lower;
      sp += 2;
      st_splits++;
      continue;
    }
    // (overflow with cnt == 1: more photons tied at the k-th distance than the pool holds; the pool so far serves)
    // pad to a multiple of 4 entries with photons at infinity: the passes below read four entries per step
    {
      const int padded = (count + 3) & ~3;
      if (lane < padded - count) pool[count + lane] = make_float4(INFINITY, 0.0f, 0.0f, __int_as_float(0));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    st_pool += count;

    // ---- 3. per lane: the k-th smallest squared distance over the pool ----
    // invariant: #(bits < lo) = cl < k <= ch = #(bits < hi); the pool is complete within comp2 of the lane's query
    float comp2;
    {
      const float comp = fmaxf(sqrt32(P2) * (1.0f / (GG_SLACK * GG_SLACK)) - rho_l, 0.0f);
      comp2 = exact_bound ? P2 : comp * comp;
    }
    uint32_t lo = 0u, hi = min(__float_as_uint(comp2), 0x7f7fffffu) + 1u;
    int cl = 0, ch = 0;
    uint32_t Tk = 0x7f800000u;  // fewer than k photons in reach: all of them
    bool open = in;             // still bisecting
    bool bounded = false;       // the k-th distance exists inside the complete region
    for (int pass = 0;; ++pass) {
      uint32_t t1 = 0u, t2 = 0u, t3 = 0u;
      if (open) {
        if (pass == 0) {  // first guesses around the bound at the centre; the third threshold counts the complete region
          const float est = B2 < INFINITY ? B2 : 0.5f * comp2;
          t3 = hi;
          t1 = min(max(__float_as_uint(0.7f * est), 1u), hi);
          t2 = min(max(__float_as_uint(1.1f * est), t1), hi);
        } else {
          const int m = ch - cl, j = k - cl;
          const float lf = __uint_as_float(lo), hf = __uint_as_float(hi - 1u);
          const float per = (hf - lf) / (float)m;
          const float mid = fmaf(per, (float)j - 0.5f, lf);
          const float w = per * (1.0f + 1.5f * sqrt32((float)min(j, m - j + 1)));
          const uint32_t a = min(max(__float_as_uint(fmaxf(mid - w, 0.0f)), lo + 1u), hi - 1u);
          const uint32_t b = min(max(__float_as_uint(fmaxf(mid + w, 0.0f)), lo + 1u), hi - 1u);
          const uint32_t c = lo + ((hi - lo) >> 1);  // bit-space midpoint: the bracket at least halves
          t1 = min(a, c), t3 = max(b, c);
          t2 = max(a, min(b, c));  // the median (a <= b)
        }
      }
      if (!__builtin_amdgcn_ballot_w64(open)) break;
      st_passes++;
      int c1 = 0, c2 = 0, c3 = 0;
      for (int i = 0; i < count; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 e = pool[i + u];  // same address in every lane: a broadcast
          const float dx = e.x - qx, dy = e.y - qy, dz = e.z - qz;
          const uint32_t b = __float_as_uint(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
          c1 += b < t1, c2 += b < t2, c3 += b < t3;
        }
      }
      if (open) {
        if (pass == 0 && c3 < k) {
          open = false;  // fewer than k photons where the pool is complete: beyond the radius, or the whole map
        } else {
          bounded = true;
          if (c1 >= k) hi = t1, ch = c1;
          else if (c2 >= k) lo = t1, cl = c1, hi = t2, ch = c2;
          else if (c3 >= k) lo = t2, cl = c2, hi = t3, ch = c3;
          else lo = t3, cl = c3;
          if (ch - cl == 1 || hi - lo <= 1u) open = false;  // one value left, or one float: the fetch below finds it
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(bounded)) {  // the smallest value >= lo is the k-th smallest
      uint32_t v = 0xffffffffu;
      for (int i = 0; i < count; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 e = pool[i + u];
          const float dx = e.x - qx, dy = e.y - qy, dz = e.z - qz;
          const uint32_t b = __float_as_uint(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
          v = min(v, b >= lo ? b : 0xffffffffu);
        }
      }
      if (bounded) Tk = v;
    }

    // ---- 4. the loop of photon_mapping.go:138-161, one query per lane, pooled photons in pool order ----
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    uint32_t r2 = 0u;
    int found = 0, reach_cnt = 0;
    // 64 pooled photons at a time: every lane fetches ONE photon's power and direction (a vector gather, its latency paid
    // once per 64 photons), then the photons are handed round with v_readlane
    for (int base = 0; base < count; base += 64) {
      const int j = base + lane;
      const float4 e = pool[j < count ? j : 0];
      const float4 pw = T.pwb[__float_as_int(e.w)];
      const int bits = __float_as_int(pw.w);
      const int th = (bits >> 8) & 255, ph = (bits >> 16) & 255;
      const float st = lut[256 + th];
      const float ddx = st * lut[512 + ph], ddy = st * lut[768 + ph], ddz = lut[th];  // PhotonDir, :117-122
      const int nb = min(64, count - base);  // wave-uniform
      for (int t = 0; t < nb; ++t) {
        auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t)); };
        const float dx = bc(e.x) - qx, dy = bc(e.y) - qy, dz = bc(e.z) - qz;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const uint32_t b = __float_as_uint(d2);
        reach_cnt += (radius == 0.0f) | (d2 <= radius2);
        if ((b <= Tk) & ((radius == 0.0f) | (d2 < radius2)) & (bc(ddx) * nx + bc(ddy) * ny + bc(ddz) * nz < 0.0f)) {  // :141, :144
          sx += bc(pw.x), sy += bc(pw.y), sz += bc(pw.z);
          r2 = max(r2, b);
          found++;
        }
      }
    }
    if (in) {
      if (found >= 8) {  // :154-161
        const float tmp = rcp32(PI_F * __uint_as_float(r2));
        sx *= tmp, sy *= tmp, sz *= tmp;
      }
      irrad[3 * (size_t)qi] = sx, irrad[3 * (size_t)qi + 1] = sy, irrad[3 * (size_t)qi + 2] = sz;
      if (rk2_out) {
        const bool kth_known = bounded && (radius == 0.0f || __uint_as_float(Tk) <= radius2);
        rk2_out[qi] = kth_known ? __uint_as_float(Tk) : (reach_cnt > 0 ? INFINITY : -1.0f);
      }
    }
  }
  if (stats && lane == 0) {
    atomicAdd(stats + 0, (unsigned long long)st_groups);
    atomicAdd(stats + 1, (unsigned long long)st_pool);
    atomicAdd(stats + 2, (unsigned long long)st_passes);
    atomicAdd(stats + 3, (unsigned long long)st_splits);
    atomicAdd(stats + 7, (unsigned long long)st_nodes), atomicAdd(stats + 8, (unsigned long long)st_leaves), atomicAdd(stats + 9, (unsigned long long)st_tight);
    atomicAdd(stats + 4, (unsigned long long)(st_rho * 1000.0f)), atomicAdd(stats + 5, (unsigned long long)(st_B * 1000.0f)), atomicAdd(stats + 6, (unsigned long long)(st_P * 1000.0f));
  }
}
