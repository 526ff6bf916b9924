// grt_kernels.h — the render kernels (gfx950).  Included once, by grt_api.hip.
//
// k_pathtrace     Scene.Render's per-pixel sample loop (scene.go:86-104) with trace()
//                 (scene.go:237-349) unrolled into a per-lane state machine
// k_sum_chunks    fixed-order reduction of the per-chunk partial sums
// k_tonemap       mean -> Gamma(2) -> Clamp -> uint8(255.99 c), B,G,R  (scene.go:98-99,67-74)
// k_trace_samples per-sample radiance of chosen pixels (parity aid)
// k_primary_hits  Camera.Ray + intersect + Surface per pixel, FP32 with FP64 re-test (C2)
// k_primary_sweep FP32 intersection roofline loop (C2 roofline variant)
#pragma once
#include "grt_device.h"
#include "grt_parity.h"

// Work decomposition of one shard of a frame.  A wave owns one 8x8 pixel tile for one chunk
// of the sample range; tile rows coincide with the GRT_BAND_ROWS-row bands that are dealt
// round-robin to shards, so a shard's tiles are whole.
#define GRT_MAX_CHUNKS 63
struct RenderJob {
  int shard, n_shards;  // this shard renders bands shard, shard+n_shards, ...
  int rows_local;       // rows this shard owns (packed, ascending)
  int tiles_x, n_tiles; // 8x8 tiles across, and in total for this shard
  int n_chunks, chunk_spp, spp;  // chunk_spp: the size of a full chunk (explicit choice), or of the largest one (automatic schedule)
  int s_first;          // first sample index (sample-range sharding; 0 otherwise)
  // chunk c renders samples s_first + [chunk_start[c], chunk_start[c+1]).  Equal chunks for an explicit chunk_spp; the
  // automatic schedule ends with ever smaller chunks so that the last wave-items of a launch are short (grt_api.hip)
  int chunk_start[GRT_MAX_CHUNKS + 1];
  PmOut pm;             // photon-mapped trace(): gather requests of this launch (pos == NULL: off)
  int pm_batch;         // samples per pixel the request arrays hold
  int shade_min;        // hierarchy scenes: lanes that must wait for shading before the wave shades (64: every level in lockstep)
  int inner_weight;     // hierarchy scenes: an inner-node step runs when inner lanes * this >= leaf lanes * 4
  int inner_again;      // hierarchy scenes: another inner-node step follows at once while this many lanes still hold one ...
  int inner_run;        // ... up to this many steps per scheduling round
  int no_octants;       // A/B: scenes of <= 8 objects through the min/max slab test instead of the direction-octant tables
  double acc_scale, acc_inv;  // k_pathtrace_bvh's fixed-point pixel sums: 2^e and 2^-e (make_job)
};

GRT_DEV unsigned long long wave_sum_u32(unsigned v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  return s;
}

// Scene tables for the per-lane record gathers of path_advance: objects and materials are
// copied to LDS once per workgroup (NP > 0) and the objects are also loaded into registers
// for the unrolled intersect loop.
#define GRT_FAST_PRIMS 8
// Mid-size scenes (9 .. GRT_LDS_PRIMS objects and materials, below the hierarchy's crossover): the
// render kernel keeps the reference's linear loop but reads the records from LDS copies (every lane
// the same address: a broadcast) instead of waiting for a scalar load per object — one copy in index
// order for the hit's Surface and material, one sorted by kind so that the loop has no kind branch.
#define GRT_LDS_PRIMS 224

template <int NP>
struct SceneTables {
  const float4* tp;
  const float4* tm;
  const float4* ts = nullptr;  // mid-size scenes: the objects sorted by kind (intersect_sorted)
  int n_boxes = 0;
  const float* oct = nullptr;  // NP > 0: the 8 direction-octant tables (intersect_lds_oct)
};

// The host pads the staged arrays to GRT_FAST_PRIMS records, so the copies are unconditional.
template <int NP, bool LDSGEN = false>
GRT_DEV void load_tables(const KScene& S, float4* lds, SceneTables<NP>& T, float* oct = nullptr) {
  if (NP == 0 && LDSGEN) {
    // [0, 2N) objects in index order (Surface and material of the hit), [2N, 5N) materials,
    // [5N, 7N) objects sorted by kind: a stable partition by the first wave (ballot + mbcnt)
    for (int i = threadIdx.x; i < 2 * S.n_prims; i += blockDim.x) lds[i] = S.prims[i];
    for (int i = threadIdx.x; i < GRT_MAT_STRIDE * S.n_mats; i += blockDim.x) lds[2 * GRT_LDS_PRIMS + i] = S.mats[i];
    float4* sorted = lds + (2 + GRT_MAT_STRIDE) * GRT_LDS_PRIMS;
    int n_boxes = 0;
    for (int base = 0; base < S.n_prims; base += 64) {  // every wave counts (uniform), the first one writes
      const int i = base + (threadIdx.x & 63);
      n_boxes += __builtin_popcountll(__ballot(i < S.n_prims && (__float_as_int(S.prims[2 * min(i, S.n_prims - 1) + 1].w) & 1)));
    }
    if (threadIdx.x < 64) {
      int bpos = 0, spos = n_boxes;
      for (int base = 0; base < S.n_prims; base += 64) {
        const int i = base + threadIdx.x;
        const bool in = i < S.n_prims;
        const float4 pa = S.prims[2 * (in ? i : 0)];
        float4 pb = S.prims[2 * (in ? i : 0) + 1];
        const int bits = __float_as_int(pb.w);
        const unsigned long long mb = __ballot(in && (bits & 1)), ms = __ballot(in && !(bits & 1));
        const int lane_rank_b = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0));
        const int lane_rank_s = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0));
        if (in) {
          const int slot = (bits & 1) ? bpos + lane_rank_b : spos + lane_rank_s;
          pb.w = __int_as_float((int)((uint32_t)i << 3 | ((uint32_t)bits & 7u)));
          sorted[2 * slot] = pa, sorted[2 * slot + 1] = pb;
        }
        bpos += __builtin_popcountll(mb), spos += __builtin_popcountll(ms);
      }
    }
    __syncthreads();
    T.tp = lds;
    T.tm = lds + 2 * GRT_LDS_PRIMS;
    T.ts = sorted;
    T.n_boxes = n_boxes;
  } else if (NP > 0) {
    // [0, 2NP) objects, [2NP, 5NP) materials (a scene never has more materials than objects
    // after de-duplication; the host checks n_mats <= NP before choosing this kernel)
    for (int i = threadIdx.x; i < GRT_MAT_STRIDE * NP; i += blockDim.x) {
      if (i < 2 * NP) lds[i] = S.prims[i];
      lds[2 * NP + i] = S.mats[i];
    }
    if (oct)
      for (int i = threadIdx.x; i < 8 * NP; i += blockDim.x)
        octant_record(S.prims[2 * (i % NP)], S.prims[2 * (i % NP) + 1], i / NP, oct + (i / NP) * GRT_OCT_STRIDE(NP) + 8 * (i % NP));
    __syncthreads();
    T.tp = lds;
    T.tm = lds + 2 * NP;
    T.oct = oct;
  } else {
    T.tp = S.prims;
    T.tm = S.mats;
  }
}

// One wave = one (tile, chunk) work item; one lane = one pixel.  Each lane runs its pixel's
// samples back to back: when its path ends it adds the radiance and starts the next sample
// in the same loop iteration, so every iteration of the wave-wide loop does one intersect()
// for every lane that still has samples left — terminated paths never leave idle lanes
// behind (the wave only drains when a lane runs out of samples; __any() is the ballot).
#ifndef GRT_PT_WAVES
#define GRT_PT_WAVES 8
#endif
template <int NP, bool RELOAD = false, bool PREFETCH = false, bool QUEUE = false, bool VIEW = false>
// The production instance for scenes of <= 8 objects is compiled for 8 waves per SIMD (64 VGPRs).  Round 3, same box, built
// without SLP vectorisation (csrc/Makefile): C3 at 1920x1080 7.15 ms at 7 waves, 6.93 ms at 8, 8.03 ms at 6 (with SLP: 7.70 /
// 7.62 / 8.03); rounds 1-2 ran it at 7 (37.5 ms against 38.1 ms at 6 for the full frame, 8 the same).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NP == GRT_FAST_PRIMS && RELOAD && !PREFETCH && !QUEUE) ? GRT_PT_WAVES : 4)))
void k_pathtrace(KScene S, RenderJob J, float* __restrict__ out,
                                                   unsigned long long* __restrict__ ray_count) {
  __shared__ float4 lds[NP > 0 ? (2 + GRT_MAT_STRIDE) * NP : (NP == 0 && RELOAD ? (4 + GRT_MAT_STRIDE) * GRT_LDS_PRIMS : 1)];
  __shared__ int bvh_stack[NP < 0 ? GRT_BVH_STACK * 256 : 1];  // [depth][thread]
  __shared__ float qacc[QUEUE ? 4 * 192 : 1];                  // wave-queue mode: [wave][pixel][rgb]
  __shared__ uint32_t qkey[QUEUE ? 4 * 64 : 1];
  constexpr bool OCT = NP > 0 && RELOAD && !PREFETCH && !QUEUE;  // the production instance for <= 8 objects
  __shared__ __attribute__((aligned(16))) float oct_tables[OCT ? 8 * GRT_OCT_STRIDE(NP > 0 ? NP : 1) : 4];
  SceneTables<NP> T;
  load_tables<NP, NP == 0 && RELOAD>(S, lds, T, OCT && !J.no_octants ? oct_tables : nullptr);
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= J.n_tiles * J.n_chunks) return;  // wave-uniform
  const int tile = item % J.n_tiles, chunk = item / J.n_tiles;
  const int tx = tile % J.tiles_x, lband = tile / J.tiles_x;
  const int x = tx * 8 + (lane & 7);
  const int y = (J.shard + lband * J.n_shards) * 8 + (lane >> 3);
  const int lrow = lband * 8 + (lane >> 3);
  const bool valid = (x < S.W) & (y < S.H);

  const int s0 = J.s_first + J.chunk_start[chunk];
  const int s_end = J.s_first + J.chunk_start[chunk + 1];
  uint32_t pix_key = grt_rng_pixel_key(S.seed32, (uint32_t)y * (uint32_t)S.W + (uint32_t)x);
  float fx = (float)x, fy = (float)y;

  int n_rays = 0;
  bool alive = valid & (s0 < s_end);
  // A tile whose rays all pass the scene's bounding box by: every sample is a miss (black,
  // scene.go:244-250) and no intersect() call is made for it.
  const bool culled = S.tile_live && !S.tile_live[(J.shard + lband * J.n_shards) * J.tiles_x + tx];  // wave-uniform
  if (culled) alive = false;
  LaneState L;
  lane_begin(S, fx, fy, pix_key, s0, L);
  f3 result;
  if (QUEUE) {
    // The wave's (pixel, sample) pairs are one queue (see WaveQueue).  Lanes of pixels outside
    // the frame (edge tiles) take part: their paths are traced and dropped.
    const int w = threadIdx.x >> 6;
    WaveQueue Q;
    Q.acc = qacc + 192 * w, Q.pixkey = qkey + 64 * w;
    Q.acc[3 * lane] = 0.0f, Q.acc[3 * lane + 1] = 0.0f, Q.acc[3 * lane + 2] = 0.0f;
    qkey[64 * w + lane] = pix_key;
    Q.s0 = s0, Q.x0 = tx * 8, Q.y0 = (J.shard + lband * J.n_shards) * 8;
    Q.next = 64, Q.qend = culled ? 0 : 64 * (s_end - s0);
    L.pix = lane;
    alive = (s0 < s_end) & !culled;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (__any(alive)) {
      if (alive)
        path_advance<NP, RELOAD, PREFETCH, true>(S, T.tp, T.tm, fx, fy, pix_key, s_end, L, alive, n_rays, bvh_stack + threadIdx.x, &Q);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    result = mk3(Q.acc[3 * lane], Q.acc[3 * lane + 1], Q.acc[3 * lane + 2]);
  } else {
    const int pm_base = (lrow * S.W + x) * J.pm_batch - J.s_first;  // request slot of sample s: pm_base + s
#ifdef GRT_COUNT_LEVELS
    int levels = 0;
#endif
    while (__any(alive)) {
#ifdef GRT_COUNT_LEVELS
      levels++;
#endif
      if (alive)
        path_advance<NP, RELOAD, PREFETCH, false, VIEW>(S, T.tp, T.tm, fx, fy, pix_key, s_end, L, alive, n_rays, bvh_stack + threadIdx.x,
                                                        nullptr, &J.pm, pm_base, T.ts, T.n_boxes, T.oct);
    }
    result = L.acc;
#ifdef GRT_COUNT_LEVELS
    if (ray_count && lane == 0) atomicAdd(ray_count + 1, 64ull * (unsigned long long)levels);
#endif
  }
  if (valid) {
    float* o = out + (((size_t)chunk * (size_t)J.rows_local + (size_t)lrow) * (size_t)S.W + (size_t)x) * 3;
    o[0] = result.x;
    o[1] = result.y;
    o[2] = result.z;
  }
  if (ray_count) {
    unsigned long long tot = wave_sum_u32((unsigned)n_rays);
    if (lane == 0) atomicAdd(ray_count, tot);
  }
}

// k_pathtrace for hierarchy scenes (grt_bvh.h).  Same work items (one wave = one 8x8 tile x one
// chunk of samples, one lane = one pixel) and the same per-sample RNG streams, but the level is
// not run in lockstep: walks differ a lot in length (C5: 20 % of the lanes were active when every
// lane waited for the wave's longest walk before shading).  Every lane is ready for one of three
// steps (inner node, leaf, shading) and the wave runs, round by round, the step most lanes are
// ready for; the others keep their state — reference in hand, stack, hit so far —
// and wait.  Which lanes meet in a round depends only on the wave's own paths: the frame is
// reproducible and independent of sharding and of the thresholds.  The kernel is bound by memory
// latency (dependent node fetches), so registers are spent sparingly — the walk keeps the ray
// only in grid form (BvhRay) plus O and D — to hold 7 waves per SIMD.
// COUNT (profiling build; ray_count must then hold 16 words): ray_count[1] += inner nodes visited
// (two box tests, 32 B each), ray_count[2] += objects tested (32 B each), [3..10] the schedule's step and lane counts.
// DIFFUSE: the scene holds plain-diffuse objects: their shadow rays (and, with photon maps, their gather
// requests) go through path_shade's hierarchy branch; without them that code is compiled out.
// QUEUE (production): the wave's (pixel, sample) pairs are ONE queue — pair q is sample s0 + q / 64 of pixel q % 64 — and
// a lane whose path has ended takes the next pair, whichever pixel it belongs to.  With lane = pixel the lanes of a
// tile differ systematically in work (at C5 a pixel's primary hit decides how deep its paths go: the lanes of a tile
// average 62 % of the busiest lane's work however long the chunk, tools/exp_c5_map.py), and 40 % of a wave's lanes sat
// without work.  A path's radiance goes to the lane that owns its pixel (lane p holds pixel p's sum in registers):
// v_readlane broadcasts of the few non-black paths of a shading step, added in FIXED POINT (J.acc_scale, a power of
// two chosen from the scene's brightest source so that a chunk cannot overflow 63 bits).  Integer addition is exact,
// so the sum does not depend on the order in which paths end: the frame stays reproducible and independent of sharding
// and of the scheduling thresholds.  Pairs of pixels outside the frame (edge tiles) are born finished: no ray.
template <int WAVES, int STACK, bool COUNT = false, bool DIFFUSE = false, bool QUEUE = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_pathtrace_bvh(
    KScene S, RenderJob J, float* __restrict__ out, unsigned long long* __restrict__ ray_count) {
  __shared__ int bvh_stack[STACK * 256];  // [wave][depth][lane]
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= J.n_tiles * J.n_chunks) return;  // wave-uniform
  const int tile = item % J.n_tiles, chunk = item / J.n_tiles;
  const int tx = tile % J.tiles_x, lband = tile / J.tiles_x;
  const int x = tx * 8 + (lane & 7);
  const int y = (J.shard + lband * J.n_shards) * 8 + (lane >> 3);
  const int lrow = lband * 8 + (lane >> 3);
  const bool valid = (x < S.W) & (y < S.H);
  const int s0 = J.s_first + J.chunk_start[chunk];
  const int s_end = J.s_first + J.chunk_start[chunk + 1];
  uint32_t pix_key = grt_rng_pixel_key(S.seed32, (uint32_t)y * (uint32_t)S.W + (uint32_t)x);
  float fx = (float)x, fy = (float)y;
  int n_rays = 0;
  const bool culled = S.tile_live && !S.tile_live[(J.shard + lband * J.n_shards) * J.tiles_x + tx];  // wave-uniform
  bool alive = (QUEUE | valid) & (s0 < s_end) & !culled;
  LaneState L;
  lane_begin(S, fx, fy, pix_key, s0, L);
  L.pix = lane;
  WaveQueue Q;
  Q.acc = nullptr, Q.pixkey = nullptr;
  Q.s0 = s0, Q.x0 = tx * 8, Q.y0 = (J.shard + lband * J.n_shards) * 8;
  Q.next = 64, Q.qend = 64 * (s_end - s0);
  long long sum_x = 0, sum_y = 0, sum_z = 0;  // QUEUE: this lane's PIXEL, fixed point
  UnitRay r;
  BvhRay g;
  BvhWalk wk;
  r.O = L.O, r.D = L.D;
  bvh_ray_setup(g, r, S.bvh_grid);
  bvh_walk_begin(wk);
  if (!alive) wk.ref = GRT_BVH_DONE;
  if (QUEUE && !valid) wk.ref = GRT_BVH_DONE;  // born finished: it reaches shading as a miss and takes the next pair
  int* stack = bvh_stack + STACK * 64 * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // the wave's [depth][lane] block
  const int pm_base = DIFFUSE ? (lrow * S.W + x) * J.pm_batch - J.s_first : 0;  // request slot of sample s: pm_base + s
  unsigned long long zmask = __ballot(alive & ray_has_zero(r));  // lanes whose ray needs the careful box form
  int n_inner = 0, n_objs = 0;
  unsigned c_steps[3] = {0, 0, 0}, c_lanes[3] = {0, 0, 0}, c_alive = 0, c_rounds = 0;  // COUNT: wave-uniform
#ifdef GRT_COUNT_DEPTH
  unsigned c_depth[4] = {0, 0, 0, 0};  // per lane
#endif
  for (;;) {
    // Three kinds of step — an inner node, a leaf, a trace() level's shading — and every lane is
    // ready for exactly one of them (or has no samples left).  The wave runs the step most lanes
    // are ready for; lanes ready for another keep their state and wait.
    const bool at_inner = bvh_at_inner(wk), at_leaf = bvh_at_leaf(wk);
    const bool to_shade = alive & !at_inner & !at_leaf;
    const unsigned long long mI = __ballot(at_inner), mL = __ballot(at_leaf), mS = __ballot(to_shade);
    const int nI = __builtin_popcountll(mI), nL = __builtin_popcountll(mL), nS = __builtin_popcountll(mS);
    if ((nI | nL | nS) == 0) break;
    if (COUNT) c_rounds++, c_alive += nI + nL + nS;
    if (nS >= J.shade_min || (nI | nL) == 0) {
      if (COUNT) c_steps[2]++, c_lanes[2] += nS;
      int owner = 0;
      f3 give = mk3(0.0f, 0.0f, 0.0f);
      if (to_shade) {
        float t;
        const int id = bvh_walk_result(wk, t);
        Ray32 fr;
        ray_setup_unit(fr, r.O, r.D);
        L.O = r.O, L.D = r.D;
        if (QUEUE) {
          owner = L.pix;
          const int px = Q.x0 + (L.pix & 7), py = Q.y0 + (L.pix >> 3);
          n_rays += (px < S.W) & (py < S.H);  // scene.go:242; a pair outside the frame made no call
          const int base = DIFFUSE ? ((lband * 8 + (L.pix >> 3)) * S.W + px) * J.pm_batch - J.s_first : 0;
          path_shade<2, DIFFUSE ? 2 : 1>(S, S.prims, S.mats, fx, fy, pix_key, s_end, L, alive, fr, id, t, &Q, &J.pm, base, stack, 0);
          give = L.acc;
        } else {
          n_rays++;  // scene.go:242
          path_shade<0, DIFFUSE ? 2 : 1>(S, S.prims, S.mats, fx, fy, pix_key, s_end, L, alive, fr, id, t, nullptr, &J.pm, pm_base,
                                         stack, 0);
        }
        if (alive) {
          r.O = L.O, r.D = L.D;
          bvh_ray_setup(g, r, S.bvh_grid);
          bvh_walk_begin(wk);
          if (QUEUE && !((Q.x0 + (L.pix & 7) < S.W) & (Q.y0 + (L.pix >> 3) < S.H))) wk.ref = GRT_BVH_DONE;
        }
      }
      if (QUEUE) {
        Q.next += __builtin_popcountll(__ballot(to_shade & (L.depth == 1)));  // the pairs path_shade has just handed out
        unsigned long long m = __ballot(to_shade & ((give.x != 0.0f) | (give.y != 0.0f) | (give.z != 0.0f)));
        while (m) {  // usually empty: most levels carry no radiance
          const int l = __builtin_ctzll(m);
          m &= m - 1;
          const int p = __builtin_amdgcn_readlane(owner, l);
          const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(give.x), l));
          const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(give.y), l));
          const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(give.z), l));
          int me;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(me));
          if (me == p) {
            // round to nearest (truncation towards zero would bias dim scenes down by half a unit per path)
            sum_x += __double2ll_rn((double)cx * J.acc_scale);
            sum_y += __double2ll_rn((double)cy * J.acc_scale);
            sum_z += __double2ll_rn((double)cz * J.acc_scale);
          }
        }
      }
      zmask = (zmask & ~mS) | __ballot(to_shade & alive & ray_has_zero(r));
    } else if (nI * J.inner_weight >= nL * 4) {
      if (COUNT) n_inner += at_inner, c_steps[0]++, c_lanes[0] += nI;
#ifdef GRT_COUNT_DEPTH  // ray_count[11 + r] += node visits at depth = r (mod 3); [14] += visits at depth >= 17
      if (COUNT && at_inner) {
        const int dp = S.bvh_slot_depth[wk.ref];
        c_depth[dp % 3]++, c_depth[3] += dp >= 17;
      }
#endif
      if (at_inner) {
        if (zmask & mI) bvh_inner_step<true>(S.bvh_nodes, S.bvh_grid, r, g, wk, stack, 0);
        else bvh_inner_step<false>(S.bvh_nodes, S.bvh_grid, r, g, wk, stack, 0);
      }
      // more nodes right away while enough lanes still hold one (at most inner_run per round): saves
      // scheduling rounds; the lanes at leaves or waiting for shading idle meanwhile
      for (int it = 1; it < J.inner_run; ++it) {
        const bool again = bvh_at_inner(wk);
        const unsigned long long mA = __ballot(again);
        if (__builtin_popcountll(mA) < J.inner_again) break;
        if (COUNT) n_inner += again, c_steps[0]++, c_lanes[0] += __builtin_popcountll(mA), c_rounds++, c_alive += nI + nL + nS;
#ifdef GRT_COUNT_DEPTH
        if (COUNT && again) {
          const int dp = S.bvh_slot_depth[wk.ref];
          c_depth[dp % 3]++, c_depth[3] += dp >= 17;
        }
#endif
        if (again) {
          if (zmask & mA) bvh_inner_step<true>(S.bvh_nodes, S.bvh_grid, r, g, wk, stack, 0);
          else bvh_inner_step<false>(S.bvh_nodes, S.bvh_grid, r, g, wk, stack, 0);
        }
      }
    } else {
      if (COUNT && at_leaf) n_objs += (~wk.ref) & 7;
      if (COUNT) c_steps[1]++, c_lanes[1] += nL;
      if (at_leaf) {
        if (zmask & mL) bvh_leaf_step<true>(S.bvh_prims, r, S.bias, wk, stack, 0);
        else bvh_leaf_step<false>(S.bvh_prims, r, S.bias, wk, stack, 0);
      }
    }
  }
  if (valid) {
    float* o = out + (((size_t)chunk * (size_t)J.rows_local + (size_t)lrow) * (size_t)S.W + (size_t)x) * 3;
    if (QUEUE) {
      o[0] = (float)((double)sum_x * J.acc_inv);
      o[1] = (float)((double)sum_y * J.acc_inv);
      o[2] = (float)((double)sum_z * J.acc_inv);
    } else {
      o[0] = L.acc.x;
      o[1] = L.acc.y;
      o[2] = L.acc.z;
    }
#ifdef GRT_COUNT_MAP  // experiment (tools/exp_c5_map.py, per-pixel loops only): the frame shows every pixel's work instead of its radiance
    if (COUNT && !QUEUE) o[0] = (float)n_inner, o[1] = (float)n_objs, o[2] = (float)n_rays;
#endif
  }
  if (ray_count) {
    unsigned long long tot = wave_sum_u32((unsigned)n_rays);
    if (lane == 0) atomicAdd(ray_count, tot);
    if (COUNT) {
      const unsigned long long ti = wave_sum_u32((unsigned)n_inner), to = wave_sum_u32((unsigned)n_objs);
      if (lane == 0) {
        atomicAdd(ray_count + 1, ti), atomicAdd(ray_count + 2, to);
        // [3..5] wave steps of each kind (inner, leaf, shading), [6..8] the lanes that took part, [9] rounds (one per
        // step), [10] lanes with work of any kind summed over the rounds
        for (int k = 0; k < 3; ++k) atomicAdd(ray_count + 3 + k, (unsigned long long)c_steps[k]), atomicAdd(ray_count + 6 + k, (unsigned long long)c_lanes[k]);
        atomicAdd(ray_count + 9, (unsigned long long)c_rounds), atomicAdd(ray_count + 10, (unsigned long long)c_alive);
      }
#ifdef GRT_COUNT_DEPTH
      for (int k = 0; k < 4; ++k) {
        const unsigned long long t = wave_sum_u32(c_depth[k]);
        if (lane == 0) atomicAdd(ray_count + 11 + k, t);
      }
#endif
    }
  }
}

// out[i] = slab[0][i] + slab[1][i] + ... in chunk order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_sum_chunks(const float* __restrict__ slab, int n_chunks, size_t n,
                                                    float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float a = slab[i];
    for (int c = 1; c < n_chunks; ++c) a += slab[(size_t)c * n + i];
    out[i] = a;
  }
}

// Photon-mapped trace(): total[i] += frame[i] + sum over the batch's samples of
// w * (irradiance from the global map + irradiance from the caustic map)  (scene.go:313-319).
// Requests that were never written (NaN position) contribute nothing.  Fixed order: reproducible.
__global__ __launch_bounds__(256) void k_pm_accumulate(const float* __restrict__ frame, const float* __restrict__ qpos,
                                                       const float* __restrict__ qw, const float* __restrict__ irr_g,
                                                       const float* __restrict__ irr_c /* may be NULL: one map */, int batch,
                                                       size_t n_pixels, float* __restrict__ total, int first_batch) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  float r = frame[3 * i], g = frame[3 * i + 1], b = frame[3 * i + 2];
  for (int s = 0; s < batch; ++s) {
    const size_t q = 3 * (i * (size_t)batch + (size_t)s);
    if (qpos[q] == qpos[q]) {
      r = fmaf(qw[q], irr_c ? irr_g[q] + irr_c[q] : irr_g[q], r);
      g = fmaf(qw[q + 1], irr_c ? irr_g[q + 1] + irr_c[q + 1] : irr_g[q + 1], g);
      b = fmaf(qw[q + 2], irr_c ? irr_g[q + 2] + irr_c[q + 2] : irr_g[q + 2], b);
    }
  }
  if (first_batch) total[3 * i] = r, total[3 * i + 1] = g, total[3 * i + 2] = b;
  else total[3 * i] += r, total[3 * i + 1] += g, total[3 * i + 2] += b;
}

// scene.go:98-99 + WriteColor (scene.go:67-74): c = sum/samples; Gamma(2) = sqrt; Clamp =
// min(1, c); uint8(255.99*c) truncation; bytes B,G,R.  bpp >= 3; byte 3.. untouched.
// shard / n_shards: the rows of rgb_sum are a shard's packed bands (row r = row r % 8 of band shard + (r / 8) *
// n_shards) and `bgr` is the WHOLE frame: the bytes go to their frame rows (n_shards == 1: row r is row r).
__global__ __launch_bounds__(256) void k_tonemap(const float* __restrict__ rgb_sum, int W, int rows,
                                                 float inv_samples, uint8_t* __restrict__ bgr, int pitch,
                                                 int bpp, int shard, int n_shards) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)W * (size_t)rows, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    size_t yy = i / (size_t)W;
    const size_t xx = i - yy * (size_t)W;
    if (n_shards > 1) yy = ((size_t)shard + (yy >> 3) * (size_t)n_shards) * 8 + (yy & 7);
    const float* c = rgb_sum + i * 3;
    // Correctly rounded sqrt: the byte must not depend on a 1-ulp sqrt difference.  A negative
    // mean (only NormalMaterial produces one: (n+0.5)*0.5 with n = -1) gives NaN through
    // Pow(c, 0.5), Min(1, NaN) = NaN, and uint8(NaN) is 0 on amd64 (CVTTSD2SL -> low byte).
    float r = __fsqrt_rn(c[0] * inv_samples), g = __fsqrt_rn(c[1] * inv_samples), b = __fsqrt_rn(c[2] * inv_samples);
    r = r != r ? 0.0f : fminf(1.0f, r);
    g = g != g ? 0.0f : fminf(1.0f, g);
    b = b != b ? 0.0f : fminf(1.0f, b);
    uint8_t* o = bgr + yy * (size_t)pitch + xx * (size_t)bpp;
    o[0] = (uint8_t)(255.99f * b);
    o[1] = (uint8_t)(255.99f * g);
    o[2] = (uint8_t)(255.99f * r);
  }
}

// One thread per (pixel, sample): rgb = trace(), n_rays = intersect() calls.  Same
// path_advance as k_pathtrace, stopped after one sample.
template <int NP>
__global__ __launch_bounds__(256) void k_trace_samples(KScene S, const int* __restrict__ pixel_xy, int n_pixels,
                                                       int samples, float* __restrict__ rgb,
                                                       int* __restrict__ n_rays_out) {
  __shared__ float4 lds[NP > 0 ? (2 + GRT_MAT_STRIDE) * NP : 1];
  __shared__ int bvh_stack[NP < 0 ? GRT_BVH_STACK * 256 : 1];
  SceneTables<NP> T;
  load_tables<NP>(S, lds, T);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < (size_t)n_pixels * (size_t)samples;
  const size_t j = valid ? i : 0;
  const int pi = (int)(j / (size_t)samples), s = (int)(j % (size_t)samples);
  const int x = pixel_xy[2 * pi], y = pixel_xy[2 * pi + 1];
  const uint32_t pix_key = grt_rng_pixel_key(S.seed32, (uint32_t)y * (uint32_t)S.W + (uint32_t)x);
  const float fx = (float)x, fy = (float)y;
  LaneState L;
  lane_begin(S, fx, fy, pix_key, s, L);
  int n_rays = 0;
  bool alive = valid;
  while (__any(alive)) {
    if (alive) path_advance<NP>(S, T.tp, T.tm, fx, fy, pix_key, s + 1, L, alive, n_rays, bvh_stack + threadIdx.x);
  }
  if (valid) {
    rgb[i * 3 + 0] = L.acc.x;
    rgb[i * 3 + 1] = L.acc.y;
    rgb[i * 3 + 2] = L.acc.z;
    if (n_rays_out) n_rays_out[i] = n_rays;
  }
}

// Config C2, parity variant.  FP32 intersect32/surface32 (the functions the renderer uses);
// lanes whose decision margin is small (ambiguous32) are re-tested in FP64 with the
// reference's operation order, which makes hit/miss and the object id bit-exact.  Hierarchy
// scenes take the same three steps through the walks of grt_parity.h.
__global__ __launch_bounds__(256) void k_primary_hits(KScene S, KScene64 S64, int jitter_mode, int* __restrict__ id_out,
                                                      float* __restrict__ t_out, float* __restrict__ n_out,
                                                      unsigned long long* __restrict__ fallback_count,
                                                      int force_mode /*0 auto, 1 all FP32, 2 all FP64*/) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)S.W * (size_t)S.H;
  const bool valid = i < n;
  const size_t j = valid ? i : 0;
  const int y = (int)(j / (size_t)S.W), x = (int)(j - (size_t)y * (size_t)S.W);
  uint32_t b0 = 1u << 15, b1 = 1u << 15;  // 0.5 * 2^16
  if (jitter_mode == 1) {
    const uint32_t h = grt_rng_hash(grt_rng_path_key(grt_rng_pixel_key(S.seed32, (uint32_t)j), 0), GRT_PAIR_JITTER);
    b0 = h & 0xffffu, b1 = h >> 16;
  }
  // Rays are generated in float64 exactly as the reference does (scene.go:93-95,
  // camera.go:37) and rounded once to FP32 for the FP32 intersection.
  d3 O64, D64;
  make_primary64(S64, S.W, S.H, x, y, (double)b0 * (1.0 / 65536.0), (double)b1 * (1.0 / 65536.0), O64, D64);
  f3 O = mk3((float)O64.x, (float)O64.y, (float)O64.z), D = mk3((float)D64.x, (float)D64.y, (float)D64.z);
  Ray32 r;
  ray_setup(r, O, D);
  float t;
  int id;
  __shared__ int bvh_stack[GRT_BVH_STACK * 256];
  bool amb;
  if (S.bvh_nodes) {  // hierarchy scenes (grt_parity.h)
    int* stack = bvh_stack + threadIdx.x;
    const bool careful = __any(ray_has_zero(r));
    float eb = 0.0f;
    amb = false;
    if (force_mode == 1)  // the renderer's own walk
      id = careful ? intersect_bvh_t<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, stack, 256)
                   : intersect_bvh_t<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, stack, 256);
    else if (force_mode == 2)
      id = -1, amb = true;
    else
      id = careful ? intersect_bvh_margin<true>(S, r, t, eb, amb, stack, 256) : intersect_bvh_margin<false>(S, r, t, eb, amb, stack, 256);
    f3 p, nn32 = mk3(0.0f, 0.0f, 0.0f);
    if (!amb && id >= 0) surface32(S, id, r, t, p, nn32);
    // a certified winner whose FP32 t or normal may miss 1e-5: its own Hit and Surface in float64, no walk
    bool refine = force_mode == 0 && !amb && id >= 0 && hit_accuracy_doubtful(S, r, id, t, eb, t);
    double t64 = 0.0;
    if (refine) {
      t64 = prim_hit64(S64.prims + 8 * (size_t)id, O64, D64, S64.bias, 1.7976931348623157e308);
      if (!(t64 > S64.bias)) amb = true, refine = false;  // cannot happen for a certified winner; belt and braces
    }
    if (__any(amb)) {
      double tw;
      const int id64 = careful ? intersect_bvh64<true>(S, S64, r, O64, D64, amb, tw, stack, 256)
                               : intersect_bvh64<false>(S, S64, r, O64, D64, amb, tw, stack, 256);
      if (amb) id = id64, t64 = tw;
    }
    f3 nn = nn32;
    if (id < 0) t = 0.0f, nn = mk3(0.0f, 0.0f, 0.0f);
    if ((amb | refine) && id >= 0) {
      const d3 n64 = surface64(S64.prims + 8 * (size_t)id, O64, D64, t64);
      nn = mk3((float)n64.x, (float)n64.y, (float)n64.z);
      t = (float)t64;
    }
    amb |= refine;  // counted as re-tested in float64
    if (valid) {
      id_out[i] = id;
      t_out[i] = t;
      n_out[i * 3 + 0] = nn.x;
      n_out[i * 3 + 1] = nn.y;
      n_out[i * 3 + 2] = nn.z;
    }
    if (fallback_count) {
      unsigned long long m = __ballot(amb && valid);
      if ((threadIdx.x & 63) == 0 && m) atomicAdd(fallback_count, (unsigned long long)__popcll(m));
    }
    return;
  }
  id = intersect32(S, r, t);
  f3 p, nn = mk3(0.0f, 0.0f, 0.0f);
  if (id >= 0) surface32(S, id, r, t, p, nn);
  else t = 0.0f;
  amb = force_mode == 2 || (force_mode == 0 && ambiguous32(S, r, id, t));
  if (amb) {
    double t64;
    id = intersect64(S64, S.n_prims, O64, D64, t64);
    nn = mk3(0.0f, 0.0f, 0.0f);
    t = 0.0f;
    if (id >= 0) {
      d3 n64 = surface64(S64.prims + 8 * id, O64, D64, t64);
      nn = mk3((float)n64.x, (float)n64.y, (float)n64.z);
      t = (float)t64;
    }
  }
  if (valid) {
    id_out[i] = id;
    t_out[i] = t;
    n_out[i * 3 + 0] = nn.x;
    n_out[i * 3 + 1] = nn.y;
    n_out[i * 3 + 2] = nn.z;
  }
  if (fallback_count) {
    unsigned long long m = __ballot(amb && valid);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(fallback_count, (unsigned long long)__popcll(m));
  }
}

// Config C2, roofline variant: `samples` jittered primary rays per pixel
// in-kernel, FP32 only, one checksum word per pixel = sum of (id+1).  A wave owns an 8x8
// tile so its rays are coherent.  NP > 0: objects in LDS, unrolled loop (as k_pathtrace).
// mode bit 0: min/max slab test instead of the direction-octant tables (A/B, opts.flags variant 7); bit 1: round 2's loop
// for every tile (per-lane octant, world coordinates; variant 9); bit 2: the eye lies outside every box of the scene
// (decided on the host in float64: intersect_primary_rel's EYE_OUT).
// Production path for <= 8 objects (intersect_primary_rel, grt_device.h): the tables hold the objects relative to the
// eye; a tile whose corner directions agree in every sign reads ONE octant's table at a wave-uniform address; the ray
// direction is base(pixel) + (horizontal / W) ju + (vertical / H) jv with the 2^-16 of the 16-bit jitters folded into
// the two vectors (6 FMA); one hash gives both jitters (grt_rng.h).  Tiles crossed by a zero of a direction component
// (the centre column and row of the image: < 1 % of the tiles) take round 2's loop.
template <int NP, bool EYE_OUT>
GRT_DEV uint32_t sweep_tile_rel(const KScene& S, const float* __restrict__ q, uint32_t pix_key, f3 base, int samples) {
  const float su = (1.0f / 65536.0f) / (float)S.W, sv = (1.0f / 65536.0f) / (float)S.H;
  const f3 hu = mk3(S.cam.hx * su, S.cam.hy * su, S.cam.hz * su), vv = mk3(S.cam.vx * sv, S.cam.vy * sv, S.cam.vz * sv);
  uint32_t sum = 0;
  for (int s = 0; s < samples; ++s) {
    asm volatile("" ::: "memory");  // re-read the table from LDS each ray (keeps it out of long-lived VGPRs)
    const uint32_t h = grt_rng_hash(grt_rng_path_key(pix_key, (uint32_t)s), GRT_PAIR_JITTER);
    const float ju = (float)(h & 0xffffu), jv = (float)(h >> 16);
    const f3 D = mk3(fmaf(vv.x, jv, fmaf(hu.x, ju, base.x)), fmaf(vv.y, jv, fmaf(hu.y, ju, base.y)), fmaf(vv.z, jv, fmaf(hu.z, ju, base.z)));
    const f3 invD = mk3(rcp32(D.x), rcp32(D.y), rcp32(D.z));
    float t;
    sum += (uint32_t)(intersect_primary_rel<NP, EYE_OUT>(q, S.box_mask, D, invD, rcp32(dot3(D, D)), S.bias, t) + 1);
  }
  return sum;
}

template <int NP, bool PREFETCH = false>
__global__ __launch_bounds__(256) void k_primary_sweep(KScene S, int samples, uint32_t* __restrict__ checksum, int mode) {
  __shared__ float4 lds[NP > 0 ? (2 + GRT_MAT_STRIDE) * NP : 1];
  __shared__ int bvh_stack[NP < 0 ? GRT_BVH_STACK * 256 : 1];
  constexpr bool OCT = NP > 0 && !PREFETCH;
  constexpr int NPP = NP > 0 ? NP : 1;
  __shared__ __attribute__((aligned(16))) float oct_tables[OCT ? 8 * GRT_OCT_STRIDE(NPP) : 4];
  __shared__ __attribute__((aligned(16))) float rel_tables[OCT ? 8 * GRT_OCT_STRIDE(NPP) : 4];
  SceneTables<NP> T;
  const bool rel = OCT && !(mode & 3);
  if (OCT && rel) {
    const f3 eye = mk3(S.cam.ox, S.cam.oy, S.cam.oz);
    for (int i = threadIdx.x; i < 8 * NPP; i += blockDim.x)
      octant_record_rel(S.prims[2 * (i % NPP)], S.prims[2 * (i % NPP) + 1], i / NPP, eye, rel_tables + (i / NPP) * GRT_OCT_STRIDE(NPP) + 8 * (i % NPP));
  }
  load_tables<NP>(S, lds, T, OCT && !(mode & 1) ? oct_tables : nullptr);  // ends with a barrier
  const int lane = threadIdx.x & 63;
  const int tiles_x = (S.W + 7) >> 3, tiles_y = (S.H + 7) >> 3;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tiles_x * tiles_y) return;
  const int x = (tile % tiles_x) * 8 + (lane & 7), y = (tile / tiles_x) * 8 + (lane >> 3);
  const bool valid = (x < S.W) & (y < S.H);
  const uint32_t pix_key = grt_rng_pixel_key(S.seed32, (uint32_t)y * (uint32_t)S.W + (uint32_t)x);
  uint32_t sum = 0;
  if (OCT && rel) {
    // the directions through the four corners of the lane's pixel bound every jittered direction of the pixel
    f3 O, d00, d10, d01, d11;
    make_primary(S, x, y, 0.0f, 0.0f, O, d00);
    make_primary(S, x, y, 1.0f, 0.0f, O, d10);
    make_primary(S, x, y, 0.0f, 1.0f, O, d01);
    make_primary(S, x, y, 1.0f, 1.0f, O, d11);
    const f3 mn = mk3(fminf(fminf(d00.x, d10.x), fminf(d01.x, d11.x)), fminf(fminf(d00.y, d10.y), fminf(d01.y, d11.y)),
                      fminf(fminf(d00.z, d10.z), fminf(d01.z, d11.z)));
    const f3 mx = mk3(fmaxf(fmaxf(d00.x, d10.x), fmaxf(d01.x, d11.x)), fmaxf(fmaxf(d00.y, d10.y), fmaxf(d01.y, d11.y)),
                      fmaxf(fmaxf(d00.z, d10.z), fmaxf(d01.z, d11.z)));
    // clear of zero by far more than the FP32 rounding of a direction (a few 1e-7 of the terms' magnitudes)
    const f3 mg = mk3(1e-5f * (fabsf(S.cam.lx) + fabsf(S.cam.hx) + fabsf(S.cam.vx)), 1e-5f * (fabsf(S.cam.ly) + fabsf(S.cam.hy) + fabsf(S.cam.vy)),
                      1e-5f * (fabsf(S.cam.lz) + fabsf(S.cam.hz) + fabsf(S.cam.vz)));
    const bool clear = ((mn.x > mg.x) | (mx.x < -mg.x)) & ((mn.y > mg.y) | (mx.y < -mg.y)) & ((mn.z > mg.z) | (mx.z < -mg.z));
    const unsigned long long all = __ballot(true), bx = __ballot(mx.x < 0.0f), by = __ballot(mx.y < 0.0f), bz = __ballot(mx.z < 0.0f);
    const bool uniform = (__ballot(clear) == all) & ((bx == 0) | (bx == all)) & ((by == 0) | (by == all)) & ((bz == 0) | (bz == all));
    if (uniform) {  // wave-uniform
      const int oct = (bx ? 1 : 0) | (by ? 2 : 0) | (bz ? 4 : 0);
      const float* q = rel_tables + oct * GRT_OCT_STRIDE(NPP);
      sum = (mode & 4) ? sweep_tile_rel<NPP, true>(S, q, pix_key, d00, samples) : sweep_tile_rel<NPP, false>(S, q, pix_key, d00, samples);
      if (valid) checksum[(size_t)y * (size_t)S.W + (size_t)x] = sum;
      return;
    }
  }
  for (int s = 0; s < samples; ++s) {
    if (NP > 0) asm volatile("" ::: "memory");  // re-read the objects from LDS each ray (see path_advance)
    const grt_key pk = grt_rng_path_key(pix_key, (uint32_t)s);
    f3 O, D;
    float ju, jv;
    grt_rng_draw2(pk, GRT_PAIR_JITTER, &ju, &jv);
    make_primary(S, x, y, ju, jv, O, D);
    Ray32 r;
    ray_setup(r, O, D);
    float t;
    int id;
    if (NP < 0)
      id = __any(ray_has_zero(r)) ? intersect_bvh_t<true>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, bvh_stack + threadIdx.x, 256)
                                  : intersect_bvh_t<false>(S.bvh_nodes, S.bvh_prims, S.bvh_grid, r, S.bias, t, bvh_stack + threadIdx.x, 256);
    else if (__builtin_expect(__any(ray_has_zero(r)), 0)) id = intersect32_t<true>(S, r, t);
    else if (NP > 0 && T.oct) id = intersect_lds_oct<NPP>(T.oct + ray_octant(r.D) * GRT_OCT_STRIDE(NPP), S.box_mask, r, S.bias, t);
    else if (NP > 0) id = intersect_lds<NP, PREFETCH>(T.tp, S.box_mask, r, S.bias, t);
    else id = intersect32_t<false>(S, r, t);
    sum += (uint32_t)(id + 1);
  }
  if (valid) checksum[(size_t)y * (size_t)S.W + (size_t)x] = sum;
}
