// grt_rng.h — counter-based RNG shared by host and device code.
//
// Replaces the reference's sequential math/rand streams (pkg/tracer/scene.go:86,93-94,114
// and the process-global source at pkg/geom/vec3.go:152).  A uniform is named by
// (seed, pixel, sample, dimension); there is no state to carry between draws, so a lane can
// start any sample of any pixel and the image does not depend on how the work is sharded.
//
//   mix32       2-multiply avalanche hash ("lowbias32"), a bijection on 32 bits
//   pixel_key   once per pixel  (per lane, amortised over all samples of the pixel); a bijection of the pixel
//               index for a given seed
//   path_key    once per path: TWO words, a = mix32(pixel_key + sample * C2 + ..), b = pixel_key + sample * C4
//               + .. (not hashed: it only has to tell apart the rare paths whose first words collide, and one
//               add per path keeps it current).  C2 - C4 is odd, so two (pixel, sample) pairs that agree in both
//               words agree in pixel_key + sample * C2 and in pixel_key + sample * C4, hence in sample and in
//               pixel: no two paths of a frame share a key.  (With one 32-bit word, as in round 1, a third of the 2*10^9
//               paths of a 1080p x 1000 spp frame shared their whole stream with a path of another pixel.)
//   hash        one mix32 per PAIR of uniforms with word b folded in between its two multiply rounds (one
//               three-input xor, no extra instruction); 16 random bits per uniform, exact in f32 and f64
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GRT_HD __host__ __device__ __forceinline__
#else
#define GRT_HD static inline
#endif

GRT_HD uint32_t grt_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
GRT_HD uint32_t grt_rng_seed32(uint64_t seed) {
  return grt_mix32((uint32_t)seed ^ 0x9e3779b9U) ^ grt_mix32((uint32_t)(seed >> 32) + 0x85ebca6bU);
}
GRT_HD uint32_t grt_rng_pixel_key(uint32_t seed32, uint32_t pixel) {
  return grt_mix32(seed32 ^ (pixel * 0x9E3779B1U));
}
typedef struct {
  uint32_t a, b;
} grt_key;
#define GRT_RNG_C4 0x27D4EB2EU  // 0x85EBCA77 - 0x27D4EB2E is odd
GRT_HD grt_key grt_rng_path_key(uint32_t pixel_key, uint32_t sample) {
  grt_key k;
  k.a = grt_mix32(pixel_key + sample * 0x85EBCA77U + 0xC2B2AE3DU);
  k.b = pixel_key + sample * GRT_RNG_C4 + 0x165667B1U;
  return k;
}
// The key of the NEXT sample of the same pixel from the current one: word b by one add.
GRT_HD grt_key grt_rng_next_path_key(uint32_t pixel_key, uint32_t next_sample, grt_key cur) {
  grt_key k;
  k.a = grt_mix32(pixel_key + next_sample * 0x85EBCA77U + 0xC2B2AE3DU);
  k.b = cur.b + GRT_RNG_C4;
  return k;
}
// One hash serves a PAIR of uniforms (round 3): pair p of a path's stream is hash(key, p), its low 16 bits are the
// first uniform of the pair (dimension 2p), its high 16 bits the second (dimension 2p + 1).  Every consumer on this
// path draws its uniforms in pairs — the two pixel jitters (scene.go:93-94), SampleHemisphereNormal's two draws
// (vec3.go:203-204) — so a trace() level costs one hash instead of two.  For a fixed key the hash is a bijection of
// 32 bits, so the pairs (u1, u2) are exactly equidistributed on the 2^16 x 2^16 grid.  16 random bits per uniform
// (round 2: 24): exactly representable in f32 and f64; the grid's bias (2^-17 relative) is far below the 8-bit output.
GRT_HD uint32_t grt_rng_hash(grt_key key, uint32_t pair) {
  uint32_t x = key.a + (pair + 1U) * 0x9E3779B9U;
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= (x >> 15) ^ key.b;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
GRT_HD uint32_t grt_rng_bits16(grt_key key, uint32_t dim) {
  const uint32_t h = grt_rng_hash(key, dim >> 1);
  return (dim & 1U) ? h >> 16 : h & 0xffffU;
}
// Both uniforms of pair `pair` (dimensions 2 pair, 2 pair + 1) from one hash.
GRT_HD void grt_rng_draw2(grt_key path_key, uint32_t pair, float* u1, float* u2) {
  const uint32_t h = grt_rng_hash(path_key, pair);
  *u1 = (float)(h & 0xffffU) * (1.0f / 65536.0f);
  *u2 = (float)(h >> 16) * (1.0f / 65536.0f);
}
GRT_HD float grt_rng_draw(grt_key path_key, uint32_t dim) {
  return (float)grt_rng_bits16(path_key, dim) * (1.0f / 65536.0f);
}
GRT_HD double grt_rng_draw64(grt_key path_key, uint32_t dim) {
  return (double)grt_rng_bits16(path_key, dim) * (1.0 / 65536.0);
}

// Dimension layout of one camera path: 0,1 = pixel jitter (scene.go:93-94); the hit at depth
// d (1-based, scene.go:237) owns 2+2(d-1)+{0,1}: SampleHemisphereNormal's two draws in call
// order (vec3.go:203-204) for Lambert / rough metal, or — the material branches of trace() are
// exclusive — Refract's roulette draw (vec3.go:152) in slot 0.
#define GRT_DIM_JITTER_U 0u
#define GRT_DIM_JITTER_V 1u
#define GRT_DIM_HIT(depth, k) (2u + 2u * (uint32_t)((depth)-1) + (uint32_t)(k))
#define GRT_PAIR_JITTER 0u                       // dimensions 0, 1
#define GRT_PAIR_HIT(depth) ((uint32_t)(depth))  // dimensions GRT_DIM_HIT(depth, 0), GRT_DIM_HIT(depth, 1)

// Photon walks (scene.go:140-165, 354-424) use pixel = photon index, sample = map id:
// 0,1 = emission direction; bounce d owns 2+4(d-1)+{0,1} hemisphere, +2 Refract, +3 roulette.
#define GRT_PDIM_EMIT_U 0u
#define GRT_PDIM_EMIT_V 1u
#define GRT_PDIM_HIT(depth, k) (2u + 4u * (uint32_t)((depth)-1) + (uint32_t)(k))
#define GRT_PPAIR_EMIT 0u
#define GRT_PPAIR_HEMI(depth) (1u + 2u * (uint32_t)((depth)-1))  // GRT_PDIM_HIT(depth, 0), (depth, 1)
#define GRT_PPAIR_RR(depth) (2u + 2u * (uint32_t)((depth)-1))    // GRT_PDIM_HIT(depth, 2) Refract, (depth, 3) roulette
