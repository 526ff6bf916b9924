/*
 * grt.h — C ABI of libgrt_hip.so, the MI355X (gfx950) implementation of the
 * go-raytracer per-pixel rendering hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own; the seam is the
 * method
 *
 *     func (scene Scene) Render(pixels []byte, pitch int, samples int)   pkg/tracer/scene.go:79
 *
 * and everything it calls per pixel-sample (camera.go:34, scene.go:169-181, :237-349,
 * sphere.go:23-58, aabb.go:25-72, vec3.go:129-159,202-218) plus the photon pass
 * (scene.go:140-165, :354-424, photon_mapping.go:125-209).  A cgo shim inside package
 * `tracer` flattens `Scene` into the POD records below and calls these entry points
 * (INTEGRATION.md shows the shim).
 *
 * Conventions
 *   - every function returns 0 on success, a negative GRT_E* code on failure, and sets a
 *     thread-local message readable through grt_last_error();
 *   - all structs are plain little-endian POD with no pointers inside, so Go slices /
 *     numpy arrays can be passed directly (cgo pointer rule);
 *   - host buffers are caller-owned and never retained after the call returns;
 *   - "_dev" entry points take DEVICE pointers and a HIP stream (void* = hipStream_t,
 *     NULL = the null stream), work on the CURRENT device and enqueue without synchronising;
 *     host-buffer entry points select opts->first_device (or the device a photon map lives
 *     on) and leave the caller's current device as they found it;
 *   - object id == index into the prims array == index into Scene.Objects
 *     (cmd/raytracer/main.go:40-49); -1 means miss.
 *
 * All reference arithmetic is float64; the records keep float64 so no information is lost
 * at the boundary.  The render kernels compute in FP32 (north star), the primary-hit
 * parity kernel re-tests in FP64 every lane whose FP32 decision margin is small.
 */
#ifndef GRT_H
#define GRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRT_OK 0
#define GRT_EINVAL (-1)  /* bad argument (NULL, negative size, pitch too small, ...) */
#define GRT_EHIP (-2)    /* a HIP runtime call failed; message holds hipGetErrorString */
#define GRT_ENODEV (-3)  /* no usable gfx950 device */
#define GRT_ELIMIT (-4)  /* size beyond a documented maximum */
#define GRT_EHOST (-5)   /* host-side failure inside the library (out of memory, no thread): nothing crosses the C ABI as a C++ exception */

#define GRT_KIND_SPHERE 0
#define GRT_KIND_AABB 1

/* Camera exactly as NewCamera leaves it (pkg/tracer/camera.go:10-13,16-30). */
typedef struct {
  double origin[3];
  double lower_left[3];
  double horizontal[3];
  double vertical[3];
} grt_camera;

/* Material record (pkg/tracer/material.go:4-13). */
typedef struct {
  double color[3];
  double refr_index;
  double reflectivity;
  double roughness;
  double emittance;
  uint8_t transparent;
  uint8_t lambert;
  uint8_t normal;
  uint8_t _pad[5];
} grt_material;

/* Primitive: Sphere{Center,Radius,Mat} (sphere.go:10-14) or
 * AABB{MinBound,MaxBound,Center,Mat} (aabb.go:10-15; Center is derived = (min+max)/2). */
typedef struct {
  int32_t kind;     /* GRT_KIND_SPHERE | GRT_KIND_AABB */
  int32_t material; /* index into the materials array */
  double a[3];      /* sphere centre | box MinBound */
  double b[3];      /* unused        | box MaxBound */
  double radius;    /* sphere radius | unused */
} grt_prim;

/* The scalar part of tracer.Scene (scene.go:17-27) + bias (scene.go:14). */
typedef struct {
  int32_t width;     /* Scene.W */
  int32_t height;    /* Scene.H */
  int32_t max_depth; /* Scene.maxDepth, 6 in NewScene (scene.go:62) */
  int32_t accel;     /* GRT_ACCEL_*; no counterpart in the reference (it scans linearly, scene.go:173) */
  double bias; /* 0.001 (scene.go:14) */
  grt_camera cam;
} grt_scene_desc;

/* How Scene.intersect finds the nearest object.  AUTO: the reference's linear scan up to 208
 * objects, a bounding-volume hierarchy above (same result, see csrc/grt_bvh.h; measured crossover
 * near 210 objects, 18x faster at 4096); BVH forces the hierarchy, LINEAR the scan (<= 4096 objects).
 * The shadow loop of the plain-diffuse branch (scene.go:336-344) and the caustic filter of
 * tracePhotons (scene.go:359-363) run through the hierarchy in their order-free forms, so every
 * material, photon emission and photon-mapped rendering work on hierarchy scenes.
 * grt_primary_hits keeps the linear scan with its float64 re-test whenever the scene has <= 4096
 * objects and BVH was not forced. */
#define GRT_ACCEL_AUTO 0
#define GRT_ACCEL_BVH 1
#define GRT_ACCEL_LINEAR 2

/* Run options.  The reference seeds every worker with time.Now().Unix() (scene.go:114);
 * here the seed is explicit and keys a counter-based RNG (pixel, sample, dimension). */
typedef struct {
  uint64_t seed;
  int32_t first_device; /* HIP ordinal of the first device to use */
  int32_t n_devices;    /* 0 or 1 = one device; G>1 = interleaved 8-row bands over G devices */
  int32_t chunk_spp;    /* samples per work item; 0 = library default.  An explicit value must cut the samples into at
                         * most 63 chunks (ceil(samples / chunk_spp) <= 63: one partial-sum slab per chunk), else GRT_ELIMIT */
  int32_t flags;        /* GRT_FLAG_* */
} grt_opts;

#define GRT_FLAG_NONE 0

/* Rows are dealt to devices / ranks in interleaved bands of this many rows
 * (band b -> shard b mod G). */
#define GRT_BAND_ROWS 8

typedef struct grt_scene grt_scene; /* opaque: scene staged as SoA in HBM */
typedef struct grt_pmap grt_pmap;   /* opaque: photon map (SoA photons + kd-tree) in HBM */

/* ---- scene staging ---------------------------------------------------------------- */

/* Replaces NewScene (scene.go:35-64): copies the records, derives Lights / tObjects /
 * lightArea, and stages FP32 + FP64 SoA copies in HBM on first use per device. */
int grt_scene_create(const grt_scene_desc* desc, const grt_prim* prims, int32_t n_prims,
                     const grt_material* mats, int32_t n_mats, grt_scene** out);
void grt_scene_destroy(grt_scene* scene);

/* ---- path tracing ----------------------------------------------------------------- */

/* Replaces the body of Scene.Render (scene.go:79-136): for every pixel, `samples` paths
 * (scene.go:92-97), mean -> Gamma(2) -> Clamp (scene.go:98-99) -> uint8(255.99*c) written
 * as B,G,R at pixels[y*pitch + x*bpp + {0,1,2}], bpp = pitch / width (scene.go:67-74,85).
 * Bytes beyond the third of each pixel are left untouched.  `pixels` must hold (height - 1) * pitch + width * bpp
 * bytes (the last row may end after its last pixel, as a Go slice cut to scene.go:128's copies may); pitch >= 3 * width. */
int grt_render(grt_scene* scene, const grt_opts* opts, uint8_t* pixels, int32_t pitch,
               int32_t samples);

/* Same paths, but returns the linear RGB sums (float32, width*height*3, row-major, R,G,B)
 * before averaging / gamma / clamp — what scene.go:96 accumulates.  For RMSE checks. */
int grt_render_accum(grt_scene* scene, const grt_opts* opts, float* rgb_sum, int32_t samples);

/* Device-resident variant for one shard of a tile-sharded frame: renders the bands
 * b = shard, shard+n_shards, ... (GRT_BAND_ROWS rows each) on the CURRENT device
 * and writes their linear RGB sums, bands packed in ascending order, to d_rgb_sum
 * (float32 [grt_shard_rows()*width*3]).  d_workspace must hold grt_accum_workspace_bytes().
 * Enqueues on `stream`; no allocation, no synchronisation.  d_ray_count (device uint64,
 * may be NULL) is incremented by the number of intersect() calls (scene.go:169) executed. */
int grt_render_accum_dev(grt_scene* scene, const grt_opts* opts, int32_t shard, int32_t n_shards,
                         int32_t samples, float* d_rgb_sum, void* d_workspace,
                         uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream);
/* Sample-range sharding: the WHOLE frame, samples
 * [sample_first, sample_first + samples) of every pixel (the s of scene.go:92), linear RGB
 * sums to d_rgb_sum (float32 [height*width*3]).  Workspace: grt_accum_workspace_bytes(scene,
 * opts, 0, 1, samples).  The frames of disjoint ranges add up to the frame of the union. */
int grt_render_accum_samples_dev(grt_scene* scene, const grt_opts* opts, int32_t sample_first,
                                 int32_t samples, float* d_rgb_sum, void* d_workspace,
                                 uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream);
/* grt_render_accum_dev for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot call hipMalloc — code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  Same inputs and outputs as grt_render_accum_dev minus the workspace: the per-chunk
 * partial sums live in a file-scope device array of the library, GRT_NOALLOC_SCRATCH_BYTES (64 MiB) per GPU,
 * held until the process ends.  Calls no hipMalloc / hipFree.  The scene must already be staged on the
 * CURRENT device (any earlier render on that device does it), else GRT_EINVAL.  opts->chunk_spp == 0: the
 * library's chunk size, coarsened until the slabs (n_chunks * grt_shard_rows * width * 12 B) fit; an explicit
 * chunk_spp whose slabs exceed the scratch is refused with GRT_ELIMIT.  At equal chunk size the frame equals
 * grt_render_accum_dev's bit for bit.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU race. */
#define GRT_NOALLOC_SCRATCH_BYTES ((uint64_t)64 << 20)
int grt_render_accum_noalloc_dev(grt_scene* scene, const grt_opts* opts, int32_t shard, int32_t n_shards,
                                 int32_t samples, float* d_rgb_sum, uint64_t* d_ray_count, void* stream);
int32_t grt_shard_rows(int32_t height, int32_t shard, int32_t n_shards);
/* Bytes of d_workspace the device entry points above need.  0 means either "none needed" (one chunk) or that the
 * arguments are invalid (bad shard, samples <= 0, an explicit opts->chunk_spp that makes more than 63 chunks): the
 * render call with the same arguments then returns the error code and grt_last_error() says which. */
uint64_t grt_accum_workspace_bytes(const grt_scene* scene, const grt_opts* opts, int32_t shard,
                                   int32_t n_shards, int32_t samples);

/* Tone-map + pack on the device (scene.go:98-99 + WriteColor scene.go:67-74):
 * d_rgb_sum [rows*width*3] -> d_bgrx rows of `pitch` bytes, bpp = pitch/width. */
int grt_tonemap_dev(const float* d_rgb_sum, int32_t width, int32_t rows, int32_t samples,
                    uint8_t* d_bgrx, int32_t pitch, void* stream);

/* Per-sample radiance of chosen pixels (parity aid, no reference counterpart beyond
 * scene.go:93-96): for i < n_pixels and s < samples writes trace() of sample s of pixel
 * pixel_xy[2i],pixel_xy[2i+1] to rgb[(i*samples+s)*3 ..], and the number of intersect()
 * calls of that path to n_rays[i*samples+s] (may be NULL).  Host buffers. */
int grt_trace_samples(grt_scene* scene, const grt_opts* opts, const int32_t* pixel_xy,
                      int32_t n_pixels, int32_t samples, float* rgb, int32_t* n_rays);

/* ---- primary rays (config C2) ------------------------------------------------------- */

#define GRT_JITTER_CENTER 0 /* u=(x+0.5)/W, v=(y+0.5)/H */
#define GRT_JITTER_RNG 1    /* u=(x+U0)/W, v=(y+U1)/H with the counter RNG, sample index 0 */

/* One primary ray per pixel through Camera.Ray (camera.go:34-39) and Scene.intersect
 * (scene.go:169-181), then Surface() (sphere.go:56-58, aabb.go:55-72).
 * Outputs (host, row-major): id[W*H] object index or -1; t[W*H] hit distance in units of
 * the unnormalised primary direction (0 on miss); normal_xyz[W*H*3] (0 on miss).
 * fallback_lanes (may be NULL) receives how many pixels were re-tested in FP64. */
int grt_primary_hits(grt_scene* scene, const grt_opts* opts, int32_t jitter_mode, int32_t* id,
                     float* t, float* normal_xyz, uint64_t* fallback_lanes);

/* Intersection-roofline variant of the same kernel: `samples` jittered
 * primary rays per pixel in-kernel, FP32 only, one 32-bit checksum per pixel
 * (d_checksum[W*H], device) = sum over samples of (id+1).  Device pointers; enqueues on
 * `stream`. */
int grt_primary_sweep_dev(grt_scene* scene, const grt_opts* opts, int32_t samples,
                          uint32_t* d_checksum, void* stream);

/* ---- photon map (config C4) --------------------------------------------------------- */

/* mapPhotons for ONE map (scene.go:140-165) + ScalePhotonPower(1000/max)
 * (photon_mapping.go:197-209): emits photons from every light until max_photons are
 * stored, caustics != 0 selects the caustic filter (scene.go:359-363,375) and power
 * (1,1,1) instead of (15,15,15).  Builds the balanced bucket kd-tree (kdtree.New(list, true),
 * photon_mapping.go:207) and stages it with the photons in HBM.  Every leaf bucket of the tree takes 64 photon slots (between n
 * and 2 n slots of 28 bytes for n photons); a map of more than 350 000 000 slots is refused with GRT_ELIMIT (2^28 photons
 * always fit), here and in grt_photon_load / grt_photon_load_dev. */
int grt_photon_build(grt_scene* scene, const grt_opts* opts, int32_t max_photons, int32_t caustics,
                     grt_pmap** out);
int32_t grt_photon_count(const grt_pmap* pmap);
/* HIP ordinal of the device the map's photons and tree live on. */
int32_t grt_photon_device(const grt_pmap* pmap);
/* Photons emitted to fill the map (the reference's loop count, scene.go:152/157). */
uint64_t grt_photon_emitted(const grt_pmap* pmap);
/* Copies the stored photons, in storage (= emission) order, to host arrays: pos[n*3],
 * power[n*3] float32 (already scaled), theta_phi[n*2] bytes (photon_mapping.go:11-16).
 * Any pointer may be NULL. */
int grt_photon_read(const grt_pmap* pmap, float* pos, float* power, uint8_t* theta_phi);
/* Builds a map from caller-supplied photons instead of emitting (for gather parity). */
int grt_photon_load(const grt_opts* opts, const float* pos, const float* power,
                    const uint8_t* theta_phi, int32_t n, grt_pmap** out);
/* Photon-build sharding: a rank traces the emission indices
 * [first, first + count) of light `light` (the loop body of scene.go:152-160 for those
 * iterations) and receives its stored photons compacted in emission order: d_pos4[i] =
 * (x, y, z, direction bytes), d_pow4[i] = (r, g, b unscaled, offset of the emission inside the
 * slice as int bits), *d_stored = how many.  Outputs hold `count` float4 each.  The slices of
 * all ranks concatenated in rank order are the reference's store sequence; grt_photon_load_dev
 * builds the map (power * power_scale = ScalePhotonPower, photon_mapping.go:197-209) from
 * device arrays.  Device pointers; emit enqueues on `stream` and allocates nothing. */
int grt_photon_emit_dev(grt_scene* scene, const grt_opts* opts, int32_t caustics, int32_t light,
                        uint64_t first, int32_t count, float* d_pos4, float* d_pow4,
                        int32_t* d_stored, void* d_workspace, uint64_t workspace_bytes, void* stream);
uint64_t grt_photon_emit_workspace_bytes(int32_t count);
/* grt_photon_load_dev builds the map on the CURRENT device, where d_pos4 / d_pow4 live; it does not
 * switch devices (opts->first_device is not consulted). */
int grt_photon_load_dev(const grt_opts* opts, const float* d_pos4, const float* d_pow4, int32_t n,
                        float power_scale, uint64_t emitted, grt_pmap** out);
/* Lights as NewScene derives them (scene.go:42-46): how many, and int(lightArea / area) of
 * light k, the multiple of `max` its emission loop aims for (scene.go:152,157). */
int32_t grt_scene_light_count(const grt_scene* scene);
int32_t grt_scene_light_share(const grt_scene* scene, int32_t k);
/* IrradianceEst batch (photon_mapping.go:125-162): for each of n queries, the k nearest
 * photons (all photons tied at the k-th distance are kept); radius==0 => use all of them,
 * else those with d^2 < radius^2; sums power of photons with dir.normal < 0, divides by
 * pi*r2 when at least 8 were found (else returns the unscaled sum).  r2 = the largest
 * accepted squared distance (documented intent, photon_mapping.go:136).
 * k <= 512.  Host buffers: pos_xyz[n*3], nrm_xyz[n*3], irrad_rgb[n*3]. */
int grt_photon_gather(const grt_pmap* pmap, const float* pos_xyz, const float* nrm_xyz, int32_t n,
                      float radius, int32_t k, float* irrad_rgb);
/* Device-pointer variant: enqueues on `stream`, allocates nothing.  d_workspace may be NULL;
 * when given it holds grt_photon_gather_workspace_bytes(n) bytes and receives, per query, the
 * squared distance of the k-th nearest photon (+inf: fewer than k in reach, -1: none). */
int grt_photon_gather_dev(const grt_pmap* pmap, const float* d_pos_xyz, const float* d_nrm_xyz,
                          int32_t n, float radius, int32_t k, float* d_irrad_rgb,
                          void* d_workspace, void* stream);
uint64_t grt_photon_gather_workspace_bytes(int32_t n);
void grt_photon_free(grt_pmap* pmap);

/* ---- photon-mapped trace() ------------------------------------------------------------ */

/* Scene.Render with the photon block of trace() enabled (scene.go:308-320, commented out in
 * the reference; scene.globalPmap / causticPmap, scene.go:22-23): a path that ends on a plain
 * diffuse surface adds m.Color * (caustic.IrradianceEst(p,n,1,100) + global.IrradianceEst(p,n,
 * 0,100)) / pi to the direct light of scene.go:322-346.  Same buffers as grt_render /
 * grt_render_accum / grt_render_accum_dev (the _dev form also needs
 * grt_render_pm_workspace_bytes() of workspace).  One device per call; linear-scan scenes. */
int grt_render_pm(grt_scene* scene, const grt_opts* opts, const grt_pmap* global_map,
                  const grt_pmap* caustic_map, uint8_t* pixels, int32_t pitch, int32_t samples);
int grt_render_accum_pm(grt_scene* scene, const grt_opts* opts, const grt_pmap* global_map,
                        const grt_pmap* caustic_map, float* rgb_sum, int32_t samples);
int grt_render_accum_pm_dev(grt_scene* scene, const grt_opts* opts, const grt_pmap* global_map,
                            const grt_pmap* caustic_map, int32_t shard, int32_t n_shards,
                            int32_t samples, float* d_rgb_sum, void* d_workspace,
                            uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream);
uint64_t grt_render_pm_workspace_bytes(const grt_scene* scene, const grt_opts* opts, int32_t shard,
                                       int32_t n_shards, int32_t samples);

/* ---- photon-map view ------------------------------------------------------------------- */

/* Scene.irradiance (scene.go:184-232; no caller in the reference) as a render mode: Scene.Render with
 * irradiance(pmap, Cam.Ray(u,v), 1, rnd) where scene.go:96 calls trace().  Metal (f * irradiance of the
 * reflected ray when reflected.Dot(n) > 0, scene.go:207-214) and glass (scene.go:215-224) are followed up to
 * maxDepth; every other surface — lights and Lambert walls included — shows pmap.IrradianceEst(p, n, 0, 100)
 * (scene.go:228).  GRT_PMVIEW_FIRST_HIT is the debugging block of trace() (scene.go:261-271):
 * IrradianceEst(p, n, 0, 50) at the first hit.  Buffers as grt_render / grt_render_accum; one device per call. */
#define GRT_PMVIEW_IRRADIANCE 0
#define GRT_PMVIEW_FIRST_HIT 1
int grt_render_pmview(grt_scene* scene, const grt_opts* opts, const grt_pmap* pmap, int32_t mode, uint8_t* pixels,
                      int32_t pitch, int32_t samples);
int grt_render_accum_pmview(grt_scene* scene, const grt_opts* opts, const grt_pmap* pmap, int32_t mode, float* rgb_sum,
                            int32_t samples);

/* ---- misc ----------------------------------------------------------------------------- */

/* Uniform in [0,1) with 24 random bits: draw `dim` of sample `sample` of pixel `pixel`
 * (= y*W+x).  Replaces rnd.Float64() (scene.go:93-94, vec3.go:152,203-204). Host-side. */
double grt_rng_u01(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);

/* Number of intersect() calls executed by the last grt_render / grt_render_accum on this
 * thread, and the device time of its path-trace kernels in milliseconds. */
uint64_t grt_last_ray_count(void);
double grt_last_kernel_ms(void);

int grt_device_count(void);
const char* grt_last_error(void);
const char* grt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GRT_H */
