// grt_internal.h — shared between the translation units of libgrt_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/grt.h"

// The linear-scan kernels (the reference's own algorithm, scene.go:173) read every object
// for every ray; beyond this many objects a scene goes through the BVH path instead.
#define GRT_MAX_LINEAR_PRIMS 4096
#define GRT_BVH_AUTO_PRIMS 208  // GRT_ACCEL_AUTO takes the hierarchy above this (tools/bench_crossover.py: 9.1 vs 9.7 ms at 208, 10.2 vs 9.6 at 224 objects)
#define GRT_MAX_PRIMS (1 << 26)

int grt_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int grt_fail_exception(const char* entry_point);  // inside a catch (...): the current exception as GRT_EHOST + message

#define GRT_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return grt_fail(GRT_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Host-buffer entry points select a device (grt_opts.first_device, or the one a photon map lives on);
// the caller's current device is what it was when they return.  "_dev" entry points never switch: they
// work on the CURRENT device.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct DeviceScene {
  float4* prims = nullptr;
  float4* mats = nullptr;
  float4* lights = nullptr;
  double* prims64 = nullptr;
  uint4* bvh_nodes = nullptr;  // grt_bvh.h; NULL for linear-scan scenes
  float4* bvh_prims = nullptr;
#ifdef GRT_COUNT_DEPTH
  uint8_t* bvh_slot_depth = nullptr;
#endif
  int bvh_depth = 0;            // deepest leaf: the traversal stack holds at most this many entries + 1
  float bvh_grid[9] = {0, 0, 0, 1, 1, 1, 1, 1, 1};  // origin, cell size, 1 / cell size
  uint8_t* tile_live = nullptr;  // per 8x8 tile of the frame: can a camera ray reach the scene bounds?
};

// What the host-buffer entry points (grt_render, grt_render_accum) keep per scene and device between calls:
// device sums / chunk slabs / ray counter / packed bytes, pinned staging for the copies back, two events.
// A frame then costs no hipMalloc / hipFree.  `mu` is held for the length of a call: calls
// that share a scene and a device take turns.
struct HostRenderCache {
  std::mutex mu;
  float* d_sum = nullptr;
  size_t sum_cap = 0;
  void* d_ws = nullptr;
  size_t ws_cap = 0;
  uint64_t* d_rays = nullptr;
  uint8_t* d_bgr = nullptr;
  size_t bgr_cap = 0;
  float* h_sum = nullptr;  // pinned
  size_t h_sum_cap = 0;
  uint8_t* h_bgr = nullptr;  // pinned
  size_t h_bgr_cap = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

struct grt_scene {
  grt_scene_desc d;
  std::vector<grt_prim> prims;
  std::vector<grt_material> mats;
  std::vector<int> lights;    // scene.go:42-46
  std::vector<int> tobjects;  // scene.go:48-50
  double light_area = 0.0;
  bool use_bvh = false;
  bool has_diffuse = false;  // some object is plain diffuse (shadow rays, photon storage)
  int radiance_log2 = 0;     // ceil(log2) of a bound on what one path can add to its pixel (fixed-point sums of k_pathtrace_bvh)
  bool bvh_forced = false;  // hierarchy by request or by size: the parity kernel walks it too (else linear scan + FP64 re-test)
  std::mutex mu;
  std::map<int, DeviceScene> dev;  // staged copies, one per HIP device
  std::map<int, HostRenderCache> rcache;  // grt_render's buffers, one set per HIP device (node addresses are stable)
};

struct KScene;
// Stages `s` on the current device if needed and fills the kernel-side view.
int grt_scene_kernel_view(grt_scene* s, uint64_t seed, KScene* ks);
const grt_scene_desc* grt_scene_desc_of(const grt_scene* s);
// grt_api.hip: the render launch behind grt_render_accum*_dev, with the photon-request arrays of
// the photon-mapped trace() (all NULL / 0: off).
int grt_render_dev_internal(grt_scene* s, const grt_opts* o, int shard, int n_shards, int sample_first, int samples,
                            float* d_rgb_sum, void* d_workspace, uint64_t workspace_bytes, uint64_t* d_ray_count, void* stream,
                            float* pm_pos, float* pm_nrm, float* pm_w, int pm_batch, int pm_view = 0);
int grt_pm_accumulate_internal(const float* frame, const float* qpos, const float* qw, const float* irr_g, const float* irr_c,
                               int batch, size_t n_pixels, float* total, int first_batch, void* stream);
