// grt_photon.hip — photon map emission, kd-tree build and kNN gather (config C4).
#include <hip/hip_runtime.h>

#include "../../include/grt.h"
#include "grt_internal.h"

#define GRT_API extern "C" __attribute__((visibility("default")))

GRT_API int grt_photon_build(grt_scene*, const grt_opts*, int32_t, int32_t, grt_pmap**) {
  return grt_fail(GRT_EINVAL, "grt_photon_build: not built yet");
}
GRT_API int32_t grt_photon_count(const grt_pmap*) { return 0; }
GRT_API int grt_photon_read(const grt_pmap*, float*, float*, uint8_t*) {
  return grt_fail(GRT_EINVAL, "grt_photon_read: not built yet");
}
GRT_API int grt_photon_load(const grt_opts*, const float*, const float*, const uint8_t*, int32_t, grt_pmap**) {
  return grt_fail(GRT_EINVAL, "grt_photon_load: not built yet");
}
GRT_API int grt_photon_gather(const grt_pmap*, const float*, const float*, int32_t, float, int32_t, float*) {
  return grt_fail(GRT_EINVAL, "grt_photon_gather: not built yet");
}
GRT_API int grt_photon_gather_dev(const grt_pmap*, const float*, const float*, int32_t, float, int32_t, float*, void*) {
  return grt_fail(GRT_EINVAL, "grt_photon_gather_dev: not built yet");
}
GRT_API void grt_photon_free(grt_pmap*) {}
