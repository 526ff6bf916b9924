// tracer.hpp — native (C++) host side above the C ABI: a mirror of the reference's
// `pkg/tracer` + `pkg/geom` + `pkg/util` interface for the render path.  The reference's host
// is Go; no Go toolchain exists in this image, so this is the host that is built and tested
// (the cgo shim a maintainer would add instead is in ../go/render_hip.go).
//
//   geom::NewVec3                                   pkg/geom/vec3.go:16
//   tracer::NewColor                                pkg/tracer/color.go:15
//   tracer::{Normal,Diffuse,Lambert,Metalic,Dielectric,Light}Material   material.go:15-41
//   tracer::NewSphere / NewAABB                     sphere.go:17 / aabb.go:18
//   tracer::NewCamera                               camera.go:16-30
//   tracer::NewPhotonMap                            photon_mapping.go:97
//   tracer::NewScene, Scene::Render                 scene.go:35, scene.go:79
//   util::SaveToImage                               pkg/util/image.go:11
//
// Nothing here computes a pixel: Scene::Render flattens the objects (the type switch the cgo
// shim performs) and calls grt_render.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/grt.h"

namespace geom {
struct Vec3 {
  double E[3];
  double X() const { return E[0]; }
  double Y() const { return E[1]; }
  double Z() const { return E[2]; }
  Vec3 Plus(Vec3 b) const { return {{E[0] + b.E[0], E[1] + b.E[1], E[2] + b.E[2]}}; }
  Vec3 Minus(Vec3 b) const { return {{E[0] - b.E[0], E[1] - b.E[1], E[2] - b.E[2]}}; }
  Vec3 Scale(double n) const { return {{E[0] * n, E[1] * n, E[2] * n}}; }
  double LenSq() const { return E[0] * E[0] + E[1] * E[1] + E[2] * E[2]; }
  Vec3 Cross(Vec3 b) const {
    return {{E[1] * b.E[2] - E[2] * b.E[1], E[2] * b.E[0] - E[0] * b.E[2], E[0] * b.E[1] - E[1] * b.E[0]}};
  }
  Vec3 Unit() const { return Scale(1.0 / std::sqrt(LenSq())); }  // vec3.go:92-95
};
inline Vec3 NewVec3(double x, double y, double z) { return {{x, y, z}}; }
}  // namespace geom

namespace tracer {
using geom::Vec3;

struct Color {
  double E[3];
  double R() const { return E[0]; }
  double G() const { return E[1]; }
  double B() const { return E[2]; }
};
inline Color NewColor(double r, double g, double b) { return {{r, g, b}}; }

struct Material {  // material.go:4-13
  Color Color_;
  double RefrIndex, Reflectivity, Roughness, Emittance;
  bool Transparent, Lambert, Normal;
};
inline Material NormalMaterial() { return {NewColor(0, 0, 0), 1, 0, 0, 0, false, false, true}; }
inline Material DiffuseMaterial(Color c) { return {c, 1, 0, 0, 0, false, false, false}; }
inline Material LambertMaterial(Color albedo) { return {albedo, 1, 0, 0, 0, false, true, false}; }
inline Material MetalicMaterial(Color albedo, double reflectivity, double roughness) {
  return {albedo, 1, reflectivity, roughness, 0, false, false, false};
}
inline Material DielectricMaterial(double index) { return {NewColor(0, 0, 0), index, 0, 0, 0, true, false, false}; }
inline Material LightMaterial(Color intensity, double emittance) {
  return {intensity, 1, 0, 0, emittance, false, false, false};
}

// hitable.go:10-14 — Hit() itself lives on the GPU; the host keeps Material()/Pos() and the
// flattening into the boundary record.
struct Hitable {
  virtual ~Hitable() {}
  virtual Material GetMaterial() const = 0;
  virtual Vec3 Pos() const = 0;
  virtual bool Flatten(grt_prim* out) const = 0;  // false: no flat form (e.g. tracer.List)
};
struct Sphere : Hitable {  // sphere.go:10-14
  Vec3 Center;
  double Radius;
  Material Mat;
  Material GetMaterial() const override { return Mat; }
  Vec3 Pos() const override { return Center; }
  bool Flatten(grt_prim* p) const override {
    p->kind = GRT_KIND_SPHERE;
    for (int k = 0; k < 3; ++k) p->a[k] = Center.E[k], p->b[k] = 0;
    p->radius = Radius;
    return true;
  }
};
struct AABB : Hitable {  // aabb.go:10-15
  Vec3 MinBound, MaxBound, Center;
  Material Mat;
  Material GetMaterial() const override { return Mat; }
  Vec3 Pos() const override { return Center; }
  bool Flatten(grt_prim* p) const override {
    p->kind = GRT_KIND_AABB;
    for (int k = 0; k < 3; ++k) p->a[k] = MinBound.E[k], p->b[k] = MaxBound.E[k];
    p->radius = 0;
    return true;
  }
};
using HitablePtr = std::shared_ptr<Hitable>;
inline HitablePtr NewSphere(Vec3 center, double radius, Material mat) {
  auto s = std::make_shared<Sphere>();
  s->Center = center, s->Radius = radius, s->Mat = mat;
  return s;
}
inline HitablePtr NewAABB(Vec3 mn, Vec3 mx, Material mat) {
  auto b = std::make_shared<AABB>();
  b->MinBound = mn, b->MaxBound = mx, b->Center = mx.Plus(mn).Scale(0.5), b->Mat = mat;  // aabb.go:19
  return b;
}

struct Camera {  // camera.go:10-13
  Vec3 origin, horizontal, vertical, lowerLeft;
};
inline Camera NewCamera(Vec3 eye, Vec3 lookat, Vec3 vup, double vfov, double aspect) {  // camera.go:16-30
  const double pi = 3.14159265358979323846;
  double theta = vfov * pi / 180;
  double halfH = std::tan(theta / 2);
  double halfW = aspect * halfH;
  Vec3 w = eye.Minus(lookat).Unit();
  Vec3 u = vup.Cross(w).Unit();
  Vec3 v = u.Cross(w).Unit();
  Camera c;
  c.origin = eye;
  c.lowerLeft = c.origin.Minus(u.Scale(halfW)).Minus(v.Scale(halfH)).Minus(w);
  c.horizontal = u.Scale(2 * halfW);
  c.vertical = v.Scale(2 * halfH);
  return c;
}

struct PhotonMap {  // photon_mapping.go:84-94 (storage lives on the GPU: grt_pmap)
  int maxPhotons = 0;
  grt_pmap* handle = nullptr;
};
inline PhotonMap NewPhotonMap(int maxPhotons) {
  PhotonMap p;
  p.maxPhotons = maxPhotons;
  return p;
}

struct RenderOptions {       // not in the reference: the seed is time-based there (scene.go:114)
  uint64_t seed = 0;         // 0 = time(NULL), like the reference
  int first_device = 0, n_devices = 1;
};

class Scene {  // scene.go:17-27
 public:
  int W = 0, H = 0;
  Camera Cam;
  std::vector<HitablePtr> Objects;
  PhotonMap* globalPmap = nullptr;
  PhotonMap* causticPmap = nullptr;
  int maxDepth = 6;  // scene.go:62
  RenderOptions options;

  ~Scene() {
    if (handle_) grt_scene_destroy(handle_);
  }
  Scene() = default;
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;
  Scene(Scene&& o) noexcept { *this = std::move(o); }
  Scene& operator=(Scene&& o) noexcept {
    W = o.W, H = o.H, Cam = o.Cam, Objects = std::move(o.Objects), globalPmap = o.globalPmap;
    causticPmap = o.causticPmap, maxDepth = o.maxDepth, options = o.options, handle_ = o.handle_;
    o.handle_ = nullptr;
    return *this;
  }

  // scene.go:79.  pixels: H*pitch bytes, B,G,R at pixels[y*pitch + x*bpp + {0,1,2}], bpp = pitch/W.
  void Render(uint8_t* pixels, int pitch, int samples) {
    std::fprintf(stderr, "Started rendering (%d samples)\n", samples);  // scene.go:80
    std::clock_t c0 = std::clock();
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    Stage();
    grt_opts o{};
    o.seed = options.seed ? options.seed : (uint64_t)std::time(nullptr);
    o.first_device = options.first_device;
    o.n_devices = options.n_devices;
    if (grt_render(handle_, &o, pixels, pitch, samples) != GRT_OK) {
      std::fprintf(stderr, "grt_render: %s\n", grt_last_error());  // log.Fatal in the Go shim
      std::exit(1);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)c0;
    std::fprintf(stderr, "Rendering took %.3fs (GPU kernels %.3f ms, %.3g rays)\n",  // scene.go:135
                 (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), grt_last_kernel_ms(),
                 (double)grt_last_ray_count());
  }

  grt_scene* Handle() {
    Stage();
    return handle_;
  }

 private:
  grt_scene* handle_ = nullptr;

  void Stage() {
    if (handle_) return;
    std::vector<grt_prim> prims(Objects.size());
    std::vector<grt_material> mats;
    for (size_t i = 0; i < Objects.size(); ++i) {
      if (!Objects[i]->Flatten(&prims[i])) throw std::runtime_error("object has no flat form (unsupported Hitable)");
      Material m = Objects[i]->GetMaterial();
      grt_material r{};
      for (int k = 0; k < 3; ++k) r.color[k] = m.Color_.E[k];
      r.refr_index = m.RefrIndex, r.reflectivity = m.Reflectivity, r.roughness = m.Roughness;
      r.emittance = m.Emittance, r.transparent = m.Transparent, r.lambert = m.Lambert, r.normal = m.Normal;
      int idx = -1;
      for (size_t j = 0; j < mats.size() && idx < 0; ++j)
        if (!std::memcmp(&mats[j], &r, sizeof r)) idx = (int)j;
      if (idx < 0) {
        idx = (int)mats.size();
        mats.push_back(r);
      }
      prims[i].material = idx;
    }
    grt_scene_desc d{};
    d.width = W, d.height = H, d.max_depth = maxDepth, d.bias = 0.001;  // scene.go:14
    for (int k = 0; k < 3; ++k) {
      d.cam.origin[k] = Cam.origin.E[k], d.cam.lower_left[k] = Cam.lowerLeft.E[k];
      d.cam.horizontal[k] = Cam.horizontal.E[k], d.cam.vertical[k] = Cam.vertical.E[k];
    }
    if (grt_scene_create(&d, prims.data(), (int32_t)prims.size(), mats.data(), (int32_t)mats.size(), &handle_) != GRT_OK)
      throw std::runtime_error(std::string("grt_scene_create: ") + grt_last_error());
  }
};

inline Scene NewScene(int width, int height, Camera cam, std::vector<HitablePtr> objects, PhotonMap* globalPmap,
                      PhotonMap* causticPmap) {  // scene.go:35-64
  Scene s;
  s.W = width, s.H = height, s.Cam = cam, s.Objects = std::move(objects);
  s.globalPmap = globalPmap, s.causticPmap = causticPmap;
  return s;
}
}  // namespace tracer

namespace util {
// pkg/util/image.go:11-39: BGRx bytes -> 8-bit RGBA PNG (A = 255).  zlib does the deflate.
void SaveToImage(const std::string& name, int width, int height, const std::vector<uint8_t>& pixels);
}  // namespace util
