// main.cpp — the cmd/raytracer driver (cmd/raytracer/main.go) over libgrt_hip.so.
// Same flags (-w, -s, -p, -o), same scene literal (main.go:32-59), same output path
// ("output/" + name).  The SDL window of main.go:70-121 is display-only and not rebuilt:
// without -o this driver renders and prints the image statistics instead.
// Extra flags (not in the reference): -a aspect (main.go:27 hard-codes 1.0), -seed, -g devices.
#include <zlib.h>

#include <cstring>
#include <string>
#include <sys/stat.h>

#include "tracer.hpp"

namespace util {
static void put32(std::vector<uint8_t>& v, uint32_t x) {
  for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}
static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  put32(out, (uint32_t)data.size());
  std::vector<uint8_t> body(type, type + 4);
  body.insert(body.end(), data.begin(), data.end());
  out.insert(out.end(), body.begin(), body.end());
  put32(out, (uint32_t)crc32(0L, body.data(), (uInt)body.size()));
}
void SaveToImage(const std::string& name, int width, int height, const std::vector<uint8_t>& pixels) {
  const int pitch = (int)(pixels.size() / (size_t)height), bpp = pitch / width;  // image.go:12-13
  std::vector<uint8_t> raw;
  raw.reserve((size_t)height * (1 + (size_t)width * 4));
  for (int y = 0; y < height; ++y) {
    raw.push_back(0);  // filter: none
    for (int x = 0; x < width; ++x) {
      const uint8_t* p = &pixels[(size_t)y * pitch + (size_t)x * bpp];
      raw.push_back(p[2]), raw.push_back(p[1]), raw.push_back(p[0]), raw.push_back(255);  // image.go:19-24
    }
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
    std::fprintf(stderr, "png: deflate failed\n");
    std::exit(1);
  }
  z.resize(zlen);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
  put32(ihdr, (uint32_t)width), put32(ihdr, (uint32_t)height);
  ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});  // 8-bit RGBA (image.NRGBA)
  chunk(out, "IHDR", ihdr), chunk(out, "IDAT", z), chunk(out, "IEND", {});
  FILE* f = std::fopen(name.c_str(), "wb");
  if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size() || std::fclose(f) != 0) {
    std::perror(name.c_str());  // log.Fatal(err), image.go:28-36
    std::exit(1);
  }
  std::fprintf(stderr, "Image %s saved\n", name.c_str());  // image.go:38
}
}  // namespace util

int main(int argc, char** argv) {
  int width = 640, samples = 8, nphotons = 100000, devices = 1;  // main.go:21-23
  double aspect = 1.0;                                            // main.go:27
  unsigned long long seed = 0;
  std::string output;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "flag needs an argument: %s\n", a.c_str());
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "-w") width = std::atoi(next());
    else if (a == "-s") samples = std::atoi(next());
    else if (a == "-p") nphotons = std::atoi(next());
    else if (a == "-o") output = next();
    else if (a == "-a") aspect = std::atof(next());
    else if (a == "-seed") seed = std::strtoull(next(), nullptr, 10);
    else if (a == "-g") devices = std::atoi(next());
    else {
      std::fprintf(stderr,
                   "Usage of %s:\n  -o string\n    \tOutput image (PNG).\n  -p int\n    \tNumber of photons per photon map. "
                   "(default 100000)\n  -s int\n    \tAmount of samples per pixel. (default 8)\n  -w int\n    \tScene width. "
                   "(default 640)\n  -a float\n    \tAspect ratio W/H. (default 1)\n  -seed uint\n    \tRNG seed (default: time)\n"
                   "  -g int\n    \tGPUs to shard the frame over (default 1)\n",
                   argv[0]);
      return a == "-h" || a == "--help" ? 0 : 2;
    }
  }
  (void)nphotons;  // parsed and unused in the reference too (main.go:23 vs :57-58)
  const int height = (int)((double)width / aspect);  // main.go:28

  using namespace tracer;
  Material matRed = LambertMaterial(NewColor(0.65, 0.05, 0.05));  // main.go:32-37
  Material matGreen = LambertMaterial(NewColor(0.12, 0.45, 0.15));
  Material matWhite = LambertMaterial(NewColor(0.73, 0.73, 0.73));
  Material matLight = LightMaterial(NewColor(0.2, 0.2, 0.2), 10);
  Material matGlass = DielectricMaterial(1.53);
  Material matMirror = MetalicMaterial(NewColor(1, 1, 1), 1, 0);
  using geom::NewVec3;
  std::vector<HitablePtr> objects = {  // main.go:40-49
      NewAABB(NewVec3(113, 548, 127), NewVec3(443, 548.1, 432), matLight),
      NewAABB(NewVec3(0, 0, 0), NewVec3(555, 0.1, 555), matWhite),
      NewAABB(NewVec3(0, 555, 0), NewVec3(555, 555.1, 555), matWhite),
      NewAABB(NewVec3(0, 0, 555), NewVec3(555, 555, 555.1), matWhite),
      NewAABB(NewVec3(555, 0, 0), NewVec3(555.1, 555, 555), matRed),
      NewAABB(NewVec3(0, 0, 0), NewVec3(0.1, 555, 555), matGreen),
      NewSphere(NewVec3(278 + 110, 90, 227 + 120), 90, matMirror),
      NewSphere(NewVec3(278 - 110, 90, 227 - 40), 90, matGlass),
  };
  Camera cam = NewCamera(NewVec3(278, 273, -800), NewVec3(278, 278, 1), NewVec3(0, 1, 0), 40, aspect);  // main.go:51-55
  PhotonMap globalMap = NewPhotonMap(100000), causticsMap = NewPhotonMap(50000);                        // main.go:57-58
  Scene scene = NewScene(width, height, cam, objects, &globalMap, &causticsMap);                         // main.go:59
  scene.options.seed = seed;
  scene.options.n_devices = devices;

  const int bpp = 4, pitch = bpp * width;  // main.go:62-63 uses bpp*height, right only for squares
  std::vector<uint8_t> pixels((size_t)height * pitch);
  scene.Render(pixels.data(), pitch, samples);  // main.go:65
  if (!output.empty()) {
    mkdir("output", 0755);
    util::SaveToImage("output/" + output, width, height, pixels);  // main.go:66
    return 0;
  }
  double sum[3] = {0, 0, 0};
  for (size_t i = 0; i < pixels.size(); i += 4) sum[0] += pixels[i + 2], sum[1] += pixels[i + 1], sum[2] += pixels[i];
  const double n = (double)width * height;
  std::printf("no -o given: the SDL window of main.go:70-121 is out of scope; mean RGB = %.3f %.3f %.3f\n", sum[0] / n,
              sum[1] / n, sum[2] / n);
  return 0;
}
