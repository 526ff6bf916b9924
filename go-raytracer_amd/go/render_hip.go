// +build hip

// render_hip.go — the cgo shim a maintainer adds to pkg/tracer so that Scene.Render
// (pkg/tracer/scene.go:79) runs on libgrt_hip.so.  Build the reference with `-tags hip`
// and rename the body of the stock Render to renderCPU; cmd/raytracer is unchanged.
//
// NOT compiled or run in this repository (the image has no Go toolchain): it is kept
// stdlib-only and small so it can be checked by reading.  The tested native host is
// ../host/tracer.hpp + main.cpp, which performs exactly these calls.
package tracer

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../csrc -lgrt_hip -Wl,-rpath,${SRCDIR}/../csrc
#include <stdlib.h>
#include "grt.h"
*/
import "C"

import (
	"log"
	"time"
	"unsafe"
)

const maxPrims = 1 << 26 // GRT_MAX_PRIMS of the library

func vec3(dst *[3]C.double, e [3]float64) {
	for k := 0; k < 3; k++ {
		dst[k] = C.double(e[k])
	}
}

func b2u8(b bool) C.uint8_t {
	if b {
		return 1
	}
	return 0
}

// Render replaces scene.go:79-136.  Same arguments, same pixel layout (B,G,R at
// pixels[y*pitch + x*bpp + {0,1,2}]), same two log lines.
func (scene Scene) Render(pixels []byte, pitch int, samples int) {
	log.Printf("Started rendering (%d samples)", samples)
	start := time.Now()

	// The type switch that flattens []Hitable into POD records; C memory so no Go pointer
	// is retained by C (cgo pointer rule).  One material record per object: no de-duplication
	// needed for correctness.
	n := len(scene.Objects)
	if n > maxPrims { // the library's own limit (GRT_MAX_PRIMS): fail here with the count, not with an index panic
		log.Fatalf("render_hip: %d objects; the library takes at most %d", n, maxPrims)
	}
	primMem := C.calloc(C.size_t(n+1), C.size_t(unsafe.Sizeof(C.grt_prim{})))
	matMem := C.calloc(C.size_t(n+1), C.size_t(unsafe.Sizeof(C.grt_material{})))
	if primMem == nil || matMem == nil {
		log.Fatal("render_hip: out of memory for the object records")
	}
	defer C.free(primMem)
	defer C.free(matMem)
	// Slices over the C arrays with their true length (go.mod says go 1.15: unsafe.Slice is 1.17, so the
	// array type only names an upper bound no scene reaches and the slice expression sets len and cap to n+1)
	prims := (*[maxPrims + 1]C.grt_prim)(primMem)[: n+1 : n+1]
	mats := (*[maxPrims + 1]C.grt_material)(matMem)[: n+1 : n+1]
	for i, o := range scene.Objects {
		switch h := o.(type) {
		case Sphere:
			prims[i].kind = C.GRT_KIND_SPHERE
			vec3(&prims[i].a, h.Center.E)
			prims[i].radius = C.double(h.Radius)
		case AABB:
			prims[i].kind = C.GRT_KIND_AABB
			vec3(&prims[i].a, h.MinBound.E)
			vec3(&prims[i].b, h.MaxBound.E)
		default: // e.g. List (list.go:6) has no flat form
			log.Fatalf("render_hip: object %d has unsupported type %T", i, o)
		}
		m := o.Material()
		prims[i].material = C.int32_t(i)
		vec3(&mats[i].color, m.Color.E)
		mats[i].refr_index, mats[i].reflectivity = C.double(m.RefrIndex), C.double(m.Reflectivity)
		mats[i].roughness, mats[i].emittance = C.double(m.Roughness), C.double(m.Emittance)
		mats[i].transparent, mats[i].lambert, mats[i].normal = b2u8(m.Transparent), b2u8(m.Lambert), b2u8(m.Normal)
	}
	var desc C.grt_scene_desc
	desc.width, desc.height = C.int32_t(scene.W), C.int32_t(scene.H)
	desc.max_depth, desc.bias = C.int32_t(scene.maxDepth), C.double(bias)
	vec3(&desc.cam.origin, scene.Cam.origin.E)
	vec3(&desc.cam.lower_left, scene.Cam.lowerLeft.E)
	vec3(&desc.cam.horizontal, scene.Cam.horizontal.E)
	vec3(&desc.cam.vertical, scene.Cam.vertical.E)

	var h *C.grt_scene
	if C.grt_scene_create(&desc, &prims[0], C.int32_t(n), &mats[0], C.int32_t(n), &h) != 0 {
		log.Fatal("grt_scene_create: ", C.GoString(C.grt_last_error()))
	}
	defer C.grt_scene_destroy(h)
	opts := C.grt_opts{seed: C.uint64_t(time.Now().Unix()), n_devices: C.int32_t(C.grt_device_count())} // scene.go:114
	log.Printf("Rendering scene")
	if C.grt_render(h, &opts, (*C.uint8_t)(unsafe.Pointer(&pixels[0])), C.int32_t(pitch), C.int32_t(samples)) != 0 {
		log.Fatal("grt_render: ", C.GoString(C.grt_last_error()))
	}
	log.Printf("Rendering took %s", time.Since(start))
}
