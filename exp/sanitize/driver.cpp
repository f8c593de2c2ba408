// Host-side sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer on HOST code only; device code is never instrumented).
//
//   driver <scene files...>
//
// For every scene file: the library's loader parses it and builds the Partition tree (mirt_host.cpp), the PNG writer
// encodes a small pattern (mirt_png.cpp), and -- when built with -DWITH_DEVICE against the sanitized mirt_api.hip host
// code -- the scene is handed to the device, one small frame is rendered in every trace mode, a batch of rays is traced,
// the scene is saved to and loaded from a cache file, and everything is destroyed again.  The driver checks nothing but
// return codes: what it is for is the sanitizers' report (a non-zero exit status).  See exp/sanitize/run.sh.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/mirt.h"
#include "../../include/mirt_host.h"

#ifndef WITH_DEVICE
#include <stdarg.h>
// the loader reports through the device library's error text; without that library the driver holds it
static std::string g_text;
int mirt_set_error(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_text = buf;
	return code;
}
#endif

int main(int argc, char** argv) {
	int loaded = 0, refused = 0, frames = 0;
	for (int a = 1; a < argc; ++a) {
		mirt_host_scene* hs = nullptr;
		const int e = mirt_host_scene_load(argv[a], 4, &hs);
		if (e != MIRT_OK) {
			++refused;
			continue;
		}
		++loaded;
		mirt_scene_desc d;
		uint32_t w = 0, h = 0;
		if (mirt_host_scene_desc(hs, &d, &w, &h) != MIRT_OK) return 2;
		(void)mirt_host_scene_output_name(hs);
		uint64_t trunc = 0;
		(void)mirt_host_scene_truncated_nodes(hs, &trunc);
		{
			// the loader's second entrance: the same primitives handed over as arrays, tree built again
			mirt_host_scene* again = nullptr;
			if (mirt_host_scene_from_prims(&d, w, h, 3, &again) == MIRT_OK) mirt_host_scene_destroy(again);
		}
		{
			const uint32_t pw = 37 + (uint32_t)a % 9, ph = 11 + (uint32_t)a % 5;
			std::vector<uint8_t> img((size_t)pw * ph * 3);
			for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(i * 7 + a);
			const std::string path = std::string(argv[a]) + ".png";
			if (mirt_write_png(path.c_str(), img.data(), pw, ph, 1 + (uint32_t)a % 3) != MIRT_OK) return 3;
			remove(path.c_str());
		}
#ifdef WITH_DEVICE
		if (!trunc) {
			mirt_scene* s = nullptr;
			if (mirt_scene_create(&d, 0, &s) != MIRT_OK) {
				fprintf(stderr, "%s: mirt_scene_create: %s\n", argv[a], mirt_last_error());
				return 4;
			}
			const uint32_t fw = w < 96 ? w : 96, fh = h < 64 ? h : 64;
			std::vector<uint8_t> img((size_t)fw * fh * 3);
			for (uint32_t mode = 0; mode < 3; ++mode) {
				mirt_render_params p;
				memset(&p, 0, sizeof p);
				p.struct_size = sizeof p;
				p.width = fw;
				p.height = fh;
				p.feature_bits = MIRT_F_ALL;
				p.debug_mode = MIRT_DEBUG_NONE;
				p.trace_mode = mode;
				p.tile_stride = 1;
				mirt_stats st;
				const int r = mirt_render(s, &p, img.data(), &st);
				if (r == MIRT_OK) ++frames;
				else if (mode != MIRT_TRACE_REFWALK) { // (REFWALK refuses trees deeper than its stack)
					fprintf(stderr, "%s: mirt_render mode %u: %s\n", argv[a], mode, mirt_last_error());
					return 5;
				}
				if (mode == 0) {
					// a rank's share, packed tiles
					p.tile_w = 16;
					p.tile_h = 8;
					p.tile_stride = 3;
					p.first_tile = 1;
					(void)mirt_render(s, &p, img.data(), &st);
				}
			}
			{
				const uint64_t n = 257;
				std::vector<float> o(n * 3), dd(n * 3), dist(n), nrm(n * 3);
				std::vector<int32_t> ids(n);
				uint32_t x = 12345u + (uint32_t)a;
				for (size_t i = 0; i < o.size(); ++i) {
					x = x * 1664525u + 1013904223u;
					o[i] = (float)(x >> 8) / 16777216.0f * 8.0f - 4.0f;
					x = x * 1664525u + 1013904223u;
					dd[i] = (float)(x >> 8) / 16777216.0f * 2.0f - 1.0f;
				}
				dd[0] = dd[1] = dd[2] = 0.0f; // a lengthless direction takes the walk that follows the reference's recursion
				for (uint32_t mode = 0; mode < 2; ++mode)
					if (mirt_trace_rays(s, mode, n, o.data(), dd.data(), ids.data(), dist.data(), nrm.data()) != MIRT_OK) return 6;
				std::vector<int32_t> pid((size_t)fw * fh);
				std::vector<float> pd((size_t)fw * fh);
				if (mirt_trace_primary_ids(s, 0, fw, fh, pid.data(), pd.data()) != MIRT_OK) return 7;
			}
			{
				const std::string path = std::string(argv[a]) + ".cache";
				const uint32_t user[4] = {w, h, 1, 2};
				uint32_t back[4];
				if (mirt_scene_save(s, path.c_str(), user) != MIRT_OK) return 8;
				mirt_scene* t = nullptr;
				if (mirt_scene_load(path.c_str(), 0, &t, back) != MIRT_OK) return 9;
				mirt_scene_destroy(t);
				remove(path.c_str());
			}
			mirt_scene_destroy(s);
		}
#endif
		mirt_host_scene_destroy(hs);
	}
	printf("%d scene files loaded, %d refused, %d frames rendered\n", loaded, refused, frames);
	fflush(stdout);
	// (no static teardown: the sanitizer runtime of this toolchain aborts on an internal check when a worker thread of the
	// library's pool ends after the HIP runtime has unloaded -- an order of exit, not an access of this code)
	_exit(0);
}
