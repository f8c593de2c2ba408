// Scene file -> one frame on N GPUs of this node from one process -> PNG.  The C++ twin of `bench.py --gpus N`:
//   multi_gpu <scene.test> [--gpus N] [--size W H] [--out image.png] [--frames K] [--check]
// mirt_node_create makes one scene handle per GPU and an RCCL communicator over them; a frame is every GPU's interleaved
// 64x16 tiles, one grouped ncclSend/ncclRecv step into GPU 0 and the reassembly there (include/mirt_node.h).
// --check renders the same frame on GPU 0 alone through mirt_render and compares byte for byte.
// What it replaces in the reference: the loop of createRender, RayTracer/main.cpp:137-163, and the save behind it (:165-169).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../include/mirt_host.h"
#include "../include/mirt_node.h"

static int die(const char* what) {
	fprintf(stderr, "%s: %s\n", what, mirt_last_error());
	return 1;
}

int main(int argc, char** argv) {
	if (argc < 2) {
		fprintf(stderr, "usage: %s <scene.test> [--gpus N] [--size W H] [--out image.png] [--frames K] [--check]\n", argv[0]);
		return 2;
	}
	int gpus = 0, frames = 8;
	uint32_t W = 0, H = 0;
	bool check = false;
	std::string out;
	for (int i = 2; i < argc; ++i) {
		if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--size") && i + 2 < argc) { W = (uint32_t)atoi(argv[i + 1]); H = (uint32_t)atoi(argv[i + 2]); i += 2; }
		else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
		else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--check")) check = true;
	}
	if (gpus <= 0 && hipGetDeviceCount(&gpus) != hipSuccess) gpus = 0;
	if (gpus <= 0) { fprintf(stderr, "no GPU visible\n"); return 1; }

	mirt_host_scene* host = nullptr;                      // Scene scene(path): parser + Partition build (main.cpp:107)
	if (mirt_host_scene_load(argv[1], 0, &host)) return die("scene");
	mirt_scene_desc desc;
	uint32_t w0 = 0, h0 = 0;
	if (mirt_host_scene_desc(host, &desc, &w0, &h0)) return die("scene");
	if (!W) { W = w0; H = h0; }
	if (out.empty()) out = mirt_host_scene_output_name(host);

	mirt_node_ctx* node = nullptr;
	if (mirt_node_create(&desc, nullptr, (uint32_t)gpus, &node)) return die("mirt_node_create");
	mirt_render_params p;
	memset(&p, 0, sizeof p);
	p.struct_size = sizeof p;
	p.width = W;
	p.height = H;
	p.feature_bits = MIRT_F_ALL;
	p.debug_mode = MIRT_DEBUG_NONE;
	p.trace_mode = MIRT_TRACE_EXACT;
	p.tile_stride = 1;
	std::vector<uint8_t> img((size_t)W * H * 3);
	mirt_stats st;
	if (mirt_node_render(node, &p, img.data(), &st)) return die("mirt_node_render");   // warms up, and is the image that is saved
	const double rays = (double)(st.primary_rays + st.shadow_rays + st.reflection_rays);

	// frames in flight: one slot each, device buffers only
	const int inFlight = frames < MIRT_NODE_SLOTS ? (frames > 0 ? frames : 1) : MIRT_NODE_SLOTS;
	void* d = nullptr;
	for (int k = 0; k < inFlight; ++k) if (mirt_node_render_device(node, &p, (uint32_t)k, &d)) return die("mirt_node_render_device");
	for (int k = 0; k < inFlight; ++k) if (mirt_node_sync(node, (uint32_t)k)) return die("mirt_node_sync");
	const auto t0 = std::chrono::steady_clock::now();
	for (int k = 0; k < frames; ++k) if (mirt_node_render_device(node, &p, (uint32_t)(k % inFlight), &d)) return die("mirt_node_render_device");
	for (int k = 0; k < inFlight; ++k) if (mirt_node_sync(node, (uint32_t)k)) return die("mirt_node_sync");
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	uint32_t n = 0;
	const int rccl = mirt_node_uses_rccl(node, &n);
	printf("{\"gpus\": %u, \"exchange\": \"%s\", \"width\": %u, \"height\": %u, \"frames\": %d, \"frames_in_flight\": %d, \"ms_per_frame\": %.4f, \"Mrays_per_s\": %.1f, "
	       "\"rays_per_frame\": {\"primary\": %llu, \"shadow\": %llu, \"reflection\": %llu}",
	       n, rccl ? "RCCL ncclSend/ncclRecv" : "device-to-device copies (rehearsal)", W, H, frames, inFlight, sec / frames * 1e3, rays * frames / sec / 1e6,
	       (unsigned long long)st.primary_rays, (unsigned long long)st.shadow_rays, (unsigned long long)st.reflection_rays);

	int rc = 0;
	if (check) {
		mirt_scene* one = nullptr;
		if (mirt_scene_create(&desc, 0, &one)) return die("mirt_scene_create");
		std::vector<uint8_t> ref(img.size());
		mirt_stats s1;
		if (mirt_render(one, &p, ref.data(), &s1)) return die("mirt_render");
		size_t diff = 0;
		for (size_t i = 0; i < ref.size(); ++i) diff += ref[i] != img[i];
		const bool same = diff == 0 && s1.primary_rays == st.primary_rays && s1.shadow_rays == st.shadow_rays && s1.reflection_rays == st.reflection_rays;
		printf(", \"check\": \"%s\", \"bytes_differing\": %zu", same ? "image and ray counts equal one GPU's" : "MISMATCH", diff);
		if (!same) rc = 3;
		mirt_scene_destroy(one);
	}
	printf("}\n");
	if (mirt_write_png(out.c_str(), img.data(), W, H, 0)) return die("mirt_write_png");      // Renderer::createImage (Renderer.cpp:12-15)
	mirt_node_destroy(node);
	mirt_host_scene_destroy(host);
	return rc;
}
