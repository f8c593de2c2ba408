// PNG output of a rendered frame: what Renderer::createImage does through FreeImage (RayTracer/Renderer.cpp:12-15:
// FreeImage_ConvertFromRawBits(pixels, w, h, w*3, 24, masks, topdown = true) + FreeImage_Save(FIF_PNG)), i.e. an 8-bit
// RGB PNG whose pixel (row i, column j) is (R, G, B) = (pixels[3*(i*w+j)+2], [+1], [+0]), top row first.
// The file is not byte-identical to FreeImage's (compression settings are free in PNG); it decodes to the same pixels.
//
// At device speeds the deflate of a 25-100 MB frame is the slow step, and most of its time goes to the rows that show
// something (a black stripe deflates ten times faster than one full of shaded pixels).  So the image is cut into MANY
// stripes of about 384 KB of PNG rows, handed to the worker threads one at a time in image order: every stripe becomes a
// raw deflate stream of its own rows closed by a sync flush, the streams are concatenated behind one zlib header, and the
// Adler-32 of the whole is combined from the stripes' (adler32_combine).  A stripe whose pixel bytes are all zero takes its
// stream from a cache (it depends only on the stripe's shape).
//
// Two producers of PNG rows (filter byte + R,G,B per pixel) feed the same encoder: mirt_write_png makes them on the host,
// stripe by stripe inside the workers, from the caller's B,G,R buffer; mirt_render_png (mirt_api.hip) has the GPU write
// them behind the frame, Up-filtered, and brings them over in bands -- a worker waits for the band that holds its stripe,
// so deflate runs while later bands are still on the bus.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/mirt_host.h"
#include "mirt_internal.h"
#include "mirt_pool.h"

namespace {

void put32(std::vector<uint8_t>& v, uint32_t x) {
	v.push_back((uint8_t)(x >> 24));
	v.push_back((uint8_t)(x >> 16));
	v.push_back((uint8_t)(x >> 8));
	v.push_back((uint8_t)x);
}

bool write_chunk(FILE* f, const char type[4], const uint8_t* data, size_t len) {
	uint8_t head[8] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3] };
	uLong crc = crc32(0L, head + 4, 4);
	size_t done = 0;
	while (done < len) { // crc32 takes a 32-bit length
		const size_t n = len - done < (1u << 30) ? len - done : (1u << 30);
		crc = crc32(crc, data + done, (uInt)n);
		done += n;
	}
	const uint8_t tail[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
	return fwrite(head, 1, 8, f) == 8 && (len == 0 || fwrite(data, 1, len, f) == len) && fwrite(tail, 1, 4, f) == 4;
}

struct Stripe {
	uint32_t row0 = 0, row1 = 0;
	std::vector<uint8_t> own;      // this stripe's deflate stream, unless it is a cached one
	const uint8_t* out = nullptr;
	size_t outBytes = 0;
	uLong adler = 1;
	size_t rawBytes = 0;
	int err = Z_OK;
};

uLong adler_of(const uint8_t* p, size_t n) {
	uLong a = adler32(0L, Z_NULL, 0);
	for (size_t done = 0; done < n;) {
		const size_t k = n - done < (1u << 30) ? n - done : (1u << 30);
		a = adler32(a, p + done, (uInt)k);
		done += k;
	}
	return a;
}

// raw[0, n) -> raw deflate stream in `out`, sync-flushed (or finished if last).  Returns a zlib code.
int deflate_raw(const uint8_t* raw, size_t n, bool last, int strategy, std::vector<uint8_t>& out, size_t& outBytes) {
	z_stream z;
	memset(&z, 0, sizeof z);
	int rc = deflateInit2(&z, 1, Z_DEFLATED, -15, 8, strategy);
	if (rc != Z_OK) return rc;
	const size_t bound = n + n / 1000 + 64 + 5 * (n / 16000 + 2);
	if (out.size() < bound) out.resize(bound);
	size_t in = 0, o = 0;
	for (;;) {
		const size_t nin = n - in < (1u << 30) ? n - in : (1u << 30);
		const bool lastPiece = in + nin == n;
		z.next_in = const_cast<Bytef*>(raw + in);
		z.avail_in = (uInt)nin;
		do {
			if (o == out.size()) out.resize(out.size() * 2);
			const size_t room = out.size() - o < (1u << 30) ? out.size() - o : (1u << 30);
			z.next_out = out.data() + o;
			z.avail_out = (uInt)room;
			rc = deflate(&z, lastPiece ? (last ? Z_FINISH : Z_SYNC_FLUSH) : Z_NO_FLUSH);
			o += room - z.avail_out;
			if (rc == Z_STREAM_ERROR) { deflateEnd(&z); return rc; }
		} while (z.avail_out == 0 || (lastPiece && last && rc != Z_STREAM_END));
		in += nin;
		if (lastPiece) break;
	}
	deflateEnd(&z);
	outBytes = o;
	return Z_OK;
}

// Deflate streams of stripes whose pixel bytes are all zero, by shape: (rows, line bytes, filter byte, last stripe).
struct ZeroStream {
	std::vector<uint8_t> bytes;
	uLong adler;
};
const ZeroStream* zero_stream(uint32_t rows, size_t line, uint8_t filter, bool last, int strategy) {
	static std::mutex mu;
	static std::map<std::vector<uint64_t>, ZeroStream*> cache;
	const std::vector<uint64_t> key = {rows, (uint64_t)line, filter, last ? 1u : 0u, (uint64_t)strategy};
	std::lock_guard<std::mutex> lk(mu);
	auto it = cache.find(key);
	if (it != cache.end()) return it->second;
	if (cache.size() > 64) { // (shapes of a handful of image sizes; never grows in a render loop)
		for (auto& kv : cache) delete kv.second;
		cache.clear();
	}
	std::vector<uint8_t> raw((size_t)rows * line, 0);
	for (uint32_t i = 0; i < rows; ++i) raw[(size_t)i * line] = filter;
	ZeroStream* z = new ZeroStream();
	size_t n = 0;
	if (deflate_raw(raw.data(), raw.size(), last, strategy, z->bytes, n) != Z_OK) { delete z; return nullptr; }
	z->bytes.resize(n);
	z->adler = adler_of(raw.data(), raw.size());
	cache[key] = z;
	return z;
}

} // namespace

// The encoder both entry points share.  rows_of(stripe, row0, row1, scratch, zero) returns the PNG rows [row0, row1) --
// (1 + 3 * width) bytes each, filter byte first -- as soon as they exist: either a pointer into memory that holds them, or
// `scratch` after filling it; *zero = every pixel byte of the stripe is 0 (the rows need not be looked at then; their filter
// byte is `zeroFilter`); nullptr = failure.  It is called from several threads at once, for different stripes.
int mirt_png_encode(const char* path, uint32_t width, uint32_t height, uint32_t threads, uint8_t zeroFilter, int strategy,
                    const std::function<const uint8_t*(uint32_t, uint32_t, uint32_t, std::vector<uint8_t>&, bool*)>& rows_of) {
	if (!path) return mirt_set_error(MIRT_E_INVALID, "path is NULL");
	if (width == 0 || height == 0 || width > 0x7fffffffu / 4 || height > 0x7fffffffu) return mirt_set_error(MIRT_E_INVALID, "image size %ux%u cannot be written as PNG", width, height);
	try {
		const size_t line = 1 + (size_t)width * 3;
		const uint32_t nStripes = mirt_png_stripes(width, height);
		const uint32_t per = (height + nStripes - 1) / nStripes;
		std::vector<Stripe> stripes(nStripes);
		for (uint32_t k = 0; k < nStripes; ++k) {
			stripes[k].row0 = k * per;
			stripes[k].row1 = (k + 1) * per < height ? (k + 1) * per : height;
		}
		auto work = [&](uint32_t k) {
			Stripe& s = stripes[k];
			const bool last = k + 1 == nStripes;
			static thread_local std::vector<uint8_t> scratch;
			bool zero = false;
			const uint8_t* raw = rows_of(k, s.row0, s.row1, scratch, &zero);
			if (!raw && !zero) { s.err = Z_DATA_ERROR; return; }
			s.rawBytes = (size_t)(s.row1 - s.row0) * line;
			if (zero) {
				const ZeroStream* z = zero_stream(s.row1 - s.row0, line, zeroFilter, last, strategy);
				if (z) { s.out = z->bytes.data(); s.outBytes = z->bytes.size(); s.adler = z->adler; return; }
				if (!raw) { s.err = Z_MEM_ERROR; return; }
			}
			s.adler = adler_of(raw, s.rawBytes);
			s.err = deflate_raw(raw, s.rawBytes, last, strategy, s.own, s.outBytes);
			s.out = s.own.data();
		};
		// `threads` = 1 keeps everything on the calling thread; otherwise the library's pool (MIRT_THREADS sets its size)
		if (threads == 1) for (uint32_t k = 0; k < nStripes; ++k) work(k);
		else MirtPool::get().run(nStripes, work);
		uLong adler = adler32(0L, Z_NULL, 0);
		for (uint32_t k = 0; k < nStripes; ++k) {
			if (stripes[k].err != Z_OK) return mirt_set_error(stripes[k].err == Z_DATA_ERROR ? MIRT_E_DEVICE : MIRT_E_NOMEM, "PNG stripe %u failed (%d)", k, stripes[k].err);
			adler = k == 0 ? stripes[0].adler : adler32_combine(adler, stripes[k].adler, (z_off_t)stripes[k].rawBytes);
		}
		FILE* f = fopen(path, "wb");
		if (!f) return mirt_set_error(MIRT_E_INVALID, "cannot open %s for writing", path);
		static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
		std::vector<uint8_t> ihdr;
		put32(ihdr, width);
		put32(ihdr, height);
		ihdr.push_back(8); // bit depth
		ihdr.push_back(2); // colour type: RGB
		ihdr.push_back(0);
		ihdr.push_back(0);
		ihdr.push_back(0);
		bool ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr.data(), ihdr.size());
		// one zlib stream: header, the stripes' streams gathered into IDAT chunks of about 1 MB, trailer
		std::vector<uint8_t> idat = { 0x78, 0x01 };
		for (uint32_t k = 0; ok && k < nStripes; ++k) {
			const Stripe& s = stripes[k];
			for (size_t done = 0; ok && done < s.outBytes;) {
				const size_t n = s.outBytes - done < (1u << 24) ? s.outBytes - done : (1u << 24);
				idat.insert(idat.end(), s.out + done, s.out + done + n);
				done += n;
				if (idat.size() >= (1u << 20)) { ok = write_chunk(f, "IDAT", idat.data(), idat.size()); idat.clear(); }
			}
		}
		put32(idat, (uint32_t)adler);
		ok = ok && write_chunk(f, "IDAT", idat.data(), idat.size()) && write_chunk(f, "IEND", nullptr, 0);
		ok = (fclose(f) == 0) && ok;
		if (!ok) return mirt_set_error(MIRT_E_INVALID, "short write to %s", path);
		return MIRT_OK;
	} catch (const std::bad_alloc&) {
		return mirt_set_error(MIRT_E_NOMEM, "out of host memory while encoding %s", path);
	} catch (const std::exception& e) {
		return mirt_set_error(MIRT_E_INVALID, "PNG encoding failed: %s", e.what());
	}
}

// Stripes of about 384 KB of PNG rows, whole rows, at most MIRT_PNG_MAX_STRIPES.
uint32_t mirt_png_stripes(uint32_t width, uint32_t height) {
	const size_t line = 1 + (size_t)width * 3;
	uint64_t rows = (384u << 10) / line;
	if (rows < 1) rows = 1;
	uint64_t n = (height + rows - 1) / rows;
	if (n > MIRT_PNG_MAX_STRIPES) n = MIRT_PNG_MAX_STRIPES;
	if (n < 1) n = 1;
	// (rows per stripe = ceil(height / n); the last stripe may be shorter, and n is reduced until none is empty)
	const uint64_t per = (height + n - 1) / n;
	return (uint32_t)((height + per - 1) / per);
}

extern "C" int mirt_write_png(const char* path, const uint8_t* bgr, uint32_t width, uint32_t height, uint32_t threads) {
	if (!path || !bgr) return mirt_set_error(MIRT_E_INVALID, "path or pixel buffer is NULL");
	const size_t line = 1 + (size_t)width * 3;
	// rows [row0,row1) -> filter byte 0 (None) + R,G,B per pixel, made by the worker that deflates them
	return mirt_png_encode(path, width, height, threads, 0, Z_DEFAULT_STRATEGY,
	                       [&](uint32_t, uint32_t row0, uint32_t row1, std::vector<uint8_t>& scratch, bool* zero) -> const uint8_t* {
		scratch.resize((size_t)(row1 - row0) * line);
		uint8_t any = 0;
		for (uint32_t i = row0; i < row1; ++i) {
			uint8_t* dst = scratch.data() + (size_t)(i - row0) * line;
			const uint8_t* src = bgr + (size_t)i * width * 3;
			*dst++ = 0; // filter type None
			for (uint32_t j = 0; j < width; ++j) {
				const uint8_t b = src[3 * j], g = src[3 * j + 1], r = src[3 * j + 2];
				dst[3 * j] = r;
				dst[3 * j + 1] = g;
				dst[3 * j + 2] = b;
				any |= (uint8_t)(b | g | r);
			}
		}
		*zero = any == 0;
		return scratch.data();
	});
}
