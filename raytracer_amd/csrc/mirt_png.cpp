// PNG output of a rendered frame: what Renderer::createImage does through FreeImage (RayTracer/Renderer.cpp:12-15:
// FreeImage_ConvertFromRawBits(pixels, w, h, w*3, 24, masks, topdown = true) + FreeImage_Save(FIF_PNG)), i.e. an 8-bit
// RGB PNG whose pixel (row i, column j) is (R, G, B) = (pixels[3*(i*w+j)+2], [+1], [+0]), top row first.
// The file is not byte-identical to FreeImage's (compression settings are free in PNG); it decodes to the same pixels.
//
// At device speeds the deflate of a 25-100 MB frame is the slow step, so rows are compressed in independent stripes on
// CPU threads: every stripe is a raw deflate stream of its own rows closed by a sync flush, the stripes are concatenated
// behind one zlib header, and the Adler-32 of the whole is combined from the stripes' (adler32_combine).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <thread>
#include <vector>

#include "../../include/mirt_host.h"
#include "mirt_internal.h"

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
	uint32_t row0, row1;
	std::vector<uint8_t> out;
	uLong adler;
	size_t rawBytes;
	int err;
};

// rows [row0,row1) -> filter byte 0 + R,G,B per pixel -> raw deflate, sync-flushed (or finished if last)
void deflate_stripe(const uint8_t* bgr, uint32_t width, Stripe& s, bool last, int level) {
	const size_t line = 1 + (size_t)width * 3;
	std::vector<uint8_t> raw((size_t)(s.row1 - s.row0) * line);
	for (uint32_t i = s.row0; i < s.row1; ++i) {
		uint8_t* dst = raw.data() + (size_t)(i - s.row0) * line;
		const uint8_t* src = bgr + (size_t)i * width * 3;
		*dst++ = 0; // filter type None
		for (uint32_t j = 0; j < width; ++j) {
			dst[3 * j] = src[3 * j + 2];
			dst[3 * j + 1] = src[3 * j + 1];
			dst[3 * j + 2] = src[3 * j];
		}
	}
	s.rawBytes = raw.size();
	s.adler = adler32(0L, Z_NULL, 0);
	for (size_t done = 0; done < raw.size();) {
		const size_t n = raw.size() - done < (1u << 30) ? raw.size() - done : (1u << 30);
		s.adler = adler32(s.adler, raw.data() + done, (uInt)n);
		done += n;
	}
	z_stream z;
	memset(&z, 0, sizeof z);
	s.err = deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
	if (s.err != Z_OK) return;
	s.out.resize(raw.size() + raw.size() / 1000 + 64 + 5 * (raw.size() / 16000 + 2));
	size_t in = 0, out = 0;
	for (;;) {
		const size_t nin = raw.size() - in < (1u << 30) ? raw.size() - in : (1u << 30);
		const bool lastPiece = in + nin == raw.size();
		z.next_in = raw.data() + in;
		z.avail_in = (uInt)nin;
		int rc;
		do {
			if (out == s.out.size()) s.out.resize(s.out.size() * 2);
			const size_t room = s.out.size() - out < (1u << 30) ? s.out.size() - out : (1u << 30);
			z.next_out = s.out.data() + out;
			z.avail_out = (uInt)room;
			rc = deflate(&z, lastPiece ? (last ? Z_FINISH : Z_SYNC_FLUSH) : Z_NO_FLUSH);
			out += room - z.avail_out;
			if (rc == Z_STREAM_ERROR) { s.err = rc; deflateEnd(&z); return; }
		} while (z.avail_out == 0 || (lastPiece && last && rc != Z_STREAM_END));
		in += nin;
		if (lastPiece) break;
	}
	deflateEnd(&z);
	s.out.resize(out);
	s.err = Z_OK;
}

} // namespace

extern "C" int mirt_write_png(const char* path, const uint8_t* bgr, uint32_t width, uint32_t height, uint32_t threads) {
	if (!path || !bgr) return mirt_set_error(MIRT_E_INVALID, "path or pixel buffer is NULL");
	if (width == 0 || height == 0 || width > 0x7fffffffu / 4 || height > 0x7fffffffu) return mirt_set_error(MIRT_E_INVALID, "image size %ux%u cannot be written as PNG", width, height);
	try {
		unsigned nThreads = threads ? threads : std::thread::hardware_concurrency();
		if (nThreads > 16 && !threads) nThreads = 16;
		if (nThreads < 1) nThreads = 1;
		// at least 64 rows and ~1 MB per stripe, so that independent dictionaries cost little compression
		uint32_t nStripes = nThreads;
		const uint64_t bytes = (uint64_t)width * height * 3;
		if (nStripes > height / 64 + 1) nStripes = height / 64 + 1;
		if (nStripes > bytes / (1u << 20) + 1) nStripes = (uint32_t)(bytes / (1u << 20) + 1);
		std::vector<Stripe> stripes(nStripes);
		for (uint32_t k = 0; k < nStripes; ++k) {
			stripes[k].row0 = (uint32_t)((uint64_t)height * k / nStripes);
			stripes[k].row1 = (uint32_t)((uint64_t)height * (k + 1) / nStripes);
			stripes[k].err = Z_OK;
		}
		std::vector<std::thread> th;
		for (uint32_t k = 1; k < nStripes; ++k) th.emplace_back([&, k]() { deflate_stripe(bgr, width, stripes[k], k + 1 == nStripes, 1); });
		deflate_stripe(bgr, width, stripes[0], nStripes == 1, 1);
		for (auto& t : th) t.join();
		uLong adler = adler32(0L, Z_NULL, 0);
		size_t total = 2 + 4;
		for (uint32_t k = 0; k < nStripes; ++k) {
			if (stripes[k].err != Z_OK) return mirt_set_error(MIRT_E_NOMEM, "zlib deflate failed (%d)", stripes[k].err);
			adler = k == 0 ? stripes[0].adler : adler32_combine(adler, stripes[k].adler, (z_off_t)stripes[k].rawBytes);
			total += stripes[k].out.size();
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
		// one zlib stream over as many IDAT chunks as stripes (+ header and trailer chunks)
		const uint8_t zhead[2] = { 0x78, 0x01 };
		ok = ok && write_chunk(f, "IDAT", zhead, 2);
		for (uint32_t k = 0; ok && k < nStripes; ++k) {
			for (size_t done = 0; ok && done < stripes[k].out.size();) { // a chunk's length field is 31 bits
				const size_t n = stripes[k].out.size() - done < 0x40000000u ? stripes[k].out.size() - done : 0x40000000u;
				ok = write_chunk(f, "IDAT", stripes[k].out.data() + done, n);
				done += n;
			}
		}
		const uint8_t ztail[4] = { (uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler };
		ok = ok && write_chunk(f, "IDAT", ztail, 4) && write_chunk(f, "IEND", nullptr, 0);
		ok = (fclose(f) == 0) && ok;
		if (!ok) return mirt_set_error(MIRT_E_INVALID, "short write to %s", path);
		(void)total;
		return MIRT_OK;
	} catch (const std::bad_alloc&) {
		return mirt_set_error(MIRT_E_NOMEM, "out of host memory while encoding %s", path);
	} catch (const std::exception& e) {
		return mirt_set_error(MIRT_E_INVALID, "PNG encoding failed: %s", e.what());
	}
}
