// Shared between the translation units of libmirt.so; not part of the public ABI.
#pragma once
#include <stdint.h>

#include <functional>
#include <vector>

// Records the calling thread's error message (what mirt_last_error returns) and returns `code`.
int mirt_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// PNG encoder shared by mirt_write_png and mirt_render_png (mirt_png.cpp).  The image is cut into mirt_png_stripes()
// stripes of whole rows, ceil(height / stripes) rows each; rows_of(stripe, row0, row1, scratch, zero) hands over the PNG
// rows of one stripe (see mirt_png.cpp).
#define MIRT_PNG_MAX_STRIPES 1024u
uint32_t mirt_png_stripes(uint32_t width, uint32_t height);
int mirt_png_encode(const char* path, uint32_t width, uint32_t height, uint32_t threads, uint8_t zeroFilter, int strategy,
                    const std::function<const uint8_t*(uint32_t, uint32_t, uint32_t, std::vector<uint8_t>&, bool*)>& rows_of);
