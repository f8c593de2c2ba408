// Shared between the translation units of libmirt.so; not part of the public ABI.
#pragma once

// Records the calling thread's error message (what mirt_last_error returns) and returns `code`.
int mirt_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
