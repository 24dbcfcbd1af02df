// gmr_internal.h -- shared by the translation units of libgmrhip.so, not part of the C-ABI.
#ifndef GMR_INTERNAL_H
#define GMR_INTERNAL_H
// records the thread-local message returned by gmr_last_error() and returns `code`
__attribute__((visibility("hidden"))) int gmr_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#endif
