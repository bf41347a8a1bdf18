// placeholder until the reduce-side kernels land
#include "s3s_internal.h"
extern "C" {
int s3s_decompress_range(s3s_ctx*, int, int, const uint8_t*, int64_t, const int64_t*, const int64_t*, int32_t, uint8_t*, int64_t, int64_t*, int32_t*) { return S3S_E_UNSUPPORTED; }
int s3s_decompress_range_device(s3s_ctx*, int, int, const uint8_t*, int64_t, const int64_t*, const int64_t*, int32_t, uint8_t*, int64_t, int64_t*, int32_t*) { return S3S_E_UNSUPPORTED; }
int s3s_decompressed_size(s3s_ctx*, int, const uint8_t*, int64_t, int64_t*) { return S3S_E_UNSUPPORTED; }
}
