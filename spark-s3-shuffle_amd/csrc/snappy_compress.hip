// snappy_compress.hip — placeholder: the map-side Snappy kernel has not landed yet; the C-ABI
// reports S3S_E_UNSUPPORTED for S3S_CODEC_SNAPPY instead of producing anything.
#include "s3s_internal.h"
namespace s3s {
bool snappy_compress_available() { return false; }
void launch_snappy_compress(const uint8_t*, const Item*, int32_t, uint8_t*, uint32_t*, hipStream_t) {}
}  // namespace s3s
