/*
 * s3s_oracle_simd.c — host-speed checksums for the CPU BASELINE leg of bench.py.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see s3s_oracle.h): nothing under spark-s3-shuffle_amd/ links or loads this.
 *
 * Why it exists (SURVEY §8(d) "CRC32 must use a hardware-speed implementation"): the reference
 * path's checksums are java.util.zip.CRC32 / Adler32 (S3ShuffleHelper.scala:94-103,
 * S3ChecksumValidationStream.scala:58), which HotSpot runs as intrinsics — carry-less-multiply folding for CRC32,
 * a vectorised sum for Adler32.  A table-driven restatement (s3s_oracle.c: slice-by-8, byte loop) would make the host
 * look slower than the JVM really is and inflate the GPU's speed-up.  These two functions are the published
 * algorithms behind those intrinsics:
 *   CRC32   — folding by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using
 *             PCLMULQDQ Instruction", Intel 2009): four 128-bit lanes folded by x^512 mod P, reduced to one by
 *             x^128 mod P, then 128 -> 64 -> 32 bits with a Barrett reduction.  The constants are the residues
 *             x^(n) mod P of the reflected IEEE 802.3 polynomial; they are computed at start-up from P by
 *             xpow_mod() below and compared with the values the paper lists, not copied from a library.
 *   Adler32 — 32 bytes per step: byte sums by PSADBW, position-weighted sums by PMADDUBSW against the taps 32..1,
 *             the running s1 added 32 times per step through a "previous sums" accumulator; modulo every 5 552 bytes.
 * Both fall back to the scalar restatement for short inputs and when the CPU lacks the instructions, and are pinned
 * against zlib and the restatement by tests/test_oracle_pins.py (random lengths and alignments).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "s3s_oracle.h"

#if defined(__x86_64__)
#include <immintrin.h>

/* ---- CRC32 ---------------------------------------------------------------------------------------------------- */
/* x^n mod P in the reflected domain: bit 31 of the result is the x^0 coefficient.  Multiply-by-x = shift right and
 * xor 0xEDB88320 when a 1 falls out. */
static uint32_t xpow_mod(unsigned n) {
  uint32_t r = 0x80000000u; /* x^0 */
  for (unsigned i = 0; i < n; i++) r = (r >> 1) ^ ((r & 1u) ? 0xEDB88320u : 0u);
  return r;
}

/* floor(x^64 / P) reflected, 33 bits: the Barrett constant u */
static uint64_t barrett_u(void) {
  /* polynomial long division of x^64 by P (degree 32, P = 0x104C11DB7 unreflected) */
  const uint64_t P = 0x104C11DB7ull;
  uint64_t rem = 1ull << 32, q = 0; /* start with x^32 as the running remainder of x^32; extend 32 more steps */
  for (int i = 0; i < 32; i++) {
    q <<= 1;
    if (rem & (1ull << 32)) {
      q |= 1;
      rem ^= P;
    }
    rem <<= 1;
  }
  q <<= 1;
  if (rem & (1ull << 32)) q |= 1;
  /* q has 33 bits (x^32 .. x^0), unreflected; reflect into 33 bits */
  uint64_t r = 0;
  for (int i = 0; i < 33; i++)
    if (q & (1ull << i)) r |= 1ull << (32 - i);
  return r;
}

static uint64_t K512p64, K512, K128p64, K128, K64, KU, KP; /* set once */
static int crc_consts_ready = 0;

static void crc_consts(void) {
  /* the folding constants are (x^(n-32) mod P) << 1 in the reflected 64-bit lane convention of PCLMULQDQ */
  K512p64 = (uint64_t)xpow_mod(4 * 128 + 64 - 32) << 1; /* low lane of the 512-bit fold  */
  K512 = (uint64_t)xpow_mod(4 * 128 - 32) << 1;         /* high lane                     */
  K128p64 = (uint64_t)xpow_mod(128 + 64 - 32) << 1;
  K128 = (uint64_t)xpow_mod(128 - 32) << 1;
  K64 = (uint64_t)xpow_mod(64) << 1; /* 96 -> 64: the 32 bits above the low lane fold by x^64 mod P */
  KU = barrett_u();
  KP = ((uint64_t)0xEDB88320u << 1) | 1ull; /* P reflected, 33 bits */
  crc_consts_ready = 1;
}

int s3o_simd_crc_constants(uint64_t out[7]) { /* for the pin test: the paper's table */
  if (!crc_consts_ready) crc_consts();
  out[0] = K512p64; out[1] = K512; out[2] = K128p64; out[3] = K128; out[4] = K64; out[5] = KP; out[6] = KU;
  return 7;
}

__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(uint32_t crc, const uint8_t* p, size_t len) {
  /* len >= 64 and a multiple of 16; crc is the running (already inverted) register */
  __m128i x1 = _mm_loadu_si128((const __m128i*)(p + 0)), x2 = _mm_loadu_si128((const __m128i*)(p + 16));
  __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  __m128i k = _mm_set_epi64x((long long)K512, (long long)K512p64);
  p += 64;
  len -= 64;
  while (len >= 64) {
    __m128i l1 = _mm_clmulepi64_si128(x1, k, 0x00), l2 = _mm_clmulepi64_si128(x2, k, 0x00);
    __m128i l3 = _mm_clmulepi64_si128(x3, k, 0x00), l4 = _mm_clmulepi64_si128(x4, k, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k, 0x11);
    x2 = _mm_clmulepi64_si128(x2, k, 0x11);
    x3 = _mm_clmulepi64_si128(x3, k, 0x11);
    x4 = _mm_clmulepi64_si128(x4, k, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, l1), _mm_loadu_si128((const __m128i*)(p + 0)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, l2), _mm_loadu_si128((const __m128i*)(p + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, l3), _mm_loadu_si128((const __m128i*)(p + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, l4), _mm_loadu_si128((const __m128i*)(p + 48)));
    p += 64;
    len -= 64;
  }
  k = _mm_set_epi64x((long long)K128, (long long)K128p64);
  __m128i l;
  l = _mm_clmulepi64_si128(x1, k, 0x00); x1 = _mm_clmulepi64_si128(x1, k, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, l), x2);
  l = _mm_clmulepi64_si128(x1, k, 0x00); x1 = _mm_clmulepi64_si128(x1, k, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, l), x3);
  l = _mm_clmulepi64_si128(x1, k, 0x00); x1 = _mm_clmulepi64_si128(x1, k, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, l), x4);
  while (len >= 16) {
    l = _mm_clmulepi64_si128(x1, k, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, l), _mm_loadu_si128((const __m128i*)p));
    p += 16;
    len -= 16;
  }
  /* 128 -> 64: the low lane times x^128-ish folds onto the high lane */
  const __m128i mask32 = _mm_setr_epi32(-1, 0, -1, 0);
  __m128i t = _mm_clmulepi64_si128(x1, k, 0x10); /* low 64 of x1 times K128 */
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
  /* 64 -> 32 */
  __m128i k64 = _mm_set_epi64x(0, (long long)K64);
  t = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, k64, 0x00);
  x1 = _mm_xor_si128(x1, t);
  /* Barrett: q = floor(low32(x1) * u / x^32); crc = (x1 xor q * P) >> 32 */
  __m128i pu = _mm_set_epi64x((long long)KU, (long long)KP);
  t = _mm_and_si128(x1, mask32);
  t = _mm_clmulepi64_si128(t, pu, 0x10);
  t = _mm_and_si128(t, mask32);
  t = _mm_clmulepi64_si128(t, pu, 0x00);
  x1 = _mm_xor_si128(x1, t);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

/* ---- Adler32 -------------------------------------------------------------------------------------------------- */
__attribute__((target("ssse3"))) static void adler32_blocks(uint32_t* pa, uint32_t* pb, const uint8_t* p, size_t blocks) {
  /* `blocks` 32-byte steps */
  const __m128i tap1 = _mm_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17);
  const __m128i tap2 = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
  const __m128i zero = _mm_setzero_si128(), ones = _mm_set1_epi16(1);
  uint32_t a = *pa, b = *pb;
  while (blocks) {
    size_t n = blocks < 5552 / 32 ? blocks : 5552 / 32;
    blocks -= n;
    __m128i v_ps = _mm_set_epi32(0, 0, 0, (int)(a * (uint32_t)n)); /* a enters b once per byte: 32 n times in total (<< 5 below) */
    __m128i v_s2 = _mm_set_epi32(0, 0, 0, (int)b), v_s1 = zero;
    do {
      const __m128i b1 = _mm_loadu_si128((const __m128i*)p), b2 = _mm_loadu_si128((const __m128i*)(p + 16));
      v_ps = _mm_add_epi32(v_ps, v_s1);
      v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(b1, zero));
      v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(b1, tap1), ones));
      v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(b2, zero));
      v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(b2, tap2), ones));
      p += 32;
    } while (--n);
    v_s2 = _mm_add_epi32(v_s2, _mm_slli_epi32(v_ps, 5));
    v_s1 = _mm_add_epi32(v_s1, _mm_shuffle_epi32(v_s1, 0xB1));
    v_s1 = _mm_add_epi32(v_s1, _mm_shuffle_epi32(v_s1, 0x4E));
    a += (uint32_t)_mm_cvtsi128_si32(v_s1);
    v_s2 = _mm_add_epi32(v_s2, _mm_shuffle_epi32(v_s2, 0xB1));
    v_s2 = _mm_add_epi32(v_s2, _mm_shuffle_epi32(v_s2, 0x4E));
    b = (uint32_t)_mm_cvtsi128_si32(v_s2);
    a %= 65521u;
    b %= 65521u;
  }
  *pa = a;
  *pb = b;
}

static int cpu_has(const char* what) {
  __builtin_cpu_init();
  if (!strcmp(what, "pclmul")) return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (!strcmp(what, "sse4.2")) return __builtin_cpu_supports("sse4.2");
  return __builtin_cpu_supports("ssse3");
}
#else
static int cpu_has(const char* what) { (void)what; return 0; }
#endif

int s3o_simd_available(void) { return (cpu_has("pclmul") ? 1 : 0) | (cpu_has("ssse3") ? 2 : 0); }

uint32_t s3o_crc32_fast(uint32_t crc, const void* data, size_t len) {
#if defined(__x86_64__)
  static int ok = -1;
  if (ok < 0) {
    ok = cpu_has("pclmul");
    if (ok && !crc_consts_ready) crc_consts();
  }
  if (ok && len >= 64) {
    const uint8_t* p = (const uint8_t*)data;
    const size_t body = len & ~(size_t)15;
    uint32_t c = ~crc32_fold(~crc, p, body);
    return s3o_crc32(c, p + body, len - body);
  }
#endif
  return s3o_crc32(crc, data, len);
}

uint32_t s3o_adler32_fast(uint32_t adler, const void* data, size_t len) {
#if defined(__x86_64__)
  static int ok = -1;
  if (ok < 0) ok = cpu_has("ssse3");
  if (ok && len >= 64) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    const size_t blocks = len / 32;
    adler32_blocks(&a, &b, p, blocks);
    return s3o_adler32((b << 16) | a, p + blocks * 32, len - blocks * 32);
  }
#endif
  return s3o_adler32(adler, data, len);
}

/* ---- CRC32C: the processor's own instruction (SSE4.2 crc32 r64, r/m64 computes exactly this polynomial) --------------- */
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static uint32_t crc32c_insn(uint32_t c, const uint8_t* p, size_t len) {
  uint64_t c64 = c;
  while (len >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
    p += 8;
    len -= 8;
  }
  c = (uint32_t)c64;
  while (len--) c = __builtin_ia32_crc32qi(c, *p++);
  return c;
}
#endif
int s3o_crc32c_hw_available(void) {
#if defined(__x86_64__)
  static int ok = -1;
  if (ok < 0) ok = cpu_has("sse4.2");
  return ok;
#else
  return 0;
#endif
}
uint32_t s3o_crc32c_hw(uint32_t crc, const void* data, size_t len) {
#if defined(__x86_64__)
  if (s3o_crc32c_hw_available()) return ~crc32c_insn(~crc, (const uint8_t*)data, len);
#endif
  return s3o_crc32c(crc, data, len);
}

int64_t s3o_checksum_fast(int algo, const void* data, size_t len) {
  switch (algo) {
    case S3O_CHECKSUM_ADLER32:
      return (int64_t)s3o_adler32_fast(1u, data, len);
    case S3O_CHECKSUM_CRC32:
      return (int64_t)s3o_crc32_fast(0u, data, len);
    case S3O_CHECKSUM_CRC32C:
      return (int64_t)s3o_crc32c_hw(0u, data, len);
    default:
      return 0;
  }
}
