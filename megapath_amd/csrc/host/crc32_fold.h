// megapath_amd/csrc/host/crc32_fold.h -- product code: CRC-32 (the gzip polynomial) by carry-less multiplication.
//
// Every gzip member's CRC-32 is checked over the inflated bytes (as zlib does); zlib's table-driven crc32() runs at about 1 GB/s, which
// is as much time as the parallel reader needs to turn a segment's symbols into bytes.  This is the folding scheme of Gopal, Ozturk et al.,
// "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction" (Intel, 2009), with the paper's constants for the bit-reflected
// polynomial 0x1DB710641: four 128-bit lanes are folded forward by 512 bits per step (k1, k2), then into one lane (k3, k4), reduced to
// 64 bits (k5) and to 32 by Barrett reduction (mu, P).  fast_crc32() has crc32()'s interface; the first call checks the routine against
// zlib on a buffer of its own and keeps to zlib if the CPU lacks the instruction or the results differ.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mpz {

#if defined(__x86_64__)
// state in, state out (both without the final inversion); len: a multiple of 16, at least 64
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold_pclmul(const unsigned char *buf, size_t len, uint32_t crc)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0)), x2 = _mm_loadu_si128((const __m128i *)(buf + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 32)), x4 = _mm_loadu_si128((const __m128i *)(buf + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    buf += 64; len -= 64;
    while (len >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(buf + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(buf + 48)));
        buf += 64; len -= 64;
    }
    // four lanes into one
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x2), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x3), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x4), t);
    while (len >= 16) {
        t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), _mm_loadu_si128((const __m128i *)buf)), t);
        buf += 16; len -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i y = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), y);
    y = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), y);
    // Barrett reduction to 32 bits
    y = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(y, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

inline bool crc32_fold_usable()
{
#if defined(__x86_64__)
    static const bool ok = []() {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        unsigned char b[64 + 16 * 7 + 64 * 5];
        uint32_t x = 12345u;
        for (size_t i = 0; i < sizeof b; i++) { x = x * 1664525u + 1013904223u; b[i] = (unsigned char)(x >> 24); }
        for (size_t n : {(size_t)64, (size_t)80, (size_t)128, sizeof b}) {
            const uint32_t seed = (uint32_t)crc32(0L, b, 7);
            if ((uint32_t)crc32(seed, b, (uInt)n) != ~crc32_fold_pclmul(b, n, ~seed)) return false;
        }
        return true;
    }();
    return ok;
#else
    return false;
#endif
}

// crc32() of zlib, faster on long buffers
inline uint32_t fast_crc32(uint32_t crc, const unsigned char *buf, size_t len)
{
#if defined(__x86_64__)
    if (len >= 256 && crc32_fold_usable()) {
        const size_t n = len & ~(size_t)15;
        crc = ~crc32_fold_pclmul(buf, n, ~crc);
        buf += n; len -= n;
    }
#endif
    while (len) { const size_t n = len < ((size_t)1 << 30) ? len : ((size_t)1 << 30); crc = (uint32_t)crc32(crc, buf, (uInt)n); buf += n; len -= n; }
    return crc;
}

}  // namespace mpz
