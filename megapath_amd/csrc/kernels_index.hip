// megapath_amd/csrc/kernels_index.hip -- product code: builds the HBM index layouts on the device from the reference's
// file layouts: rank blocks from the MSB-first .bwt words, LSB-first text from .pac bytes, and the
// 13-mer lookup table from the text when no .lkt file is shipped (LTConstruct.c:44-105).  Streaming kernels: every
// input byte is read once, coalesced; no per-base loop runs on the host.
#include <hip/hip_runtime.h>
#include "mp_internal.h"

// reverse the order of the sixteen 2-bit symbols of a word (MSB-first <-> LSB-first)
__device__ __forceinline__ uint32_t mp_rev2(uint32_t x)
{
    uint32_t r = __brev(x);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

// the sixteen even-numbered bits of a word, packed
__device__ __forceinline__ uint32_t mp_even_bits(uint32_t x)
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

// symbols of block b that exist: the first `valid` of its 192
__device__ __forceinline__ uint32_t mp_block_word(const uint32_t *__restrict__ bwt_msb, uint64_t n, uint32_t b, int w)
{
    const uint64_t first = (uint64_t)b * MP_OCC_SYMS + 16ull * (uint64_t)w;
    if (first >= n) return 0u;
    uint32_t v = mp_rev2(bwt_msb[first >> 4]);
    const uint64_t left = n - first;
    if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
    return v;
}

__global__ void occ_count_kernel(const uint32_t *__restrict__ bwt_msb, uint64_t n, uint32_t n_blocks, uint32_t *__restrict__ cnt /* [4][n_blocks] */)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t c[4] = {0, 0, 0, 0};
    for (int w = 0; w < 12; w++) {
        const uint64_t first = (uint64_t)b * MP_OCC_SYMS + 16ull * (uint64_t)w;
        if (first >= n) break;
        const uint32_t v = mp_block_word(bwt_msb, n, b, w);
        const uint64_t left = n - first;
        const int ns = left < 16 ? (int)left : 16;
        const uint32_t mask = ns < 16 ? ((1u << (2 * ns)) - 1u) & 0x55555555u : 0x55555555u;
        for (uint32_t s = 1; s < 4; s++) {
            const uint32_t x = v ^ (s * 0x55555555u);
            c[s] += (uint32_t)__popc(~(x | (x >> 1)) & mask);
        }
        c[0] += (uint32_t)ns;
    }
    c[0] -= c[1] + c[2] + c[3];
    for (int s = 0; s < 4; s++) cnt[(uint64_t)s * n_blocks + b] = c[s];
}

__global__ void occ_build_kernel(const uint32_t *__restrict__ bwt_msb, uint64_t n, uint32_t n_blocks, const uint32_t *__restrict__ run /* [4][n_blocks] exclusive */,
                                 mp_occ_block *__restrict__ out)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    mp_occ_block blk;
    for (int s = 0; s < 4; s++) blk.cnt[s] = run[(uint64_t)s * n_blocks + b];
    for (int j = 0; j < 6; j++) {
        // two words of sixteen 2-bit symbols -> 32 low bits and 32 high bits (symbols past the end are zero: 'A', never counted)
        const uint32_t wa = mp_block_word(bwt_msb, n, b, 2 * j), wb = mp_block_word(bwt_msb, n, b, 2 * j + 1);
        blk.lo[j] = mp_even_bits(wa) | (mp_even_bits(wb) << 16);
        blk.hi[j] = mp_even_bits(wa >> 1) | (mp_even_bits(wb >> 1) << 16);
    }
    out[b] = blk;
}

__global__ void text_build_kernel(const uint8_t *__restrict__ pac, uint64_t n, uint64_t n_words, uint32_t *__restrict__ text)
{
    const uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (q >= n_words) return;
    const uint64_t first = q * 16;
    uint32_t v = 0;
    if (first < n) {
        const uint8_t *p = pac + q * 4;
        v = mp_rev2(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
        const uint64_t left = n - first;
        if (left < 16) v &= (1u << (2 * (uint32_t)left)) - 1u;
    }
    text[q] = v;
}

// one lane per text position s: the 13-mer starting at s, the text continued with 'A's past its end, leftmost base in
// the most significant bits of the key
__global__ void lkt_hist_kernel(const uint32_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ lkt)
{
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= n) return;
    // 26 bits starting at base s (LSB-first words, zero = 'A' beyond n because the text buffer is zero padded)
    const uint64_t q = s >> 4;
    const uint32_t sh = 2 * (uint32_t)(s & 15);
    const uint64_t two = ((uint64_t)text[q + 1] << 32) | text[q];
    const uint32_t lsb = (uint32_t)(two >> sh) & (MP_LKT_ENTRIES - 1);     // base s in bits 1:0
    const uint32_t key = mp_rev2(lsb) >> 6;                                // base s in bits 25:24
    atomicAdd(&lkt[key], 1u);
}

// A second look-up table over 16-mers (4^16 x 4 B = 17 GB), used by the seeding kernel behind the 17-mer pre-filter: when
// the next 17 scanned bases are known to be reachable, the SA range of the first 16 can be read directly instead of the
// 13-mer table plus three backward-search steps (none of the skipped steps is observable: all are below seedMinLength).
// Same construction as the 13-mer table (LTConstruct.c:44-105): every suffix counts with the text continued by 'A's, then
// an inclusive prefix sum -- so the row numbering is that of the suffix array.  The last three suffixes have fewer than
// four real bases: the reference's procedure (13-mer window starting three bases further) never reaches them, they sit
// in buckets whose 13 rightmost bases are all 'A', and the seeding kernel keeps the 13-mer path for those keys.
__global__ void lkt16_hist_kernel(const uint32_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ lkt16)
{
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint64_t q = s >> 4;
    const uint32_t sh = 2 * (uint32_t)(s & 15);
    const uint64_t two = ((uint64_t)text[q + 1] << 32) | text[q];
    const uint32_t lsb = (uint32_t)(two >> sh);                            // base s in bits 1:0 ('A' beyond n: zero padded)
    atomicAdd(&lkt16[mp_rev2(lsb)], 1u);                                   // base s in bits 31:30
}

__global__ void add_offset_kernel(uint32_t *__restrict__ a, uint64_t n, uint32_t off)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) a[t] += off;
}

// 17-mer presence bitmap (4^17 bits = 2 GiB), the seeding kernel's pre-filter.  A seed start whose first 17 scanned
// bases form a 17-mer that the index cannot extend to is a seed shorter than seedMinLength (17 or more): it is never
// reported and the scan restarts one base further whatever its exact length (CHECK_AND_ADD_RANGE,
// DV-DPfunctions.cpp:2197-2219), so one bit probe replaces the LKT look-up and the ~3 rank-query pairs it would cost.
// "Can be extended to" is what the reference's own procedure reaches: the 13-mer table is built over the text continued
// with 12 'A's (LTConstruct.c:44-105), the four further bases are true backward-search steps; that is the set of
// 17-base windows of (text + 12 'A') starting at 0 .. n-5.  Key: leftmost text base in the top bits, as for the LKT.
// K = 16: the same for the first 16 scanned bases (4^16 bits = 512 MiB; windows starting at 0 .. n-4) -- a weaker filter (a 16-mer is in a
// 2.95 Gbp text with probability 0.50, a 17-mer with 0.16) for index chunks that have no 2 GiB to spare: the lean layout of the NT replica.
__global__ void kmer17_build_kernel(const uint32_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ bitmap, uint32_t K = 17)
{
    const uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (n < K - 12 || p > n - (K - 12)) return;
    const uint64_t q = p >> 4;
    const uint32_t sh = 2 * (uint32_t)(p & 15);
    const uint64_t lo = ((uint64_t)text[q + 1] << 32) | text[q];
    uint64_t v = lo >> sh;                                         // bases p .. in LSB-first order
    if (sh > 30) v |= (uint64_t)text[q + 2] << (64 - sh);          // cannot happen (sh <= 30 leaves 34 bits), kept for clarity
    v &= (1ull << (2 * K)) - 1;
    // reverse the order of the K two-bit groups: base p goes to the top bits
    uint64_t r = __brevll(v);                                      // bit i -> 63 - i
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);   // restore the bit order inside each pair
    const uint64_t key = r >> (64 - 2 * K);                        // the 2 K significant bits
    atomicOr(&bitmap[key >> 5], 1u << (uint32_t)(key & 31));
}

// The same presence information, arranged so that ONE 8-byte load answers the pre-filter for three consecutive scan
// positions (a futile strand is probed at every position).  Entry y (a 15-mer) holds, for the 17-mers that contain y,
//   bits  0..15: y preceded (in scan order) by two bases -- the 17-mer probed at scan position g when y = bases g+2 .. g+16,
//   bits 16..31: y with one base either side                                        -- position g + 1,
//   bits 32..47: y followed by two bases                                            -- position g + 2.
// Keys are in scan order (first scanned base in the low bits; the leftmost text base is scanned last), as in the bitmap:
// with K the 34-bit key of a reachable 17-mer, y and the bit are K >> 4 & K & 15, K >> 2 & (K & 3 | (K >> 32) << 2),
// K & (K >> 30).  4^15 entries, 8.6 GB.
__global__ void kmer15x3_build_kernel(const uint32_t *__restrict__ text, uint64_t n, unsigned long long *__restrict__ tab)
{
    const uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (n < 5 || p > n - 5) return;
    const uint64_t q = p >> 4;
    const uint32_t sh = 2 * (uint32_t)(p & 15);
    const uint64_t lo = ((uint64_t)text[q + 1] << 32) | text[q];
    uint64_t v = lo >> sh;
    if (sh > 30) v |= (uint64_t)text[q + 2] << (64 - sh);
    v &= (1ull << 34) - 1;
    uint64_t r = __brevll(v);
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    const uint64_t K = r >> 30;                                    // as in kmer17_build_kernel: base p in bits 33:32
    const uint64_t m30 = (1ull << 30) - 1;
    atomicOr(&tab[(K >> 4) & m30], 1ull << (K & 15));
    atomicOr(&tab[(K >> 2) & m30], 1ull << (16 + ((K & 3) | (((K >> 32) & 3) << 2))));
    atomicOr(&tab[K & m30], 1ull << (32 + ((K >> 30) & 15)));
}

// Test aid: answers of the three alignments of the 15-mer table for 17-mer keys (must equal the bitmap's bit)
__global__ void kmer15x3_query_kernel(const unsigned long long *__restrict__ tab, const uint64_t *__restrict__ keys, uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t K = keys[t] & ((1ull << 34) - 1), m30 = (1ull << 30) - 1;
    const uint32_t a = (uint32_t)((tab[(K >> 4) & m30] >> (K & 15)) & 1ull);
    const uint32_t b = (uint32_t)((tab[(K >> 2) & m30] >> (16 + ((K & 3) | (((K >> 32) & 3) << 2)))) & 1ull);
    const uint32_t c = (uint32_t)((tab[K & m30] >> (32 + ((K >> 30) & 15))) & 1ull);
    out[t] = (uint8_t)(a | (b << 1) | (c << 2));
}

__global__ void kmer17_query_kernel(const uint32_t *__restrict__ bitmap, const uint64_t *__restrict__ keys, uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) { const uint64_t k = keys[t] & ((1ull << 34) - 1); out[t] = (uint8_t)((bitmap[k >> 5] >> (uint32_t)(k & 31)) & 1u); }
}

// The 16-mer table once more with 16-byte entries {l, r, pos, prev16} (see mp_dev_index::lkt16u).  One lane per suffix-array row i >= 1:
// the row that opens its 16-mer's bucket (i == l) writes the entry.  Needs the whole suffix array (sa_full[i] = SA[i]) and the
// 16-mer table (inclusive right bounds).  Keys are in scan order like the 16-mer table's (leftmost text base in the top bits).
__global__ void lkt16u_init_kernel(uint4 *__restrict__ tab, uint64_t n_entries)
{
    // grid-stride: 2^32 entries are one work-item more than a launch dimension may have
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_entries; k += (uint64_t)gridDim.x * blockDim.x)
        tab[k] = make_uint4(1u, 0u, 0xFFFFFFFFu, 0u);                           // empty range
}

__device__ __forceinline__ uint32_t mp_prev16_scan_order(const uint32_t *__restrict__ text, uint32_t pos)
{
    // text bases pos-1, pos-2, ..., pos-16 in bits 1:0, 3:2, ... (pos >= 16)
    const uint32_t p0 = pos - 16;
    const uint32_t t16 = (uint32_t)((((uint64_t)text[(p0 >> 4) + 1] << 32) | text[p0 >> 4]) >> (2 * (p0 & 15)));
    const uint32_t r = __brev(t16);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}

__global__ void lkt16u_build_kernel(const uint32_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ sa_full, const uint32_t *__restrict__ lkt16, uint4 *__restrict__ tab)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1;     // rows 1 .. n
    if (i > n) return;
    const uint32_t pos = sa_full[i];
    const uint64_t q = pos >> 4;
    const uint32_t sh = 2 * (pos & 15);
    const uint64_t two = ((uint64_t)text[q + 1] << 32) | text[q];
    const uint32_t key = mp_rev2((uint32_t)(two >> sh));                          // as lkt16_hist_kernel: 'A' beyond the text end
    const uint32_t l = key ? lkt16[key - 1] + 1 : 1, r = lkt16[key];
    if ((uint32_t)i != l) return;
    uint4 e = make_uint4(l, r, 0xFFFFFFFFu, 0u);
    if (l == r && pos >= 16 && (uint64_t)pos + 16 <= n) { e.z = pos; e.w = mp_prev16_scan_order(text, pos); }
    tab[key] = e;
}
