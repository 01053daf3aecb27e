// megapath_amd/csrc/kernels_seed.hip -- product code: MMP seeding on gfx950.
//
// K1  seed_kernel     one lane per (read, strand): LKT-seeded backward search with the reference's restart / reseed
//                     rules (mmp<0>, mmp<2>, CHECK_AND_SET_LAST, CHECK_AND_ADD_RANGE: DV-DPfunctions.cpp:2188-2377).
// K2  locate_kernel   one lane per SA hit: LF-walk to a sampled row (BWTSaValue, BWT.c:968-998) and project the hit to
//                     the read start (DV-DPfunctions.cpp:2481-2495).
// K3  merge_kernel    one lane per read: sort hits, fuzzy-merge, covered length, uniqueness and short-seed filters
//                     (DV-DPfunctions.cpp:2499-2553).
// K4  pair_kernel     one lane per pair: insert-window join of '+' seeds with the mate's '-' seeds
//                     (pairEndMerge, DV-DPfunctions.cpp:1968-2070).
// All of it is dependent integer gathers over HBM-resident tables: no LDS tiling applies, the levers are one 64-byte
// sector per rank query, every lane's dependent load of a step in flight at once (one wait per trip of the persistent
// loop), few instructions per step (bit-plane rank blocks), and no host round trips.
#include <hip/hip_runtime.h>
#include "mp_internal.h"

#define MP_WAVE 64

// The SA range of a 13-mer (LT.h:49-55): [lkt[k - 1] + 1, lkt[k]], from the plain table or from the 8-byte entries that stand for it (mp_dev_index::lkt14)
__device__ __forceinline__ void mp_lkt13_range(const mp_dev_index &ix, uint32_t k, uint32_t &l, uint32_t &r)
{
    if (ix.lkt) { l = k ? ix.lkt[k - 1] + 1 : 1; r = ix.lkt[k]; }
    else { l = ix.lkt14[k].x + 1; r = ix.lkt14[k + 1].x; }
}

// ------------------------------------------------------------------------------------------------ FM-index primitives
// A rank block in registers: running counts, and per 32 symbols one dense mask of the positions that hold symbol c.
struct mp_blk_view {
    uint32_t m[6];      // match masks of the symbol
    uint32_t S[6];      // S[j] = matches in words 0 .. j-1
    uint32_t cnt;       // occurrences before the block
};

__device__ __forceinline__ void mp_blk_match(const uint4 &q0, const uint4 &q1, const uint4 &q2, const uint4 &q3, uint32_t c, mp_blk_view &v)
{
    const uint32_t lo[6] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y}, hi[6] = {q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const uint32_t cl = (c & 1u) ? 0u : ~0u, ch = (c & 2u) ? 0u : ~0u;
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { v.m[j] = (lo[j] ^ cl) & (hi[j] ^ ch); v.S[j] = run; run += (uint32_t)__popc(v.m[j]); }
    v.cnt = c == 0 ? q0.x : c == 1 ? q0.y : c == 2 ? q0.z : q0.w;
}

// occurrences of the symbol among the first `off` (< 192) symbols of the block, plus those before the block
__device__ __forceinline__ uint32_t mp_blk_rank(const mp_blk_view &v, uint32_t off)
{
    const uint32_t wi = off >> 5, pm = (1u << (off & 31u)) - 1u;
    const bool b0 = wi & 1u, b1 = wi & 2u, b2 = wi & 4u;
    const uint32_t m01 = b0 ? v.m[1] : v.m[0], m23 = b0 ? v.m[3] : v.m[2], m45 = b0 ? v.m[5] : v.m[4];
    const uint32_t s01 = b0 ? v.S[1] : v.S[0], s23 = b0 ? v.S[3] : v.S[2], s45 = b0 ? v.S[5] : v.S[4];
    const uint32_t mw = b2 ? m45 : (b1 ? m23 : m01), sw = b2 ? s45 : (b1 ? s23 : s01);
    return v.cnt + sw + (uint32_t)__popc(mw & pm);
}

// the symbol at position so (< 192) of a block
__device__ __forceinline__ uint32_t mp_blk_symbol(const uint4 &q1, const uint4 &q2, const uint4 &q3, uint32_t so)
{
    const uint32_t lo[6] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y}, hi[6] = {q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const uint32_t wi = so >> 5;
    const bool b0 = wi & 1u, b1 = wi & 2u, b2 = wi & 4u;
    const uint32_t l01 = b0 ? lo[1] : lo[0], l23 = b0 ? lo[3] : lo[2], l45 = b0 ? lo[5] : lo[4];
    const uint32_t h01 = b0 ? hi[1] : hi[0], h23 = b0 ? hi[3] : hi[2], h45 = b0 ? hi[5] : hi[4];
    const uint32_t lw = b2 ? l45 : (b1 ? l23 : l01), hw = b2 ? h45 : (b1 ? h23 : h01);
    return ((lw >> (so & 31u)) & 1u) | (((hw >> (so & 31u)) & 1u) << 1);
}

// rank_c over BWT[0,k) with the '$' row removed from the numbering (BWTOccValue, BWT.c:597-634)
__device__ __forceinline__ uint32_t mp_occ_raw(const mp_dev_index &ix, uint32_t k, uint32_t c)
{
    uint32_t b = k / MP_OCC_SYMS, off = k - b * MP_OCC_SYMS;
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + b);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    mp_blk_view v;
    mp_blk_match(q0, q1, q2, q3, c, v);
    return mp_blk_rank(v, off);
}

__device__ __forceinline__ uint32_t mp_occ(const mp_dev_index &ix, uint32_t k, uint32_t c)
{
    k -= (k > ix.inv_sa0);
    return mp_occ_raw(ix, k, c);
}

// LF step (BWTPsiMinusValue, BWT.c:915-938): ONE 64-byte block load gives both the symbol BWT[k-1] and its rank at k
// (k-1 and k share the block unless k is a block boundary, where the rank is the next block's running count: the
// symbol's own occurrence is then added by hand).
__device__ __forceinline__ uint32_t mp_lf(const mp_dev_index &ix, uint32_t idx)
{
    if (idx == ix.inv_sa0) return 0;
    uint32_t k = idx + 1;
    k -= (k > ix.inv_sa0);
    const uint32_t s = k - 1;
    const uint32_t b = s / MP_OCC_SYMS, so = s - b * MP_OCC_SYMS;       // block of the symbol; rank wanted at offset so+1
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + b);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const uint32_t c = mp_blk_symbol(q1, q2, q3, so);
    mp_blk_view v;
    mp_blk_match(q0, q1, q2, q3, c, v);
    // symbols of this block counted: [0, so] -- so + 1 may be 192: the rank up to so, plus the symbol itself
    // (the cumulative count by selects: ix.cum[c] with a per-lane c would be a second, dependent load from the argument block in every step)
    const uint32_t cumc = c == 0 ? ix.cum[0] : c == 1 ? ix.cum[1] : c == 2 ? ix.cum[2] : ix.cum[3];
    return cumc + mp_blk_rank(v, so) + 1u;
}

// The same step with the block already in registers (locate_kernel keeps the loads of two walks in flight before it consumes either)
__device__ __forceinline__ uint32_t mp_lf_blk(const mp_dev_index &ix, uint32_t s, const uint4 &q0, const uint4 &q1, const uint4 &q2, const uint4 &q3)
{
    const uint32_t b = s / MP_OCC_SYMS, so = s - b * MP_OCC_SYMS;
    const uint32_t c = mp_blk_symbol(q1, q2, q3, so);
    mp_blk_view v;
    mp_blk_match(q0, q1, q2, q3, c, v);
    const uint32_t cumc = c == 0 ? ix.cum[0] : c == 1 ? ix.cum[1] : c == 2 ? ix.cum[2] : ix.cum[3];
    return cumc + mp_blk_rank(v, so) + 1u;
}

__global__ void occ_query_kernel(mp_dev_index ix, const uint64_t *idx, const uint32_t *sym, uint64_t n, uint64_t *out)
{
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) out[t] = mp_occ(ix, (uint32_t)idx[t], sym[t]);
}

// Denser SA samples than the file's (optional table sa_dense, derived at load time from the file's samples): since
// SA[LF(i)] = SA[i] - 1 the walk may stop at any sampled row.  With HBM to spare the table is the whole suffix array
// (shift 0: no LF step at all), else every 2nd or 4th row.
__device__ __forceinline__ uint64_t mp_sa_value(const mp_dev_index &ix, uint32_t idx, uint32_t *steps_out)
{
    uint32_t steps = 0, q;
    if (ix.sa_dense) {
        const uint32_t mask = (1u << ix.sa_dense_shift) - 1u;
        while (idx & mask) { idx = mp_lf(ix, idx); steps++; }
        *steps_out = steps;
        q = idx >> ix.sa_dense_shift;
        return (idx == 0 ? ~0ull : (uint64_t)ix.sa_dense[q]) + steps;
    }
    if (ix.sa_shift != 0xFFFFFFFFu) {             // power-of-two sampling: mask and shift instead of integer division
        const uint32_t mask = ix.sa_interval - 1;
        while (idx & mask) { idx = mp_lf(ix, idx); steps++; }
        q = idx >> ix.sa_shift;
    } else {
        while (idx % ix.sa_interval != 0) { idx = mp_lf(ix, idx); steps++; }
        q = idx / ix.sa_interval;
    }
    *steps_out = steps;
    // SA[0] is kept as -1 so that walking through the '$' row yields text position `steps - 1` (BWT.c:242,994-996)
    uint64_t base = q == 0 ? ~0ull : (uint64_t)ix.sa[q];
    return base + steps;
}

// fills sa_dense[k] = SA[k << shift] by the ordinary walk to the file's samples (ix.sa_dense must be null here)
__global__ void sa_dense_build_kernel(mp_dev_index ix, uint32_t n_rows, uint32_t shift, uint32_t *__restrict__ out)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n_rows) return;
    uint32_t steps;
    out[k] = (uint32_t)mp_sa_value(ix, (uint32_t)(k << shift), &steps);
}

__global__ void sa_query_kernel(mp_dev_index ix, const uint64_t *idx, uint64_t n, uint64_t *out)
{
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t steps;
    if (t < n) out[t] = mp_sa_value(ix, (uint32_t)idx[t], &steps);
}

__global__ void lkt_query_kernel(mp_dev_index ix, const uint32_t *key, uint64_t n, uint64_t *out_l, uint64_t *out_r)
{
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) {
        uint32_t k = key[t];
        uint32_t l, r;
        mp_lkt13_range(ix, k, l, r);
        out_l[t] = l; out_r[t] = r;
    }
}

// ------------------------------------------------------------------------------------------------ calibration
// Random 64-byte gathers over the rank blocks with a known byte count: the ceiling the seed stage is priced against
// ("measured random-64 B-gather ceiling") and the calibration of the FETCH_SIZE counter for this access shape.
// Each lane issues `per_lane` independent block loads (4 x dwordx4 each, like a rank query) at hashed block indices.
__global__ void __launch_bounds__(256)
gather64_kernel(mp_dev_index ix, uint32_t n_blocks, uint32_t per_lane, uint32_t seed, uint32_t *sink)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_lane; i++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t b = (uint32_t)(((uint64_t)x * n_blocks) >> 32);
        const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + b);
        uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        acc += q0.x ^ q1.y ^ q2.z ^ q3.w;
    }
    if (acc == 0x12345u) sink[0] = acc;      // keeps the loads alive
}

// ------------------------------------------------------------------------------------------------ scan-order reads
// scan[(read*2+strand)*S + w]: the read's bases in the order the strand's backward search consumes them
// (strand 0: right-to-left; strand 1: left-to-right, complemented), LSB-first, zero padded.  A row has S = W + 2 words rounded
// up to a multiple of four: 16-byte aligned rows (seed_kernel fetches them as 64-byte blocks), and a 34-bit key can be cut
// from two adjacent words at any scan position up to the read's end.
__host__ __device__ __forceinline__ uint32_t mp_scan_stride(uint32_t W) { return (W + 2u + 3u) & ~3u; }

__global__ void prep_scan_kernel(const uint32_t *reads, const uint32_t *lens, uint32_t n_reads, uint32_t W, uint32_t *scan)
{
    const uint32_t W2 = mp_scan_stride(W);
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)n_reads * 2 * W2;
    if (t >= total) return;
    const int w = (int)(t % W2);
    const uint32_t rs = (uint32_t)(t / W2);
    const uint32_t r = rs >> 1, strand = rs & 1;
    const int len = (int)lens[r], nw = (len + 15) / 16;
    const uint32_t *src = reads + (uint64_t)r * W;
    uint32_t out = 0;
    if (w < nw) {
        if (strand) out = ~src[w];                                 // left to right, complemented
        else {
            // right to left: the 2-bit groups of every word reversed, the words in reverse order, and the 16 nw - len pad bases that the
            // reversal brought to the front shifted out
            auto z = [&](int q) -> uint32_t {
                if (q >= nw) return 0u;
                const uint32_t x = __brev(src[nw - 1 - q]);
                return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
            };
            out = (uint32_t)((((uint64_t)z(w + 1) << 32) | z(w)) >> (2u * (uint32_t)(16 * nw - len)));
        }
        if (w == nw - 1 && (len & 15)) out &= (1u << (2 * (len & 15))) - 1u;     // zero padding past the read's end
    }
    scan[t] = out;
}

// ------------------------------------------------------------------------------------------------ K1
// Persistent lanes: every lane pulls (read, strand) items from one queue until it is empty, so a wavefront never idles
// behind its slowest read.  seed_kernel (below) issues one load per lane and trip and waits once, whatever mix of states the
// 64 lanes are in; mp_mmp_trip is the same state machine written as branches that load and use (kept for
// seed_overflow_kernel: one item per lane, run to completion).
// Seeds go to a fixed slab of MP_SEED_SLAB records per (read, strand) -- no atomics on the hot path; the rare item with
// more seeds is listed in `overflow_rs` and redone by seed_overflow_kernel with a full-size slab.
#define MP_SEED_SLAB 64
#define MP_MIN_ALIGNABLE_LEN 30   // = mp_dp_cutoff(len) for every len <= 150: shorter reads cannot reach their cutoff
#define MP_SEED_CHUNK 64

struct mp_seed_rec { uint32_t sa_l; uint16_t query_offset, seed_len; uint32_t sa_diff; };   // 12 bytes

// rank_c at k1 and k2 with a single block load when both fall into the same 64-byte block (narrow SA ranges do)
__device__ __forceinline__ void mp_occ_pair(const mp_dev_index &ix, uint32_t k1, uint32_t k2, uint32_t c, uint32_t &o1, uint32_t &o2)
{
    k1 -= (k1 > ix.inv_sa0);
    k2 -= (k2 > ix.inv_sa0);
    const uint32_t b1 = k1 / MP_OCC_SYMS, off1 = k1 - b1 * MP_OCC_SYMS;
    const uint32_t b2 = k2 / MP_OCC_SYMS, off2 = k2 - b2 * MP_OCC_SYMS;
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + b1);
    uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    mp_blk_view v;
    mp_blk_match(q0, q1, q2, q3, c, v);
    o1 = mp_blk_rank(v, off1);
    if (b2 != b1) {
        const uint4 *pp = reinterpret_cast<const uint4 *>(ix.occ + b2);
        q0 = pp[0]; q1 = pp[1]; q2 = pp[2]; q3 = pp[3];
        mp_blk_match(q0, q1, q2, q3, c, v);
    }
    o2 = mp_blk_rank(v, off2);
}

// The 13-mer table as 8-byte entries that carry the first extension step (mp_dev_index::lkt14; ix.lkt still resident here, ix.lkt14 null).  Entry Q = the 13-mer
// c.P' (first base c, then twelve): the four 13-mers P'.d lie side by side in the table, so the step "put c in front" maps their ranges onto four ranges
// side by side again -- the 14-mers Q.d -- and what the entry has to hold is where the first begins and how long each is.  Where that first range does not
// begin right behind lkt[Q - 1] (a suffix shorter than 14 bases at the end of the text), or a count does not fit a byte, the entry says 255 four times.
__global__ void lkt14_build_kernel(mp_dev_index ix, uint2 *__restrict__ out, unsigned long long *__restrict__ n_plain /* entries without the short cut */)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t > MP_LKT_ENTRIES) return;
    const uint32_t Q = (uint32_t)t;
    if (Q == MP_LKT_ENTRIES) { out[Q] = make_uint2(ix.lkt[MP_LKT_ENTRIES - 1], 0xFFFFFFFFu); return; }
    const uint32_t x = Q ? ix.lkt[Q - 1] : 0u;
    const uint32_t c = Q >> (2 * (MP_LKT_K - 1)), rest = Q & ((1u << (2 * (MP_LKT_K - 1))) - 1u);
    uint32_t packed = 0, expect = x + 1;
    bool plain = false;
#pragma unroll
    for (uint32_t d = 0; d < 4; d++) {
        uint32_t l, r, o1, o2;
        mp_lkt13_range(ix, rest * 4u + d, l, r);
        mp_occ_pair(ix, l, r + 1, c, o1, o2);                      // (an empty range [l, l - 1] yields an empty one)
        const uint32_t nl = ix.cum[c] + o1 + 1, nr = ix.cum[c] + o2;
        const uint32_t cnt = nr + 1 - nl;
        if (nl != expect || nr + 1 < nl || cnt >= 255u) plain = true;
        packed |= (cnt & 255u) << (8 * d);
        expect = nr + 1;
    }
    if (plain) { packed = 0xFFFFFFFFu; atomicAdd(n_plain, 1ull); }
    out[Q] = make_uint2(x, packed);
}

// Test aid: for the 16-mer starting at every text position s <= n - 4 (and for `n_random` arbitrary keys) compares the
// 16-mer table's range with what the index's own procedure yields: 13-mer table, then three backward-search steps.
__global__ void lkt16_check_kernel(mp_dev_index ix, uint64_t n_text, uint64_t n_random, unsigned long long *__restrict__ out /* [checked, mismatches] */)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n_text + n_random) return;
    uint32_t key16;                                            // scan order: rightmost base in bits 1:0
    if (t < n_text) {
        const uint64_t q = t >> 4;
        const uint32_t sh = 2 * (uint32_t)(t & 15);
        const uint64_t two = ((uint64_t)ix.text[q + 1] << 32) | ix.text[q];
        const uint32_t lsb = (uint32_t)(two >> sh);            // text order, base t in bits 1:0
        const uint32_t r = __brev(lsb);
        key16 = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    } else {
        uint64_t x = (t - n_text + 1) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        key16 = (uint32_t)x;
    }
    if ((key16 & (MP_LKT_ENTRIES - 1)) == 0) return;           // keys the seeding kernel leaves to the 13-mer path
    const uint32_t key13 = key16 & (MP_LKT_ENTRIES - 1);
    uint32_t l, r;
    mp_lkt13_range(ix, key13, l, r);
    for (int step = 0; step < 3 && l <= r; step++) {
        const uint32_t c = (key16 >> (26 + 2 * step)) & 3u;
        uint32_t o1, o2;
        mp_occ_pair(ix, l, r + 1, c, o1, o2);
        l = ix.cum[c] + o1 + 1; r = ix.cum[c] + o2;
    }
    const uint32_t tl = ix.lkt16[key16 - 1] + 1, tr = ix.lkt16[key16];
    bool same = l <= r ? (tl == l && tr == r) : tl > tr;
    if (ix.lkt16u) {
        // the 16-byte entries must say the same, and for a 16-mer that occurs once: its position and the 16 bases in front of it
        const uint4 e = ix.lkt16u[key16];
        if (l <= r) {
            same = same && e.x == l && e.y == r;
            if (l == r) {
                const uint32_t pos = ix.sa_dense[l];
                if (pos >= 16 && (uint64_t)pos + 16 <= ix.n) {
                    uint32_t prev = 0;
                    for (int u = 0; u < 16; u++) { const uint32_t pp = pos - 1 - (uint32_t)u; prev |= ((ix.text[pp >> 4] >> (2 * (pp & 15))) & 3u) << (2 * u); }
                    same = same && e.z == pos && e.w == prev;
                } else same = same && e.z == 0xFFFFFFFFu;
            }
        } else same = same && e.x > e.y;
    }
    atomicAdd(&out[0], 1ull);
    if (!same) atomicAdd(&out[1], 1ull);
}

// State of one (read, strand) backward search -- the variables of mmp<0>/mmp<2> (DV-DPfunctions.cpp:2224-2377)
struct mp_mmp_state {
    const uint32_t *sc;     // scan-order bases of this item
    uint32_t rs;
    int len, strand, i, seed_len;
    uint32_t l, r, last_l, last_r, last_seed_len;
    uint32_t cur_w, cur_word;
    uint32_t n_seeds, n_hits, n_located;   // n_located: seeds reported by text position (one hit each, no suffix-array read in locate)
    unsigned long long pe;  // pre-filter bits of the 15-mer at scan bases pg + 2 .. pg + 16 (ix.kmer15x3): answers positions pg .. pg + 2
    int pg;
    uint32_t upos;          // text position of the match when its SA range was a single row and it was extended by text
    bool uvalid;            // comparison (mp_mmp_trip): the record then carries the position instead of the row
    uint32_t uprev;         // the 16 text bases in front of upos (scan order), when the 16-byte table entry supplied them
    bool uprev_valid;
};
#define MP_SEED_LOCATED 0xFFFFFFFFu     // mp_seed_rec::sa_diff / mp_sa_seed::sa_diff: sa_l is a text position, one hit

template <int SLAB, class State>
__device__ __forceinline__ void mp_close_seed(State &s, const mp_dev_mmp &P, uint32_t n, int x, mp_seed_rec *slab)
{
    // CHECK_AND_ADD_RANGE (DV-DPfunctions.cpp:2197-2219)
    int diff = 0;
    if (s.seed_len >= P.seed_min_length) {
        // reseed to the last shorter seed whose SA range was larger; comparisons unsigned as in the reference
        if (s.seed_len >= P.reseed_len && (uint64_t)s.last_r - s.last_l + 1 <= (uint64_t)P.sa_size_threshold &&
            ((uint64_t)(uint32_t)s.seed_len - s.last_seed_len <= (uint64_t)P.reseed_abs_diff ||
             s.seed_len * P.reseed_rlt_ratio < (double)s.last_seed_len)) {
            diff = s.seed_len - (int)s.last_seed_len;
            s.l = s.last_l; s.r = s.last_r; s.seed_len = (int)s.last_seed_len;
        }
        const uint32_t sa_diff = (s.r - s.l) < (uint32_t)P.sa_size_threshold ? (s.r - s.l) : (uint32_t)P.sa_size_threshold;
        const bool located = s.uvalid && diff == 0;        // extended by text comparison and not re-seeded: the position is known
        if (s.n_seeds < (uint32_t)SLAB) {
            mp_seed_rec rec;
            rec.sa_l = located ? s.upos : s.l; rec.query_offset = (uint16_t)(x & 1023); rec.seed_len = (uint16_t)s.seed_len;
            rec.sa_diff = located ? MP_SEED_LOCATED : sa_diff;
            slab[s.n_seeds] = rec;
        }
        s.n_seeds++;
        s.n_hits += located ? 1u : sa_diff + 1;
        s.n_located += located ? 1u : 0u;
    }
    s.i -= diff;
    s.i -= s.seed_len < P.seed_min_length ? s.seed_len : P.seed_min_length;
    s.l = 0; s.r = n; s.seed_len = 0;
    s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
    s.uvalid = false; s.uprev_valid = false;
}

// Per-lane event counters of the seeding kernel: only events that touch memory are priced by the roofline (bench.py);
// `skipped` records the backward-search steps that never happened because a table or the text stood in for them.
struct mp_lane_events {
    uint32_t occ;        // rank queries issued (2 per extension step carried out on the rank blocks)
    uint32_t blk;        // 64-byte rank blocks loaded for them (1 when both queries fall into one block)
    uint32_t lkt;        // 13-mer table look-ups (two adjacent words)
    uint32_t lkt16;      // 16-mer table look-ups (two adjacent words)
    uint32_t kmer;       // pre-filter probes answered
    uint32_t kload;      // pre-filter entries loaded for them (one 8-byte entry answers up to three probes)
    uint32_t text;       // 16-base text words compared in place of rank queries
    uint32_t sa;         // suffix-array entries read to find the text position of a single-row match
    uint32_t skipped;    // extension steps not carried out: three per 16-mer look-up, one per base matched in the text
    uint32_t located;    // seeds handed to locate by their text position: their hit needs no suffix-array read there
};

// Runs one item to completion or for one trip.  Returns true when the item is finished.
template <int SLAB>
__device__ __forceinline__ bool mp_mmp_trip(const mp_dev_index &ix, const mp_dev_mmp &P, mp_mmp_state &s, mp_seed_rec *slab, mp_lane_events &ev)
{
    const uint32_t n = ix.n;
    bool finished = false;
    // ---- at most one LKT (re)start.  Futile starts (the pre-filter says the next 17 bases occur nowhere) cost no look-up: the scan
    // moves on one base, and keeps moving within this trip for as long as the pre-filter entry already loaded answers the next
    // position too (one 8-byte entry covers three consecutive positions) -- a strand that matches nowhere is walked three positions
    // per trip, with at most one pre-filter load per trip as before.
    bool loaded_entry = false;
    while (s.seed_len == 0 && !finished) {
        if (s.len - s.i < P.seed_min_length) { finished = true; break; }   // `break` of the scan loop
        const uint32_t wq = (uint32_t)s.i >> 4, sh = 2 * ((uint32_t)s.i & 15);
        const uint64_t two = ((uint64_t)s.sc[wq + 1] << 32) | s.sc[wq];
        bool futile = false;
        if (ix.kmer17 && P.seed_min_length >= (int)ix.kmer_k) {
            // pre-filter: the next 17 (16) scanned bases (there are at least seed_min_length left) as one bit probe
            const uint64_t k17 = (two >> sh) & ((1ull << (2 * ix.kmer_k)) - 1);
            if (ix.kmer15x3) {
                const int a = s.i - s.pg;
                if (a < 0 || a > 2) {
                    if (loaded_entry) break;                  // the next entry is for the next trip
                    s.pg = s.i; s.pe = ix.kmer15x3[(k17 >> 4) & ((1ull << 30) - 1)]; ev.kload++; loaded_entry = true;
                }
                const int aa = s.i - s.pg;
                const uint32_t bit = aa == 0 ? (uint32_t)(k17 & 15) : aa == 1 ? 16u + ((uint32_t)(k17 & 3) | ((uint32_t)(k17 >> 32) & 3u) << 2)
                                                                               : 32u + (uint32_t)((k17 >> 30) & 15);
                futile = ((s.pe >> bit) & 1ull) == 0;
            } else {
                if (loaded_entry) break;
                futile = ((ix.kmer17[k17 >> 5] >> (uint32_t)(k17 & 31)) & 1u) == 0;
                ev.kload++; loaded_entry = true;
            }
            ev.kmer++;
        }
        const uint32_t key16 = (uint32_t)(two >> sh);
        if (futile) {
            ++s.i;                                            // a seed shorter than seed_min_length: nothing reported, restart one base on
            continue;
        }
        if (ix.lkt16 && ix.kmer17 && ix.kmer_k == 17 && P.seed_min_length >= 17 && (key16 & (MP_LKT_ENTRIES - 1)) != 0) {
            // the 17-mer is reachable, so its first 16 bases are: their SA range straight from the 16-mer table instead of
            // the 13-mer table and three extension steps (all below seed_min_length, none observable).  Keys whose 13
            // rightmost bases are 'A' keep the 13-mer path (kernels_index.hip, lkt16_hist_kernel).
            uint32_t nextl, nextr;
            if (ix.lkt16u) {
                // one 16-byte entry: the range and, for a 16-mer that occurs once, where -- and what stands in front of it
                const uint4 e = ix.lkt16u[key16];
                nextl = e.x; nextr = e.y;
                if (nextl == nextr && e.z != 0xFFFFFFFFu) { s.upos = e.z; s.uvalid = true; s.uprev = e.w; s.uprev_valid = true; }
            } else { nextl = ix.lkt16[key16 - 1] + 1; nextr = ix.lkt16[key16]; }
            ev.lkt16++;
            ev.skipped += 3;                              // the three extension steps the table stands for
            s.i += 15;
            s.seed_len = 15;
            if (nextl <= nextr) { s.l = nextl; s.r = nextr; ++s.seed_len; }
            else mp_close_seed<SLAB>(s, P, n, s.strand == 0 ? s.len - s.i : s.i - s.seed_len, slab);
            ++s.i;
            if (s.i >= s.len) finished = true;
        } else {
            const uint32_t key = (uint32_t)(two >> sh) & (MP_LKT_ENTRIES - 1);
            uint32_t nextl, nextr;
            mp_lkt13_range(ix, key, nextl, nextr);
            ev.lkt++;
            s.i += MP_LKT_K - 1;
            s.seed_len = MP_LKT_K - 1;
            if (nextl <= nextr) { s.l = nextl; s.r = nextr; ++s.seed_len; }
            else mp_close_seed<SLAB>(s, P, n, s.strand == 0 ? s.len - s.i : s.i - s.seed_len, slab);
            ++s.i;
            if (s.i >= s.len) finished = true;
        }
        break;                                                // one look-up per trip
    }
    // ---- a match whose SA range is a single row is extended by comparing the read with the text itself: the next base of
    // the backward search is the text base in front of the match, 16 of them per word instead of one rank block each.  The
    // rows the search would pass through are never needed: the range cannot shrink any more (so the re-seed bookkeeping
    // stands still) and the seed is reported by its text position (MP_SEED_LOCATED).  Needs the whole suffix array.
    if (!finished && s.seed_len > 0 && s.l == s.r && ix.sa_dense && ix.sa_dense_shift == 0 && s.l != 0) {
        const uint32_t pos = s.uvalid ? s.upos : ix.sa_dense[s.l];
        if (!s.uvalid) ev.sa++;
        const int avail = min(s.len - s.i, (int)min(pos, (uint32_t)1 << 20));    // scan bases left / text bases in front of the match
        int matched = 0;
        bool stop = false;
        while (matched < avail && !stop) {
            const int si = s.i + matched;
            const uint32_t swq = (uint32_t)si >> 4, ssh = 2 * ((uint32_t)si & 15);
            const uint32_t scan16 = (uint32_t)((((uint64_t)s.sc[swq + 1] << 32) | s.sc[swq]) >> ssh);     // scan base si in bits 1:0
            const uint32_t pend = pos - (uint32_t)matched - 1;                    // text base that must equal it
            int nn = avail - matched; if (nn > 16) nn = 16;
            uint32_t trev;
            if (matched == 0 && s.uprev_valid) { trev = s.uprev; s.uprev_valid = false; ev.text--; }   // came with the table entry: no text load
            else if (pend >= 15) {
                const uint32_t p0 = pend - 15;
                const uint32_t t16 = (uint32_t)((((uint64_t)ix.text[(p0 >> 4) + 1] << 32) | ix.text[p0 >> 4]) >> (2 * (p0 & 15)));
                const uint32_t r = __brev(t16);
                trev = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);      // text base pend in bits 1:0, pend - 1 in 3:2, ...
            } else {
                trev = 0;
                for (int u = 0; u < nn; u++) { const uint32_t pp = pend - (uint32_t)u; trev |= ((ix.text[pp >> 4] >> (2 * (pp & 15))) & 3u) << (2 * u); }
            }
            ev.text++;
            const uint32_t x = scan16 ^ trev;
            int same = x ? (__ffs((int)x) - 1) >> 1 : 16;
            if (same >= nn) same = nn; else stop = true;
            matched += same;
        }
        s.upos = pos - (uint32_t)matched; s.uvalid = true;
        s.seed_len += matched; s.i += matched;
        ev.skipped += (uint32_t)(matched + (s.i < s.len ? 1 : 0));     // steps the comparison stands for (the last one is the failing step)
        if (s.i >= s.len) finished = true;
        else {
            // the base at s.i does not extend the match (or the text starts here): what the failing rank query reports
            mp_close_seed<SLAB>(s, P, n, s.strand == 0 ? s.len - s.i : s.i - s.seed_len, slab);
            ++s.i;
            if (s.i >= s.len) finished = true;
        }
    } else
    // ---- at most one backward extension
    if (!finished && s.seed_len > 0) {
        const uint32_t wq = (uint32_t)s.i >> 4;
        if (wq != s.cur_w) { s.cur_w = wq; s.cur_word = s.sc[wq]; }
        const uint32_t c = (s.cur_word >> (2 * ((uint32_t)s.i & 15))) & 3u;
        uint32_t o1, o2;
        mp_occ_pair(ix, s.l, s.r + 1, c, o1, o2);
        ev.occ += 2;
        {
            uint32_t k1 = s.l, k2 = s.r + 1;
            k1 -= (k1 > ix.inv_sa0); k2 -= (k2 > ix.inv_sa0);
            ev.blk += (k1 / MP_OCC_SYMS == k2 / MP_OCC_SYMS) ? 1u : 2u;
        }
        const uint32_t nextl = ix.cum[c] + o1 + 1, nextr = ix.cum[c] + o2;
        if (nextl <= nextr) {
            if (s.seed_len >= P.seed_min_length && nextr - nextl < s.r - s.l) { s.last_r = s.r; s.last_l = s.l; s.last_seed_len = (uint32_t)s.seed_len; }
            s.l = nextl; s.r = nextr; ++s.seed_len;
        } else {
            mp_close_seed<SLAB>(s, P, n, s.strand == 0 ? s.len - s.i : s.i - s.seed_len, slab);
        }
        ++s.i;
        if (s.i >= s.len) finished = true;
    }
    if (finished) mp_close_seed<SLAB>(s, P, n, s.strand == 0 ? 0 : s.len - s.seed_len, slab);
    return finished;
}

__device__ __forceinline__ void mp_mmp_begin(mp_mmp_state &s, const uint32_t *scan, const uint32_t *lens, uint32_t W, uint32_t rs, uint32_t n)
{
    s.sc = scan + (uint64_t)rs * mp_scan_stride(W);
    s.rs = rs; s.len = (int)lens[rs >> 1]; s.strand = (int)(rs & 1);
    // A read below the DP cutoff floor (30 bases, definitions.h:166-167) can never pass a DP (its best possible score is its
    // length), so it gets no seeds and therefore no DP instance: it ends up unaligned like any read without a hit.  (The
    // reference gives up on the whole SIMD group such an instance falls into, CPU_DP.cpp:294-299 -- not reproduced.)
    if (s.len < MP_MIN_ALIGNABLE_LEN) s.len = 0;
    s.i = 0; s.seed_len = 0; s.l = 0; s.r = n; s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
    s.cur_w = 0xFFFFFFFFu; s.cur_word = 0; s.n_seeds = 0; s.n_hits = 0; s.n_located = 0; s.upos = 0; s.uvalid = false; s.uprev = 0; s.uprev_valid = false; s.pe = 0; s.pg = -1000;
}

// ---- seed_kernel: the same state machine as mp_mmp_trip, arranged so that a wavefront waits for memory ONCE per trip.
// Every step of a backward search is a dependent load (pre-filter entry -> table entry -> rank block -> ... -> text word), and
// the 64 lanes of a wavefront sit in different steps.  Written as branches that load and then use (mp_mmp_trip), each kind of
// step costs the wavefront its own full memory latency per trip, one after the other.  Here a lane carries the ADDRESS of its
// next load in its state: a trip issues the loads of all lanes together (three instruction groups: a 64-byte block, one
// 16-byte entry, one or two words), waits once, then every lane consumes what it got and derives its next address.  The read's
// scan-order bases live in LDS (one row per lane), so cutting keys and comparing with the text costs no global load.
enum { MK_NONE = 0, MK_PRE, MK_LKT13, MK_LKT16, MK_SA, MK_TEXT, MK_LKT14,      // one or two words
       MK_LKT16U,                                                     // one 16-byte entry
       MK_OCC, MK_OCC2, MK_BEGIN };                                   // a 64-byte block

struct mp_lane_state {
    uint32_t rs;
    int len, strand, i, seed_len;
    uint32_t l, r, last_l, last_r, last_seed_len;
    uint32_t n_seeds, n_hits, n_located;
    uint32_t nf;            // pre-filter answers for scan positions pg .. pg + 5 (bits per 15-mer: two entries) or pg alone (17-mer bitmap): bit set = the
    int pg;                 // 17-mer starting there occurs in the text
    uint32_t upos;          // text position of the match once its SA range is a single row
    bool uvalid, uprev_valid;
    uint32_t kind;          // the load this lane waits for
    uint32_t aux;           // MK_OCC*: the symbol; MK_LKT13: key == 0; MK_BEGIN: 64-byte blocks of the row already fetched; MK_PRE: index of the second entry
    uint32_t o1;            // MK_OCC2: rank at the range's lower end, from the first block
    const void *ptr;        // where that load reads
};

// (94 VGPRs: five waves per SIMD.  Held to 80 registers for a sixth wave, and with two items per lane, the kernel was slower: DESIGN.md section 4)
// (Measured in round 4 and not kept: a build with the states of the optional tables compiled out for lean chunks -- 67 VGPRs, seven waves per SIMD -- runs the
// NT stage's seeding at the very same rate, 12.3 ms per 256 Ki pairs and chunk; holding this build to 96 VGPRs for a fifth wave made the human-filter step 5 % slower.)
__global__ void __launch_bounds__(256)
seed_kernel(mp_dev_index ix, const uint32_t *__restrict__ scan, const uint32_t *__restrict__ lens, uint32_t n_items, uint32_t W, mp_dev_mmp P,
            mp_seed_rec *__restrict__ slabs, uint32_t *__restrict__ seeds_per_rs, uint32_t *__restrict__ hits_per_rs,
            uint32_t *__restrict__ overflow_rs, uint32_t *__restrict__ over_slot, uint32_t overflow_cap, unsigned int *queue, mp_counters *ctr,
            uint32_t slab_limit /* <= MP_SEED_SLAB; smaller only to exercise the overflow path */)
{
    extern __shared__ uint32_t s_scan[];                 // [256][S | 1]: the scan row of the item each lane works on
    const int lane = threadIdx.x & (MP_WAVE - 1);
    const uint32_t S = mp_scan_stride(W);
    uint32_t *const sl = s_scan + threadIdx.x * (S | 1u);
    const uint32_t n = ix.n;
    const bool pf = ix.kmer17 && P.seed_min_length >= (int)ix.kmer_k;
    const uint64_t kmask = (1ull << (2 * ix.kmer_k)) - 1;             // (kmer15x3 and the 16-mer tables exist only with 17-mers)
    const bool text_ok = ix.sa_dense && ix.sa_dense_shift == 0;
    const bool packed = ix.lkt14 != nullptr;                          // the 13-mer table as 8-byte entries with the first extension step in them
    const bool lkt14_ok = packed && P.seed_min_length > MP_LKT_K;     // (a 13-base seed must not be reportable: nothing shorter than 14 bases leaves that look-up)
    const uint32_t cum0 = ix.cum[0], cum1 = ix.cum[1], cum2 = ix.cum[2], cum3 = ix.cum[3];
    mp_lane_events ev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mp_lane_state s;
    s.rs = 0; s.len = 0; s.strand = 0; s.i = 0; s.seed_len = 0; s.l = 0; s.r = n; s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
    s.n_seeds = 0; s.n_hits = 0; s.n_located = 0; s.nf = 0; s.pg = -1000; s.upos = 0; s.uvalid = false; s.uprev_valid = false;
    s.kind = MK_NONE; s.aux = 0; s.o1 = 0; s.ptr = nullptr;
    bool have = false;
    // Work queue: one global counter, but a wavefront takes MP_SEED_CHUNK items per atomic into a private range and hands
    // them to its lanes as they finish (a single word sustains only ~90 returning atomics per microsecond chip-wide).
    uint32_t pool_next = 0, pool_end = 0;       // wave-uniform
    bool drained = false;
    uint32_t wave_trips = 0, lane_busy = 0;
    for (;;) {
        wave_trips++;
        const unsigned long long need = __ballot(!have);
        if (need) {
            if (pool_next == pool_end && !drained) {
                uint32_t base = 0;
                const int leader = __ffsll((long long)need) - 1;
                if (lane == leader) base = atomicAdd(queue, (unsigned int)MP_SEED_CHUNK);
                base = __shfl(base, leader);
                if (base >= n_items) drained = true;
                else { pool_next = base; pool_end = base + MP_SEED_CHUNK < n_items ? base + MP_SEED_CHUNK : n_items; }
            }
            const uint32_t avail = pool_end - pool_next;
            if (avail) {
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (!have && rank < avail) {
                    // a new item: its first load is its scan row (and its length)
                    const uint32_t rs = pool_next + rank;
                    s.rs = rs; s.strand = (int)(rs & 1); s.len = 0; s.i = 0; s.seed_len = 0;
                    s.l = 0; s.r = n; s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
                    s.n_seeds = 0; s.n_hits = 0; s.n_located = 0; s.nf = 0; s.pg = -1000; s.upos = 0; s.uvalid = false;
                    s.kind = MK_BEGIN; s.aux = 0; s.ptr = scan + (uint64_t)rs * S;
                    have = true;
                }
                const uint32_t want = (uint32_t)__popcll(need);
                pool_next += want < avail ? want : avail;
            }
        }
        if (!__any(have)) { if (drained && pool_next == pool_end) break; else continue; }   // nothing to do this trip
        // ---------------------------------------------------------------- issue: every lane's pending load, no use in between
        // (the destinations are written by exactly the lanes that read them below; left unset otherwise so that no copy or wait
        // is needed to carry a value around the loop)
        uint4 q0, q1, q2, q3;
        uint32_t a0, a1, b0, b1, alen;
        asm volatile("" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1), "=v"(alen));    // "some value": nothing is carried over from the trip before
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
        if (have) {
            lane_busy++;
            const uint32_t kind = s.kind;
            if (kind >= MK_LKT16U) {
                const uint4 *p = reinterpret_cast<const uint4 *>(s.ptr);
                q0 = p[0];
                if (kind != MK_LKT16U) { q1 = p[1]; q2 = p[2]; q3 = p[3]; }
                if (kind == MK_BEGIN && s.aux == 0) alen = lens[s.rs >> 1];
            } else if (kind != MK_NONE) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(s.ptr);
                const bool one = kind == MK_SA || (kind == MK_PRE && !ix.kmer15x3) || (kind == MK_LKT13 && s.aux);
                a0 = p[0];
                a1 = p[one ? 0 : (packed && kind == MK_LKT13) ? 2 : 1];      // (packed: lkt[k - 1] and lkt[k] are the first words of two neighbouring entries)
                if (kind == MK_PRE && ix.kmer15x3) { const uint2 e2 = *reinterpret_cast<const uint2 *>(ix.kmer15x3 + s.aux); b0 = e2.x; b1 = e2.y; }
            }
        }
        // the one wait of the trip: everything issued above (and the stores of the trip before) has landed
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), expcnt and lgkmcnt left alone
        // ---------------------------------------------------------------- consume
        if (have) {
            const uint32_t kind = s.kind;
            uint32_t next = MK_NONE;
            bool fin = false, mid_close = false, text_now = false;
            uint32_t trev = 0;
            if (kind == MK_BEGIN) {
                const uint32_t base = s.aux * 16u;
                if (base < S) { sl[base] = q0.x; sl[base + 1] = q0.y; sl[base + 2] = q0.z; sl[base + 3] = q0.w; }
                if (base + 4 < S) { sl[base + 4] = q1.x; sl[base + 5] = q1.y; sl[base + 6] = q1.z; sl[base + 7] = q1.w; }
                if (base + 8 < S) { sl[base + 8] = q2.x; sl[base + 9] = q2.y; sl[base + 10] = q2.z; sl[base + 11] = q2.w; }
                if (base + 12 < S) { sl[base + 12] = q3.x; sl[base + 13] = q3.y; sl[base + 14] = q3.z; sl[base + 15] = q3.w; }
                if (s.aux == 0) {
                    // A read below the DP cutoff floor (30 bases, definitions.h:166-167) can never pass a DP (its best possible score
                    // is its length), so it gets no seeds and therefore no DP instance: it ends up unaligned like any read without a
                    // hit.  (The reference gives up on the whole SIMD group such an instance falls into, CPU_DP.cpp:294-299 -- not reproduced.)
                    const int ln = (int)alen;
                    s.len = ln < MP_MIN_ALIGNABLE_LEN ? 0 : ln;
                }
                s.aux++;
                if (s.aux * 16u < S) { next = MK_BEGIN; s.ptr = reinterpret_cast<const uint4 *>(s.ptr) + 4; }
            } else if (kind == MK_LKT14 && ((a1 >> (8 * s.aux)) & 255u) == 255u) {
                // no short cut for this entry (mp_dev_index::lkt14): the 13-mer path, one trip more
                const uint32_t key = (uint32_t)(((((uint64_t)sl[((uint32_t)s.i >> 4) + 1] << 32) | sl[(uint32_t)s.i >> 4]) >> (2 * ((uint32_t)s.i & 15)))) & (MP_LKT_ENTRIES - 1);
                next = MK_LKT13; s.aux = 0; s.ptr = ix.lkt14 + key; ev.lkt++;
            } else if (kind == MK_LKT13 || kind == MK_LKT16 || kind == MK_LKT16U || kind == MK_LKT14) {
                // LKT (re)start: the table stands for the first 13 (14, 16) bases of the seed
                uint32_t nextl, nextr;
                int k;
                if (kind == MK_LKT13) { nextl = s.aux ? 1u : a0 + 1; nextr = a1; k = MP_LKT_K - 1; ev.lkt++; }
                else if (kind == MK_LKT14) {
                    // a0 = rows before the 13-mer Q (the 14-mer's first 13 text bases), a1 = the four counts; s.aux = the 14-mer's last text base =
                    // the seed's first scan base.  An entry with any byte at 255 was sent to the 13-mer path above or holds 255 in all four.
                    const uint32_t d = s.aux;
                    const uint32_t below = (d > 0 ? (a1 & 255u) : 0u) + (d > 1 ? ((a1 >> 8) & 255u) : 0u) + (d > 2 ? ((a1 >> 16) & 255u) : 0u);
                    nextl = a0 + below + 1; nextr = a0 + below + ((a1 >> (8 * d)) & 255u);
                    k = MP_LKT_K; ev.lkt++; ev.skipped++;                  // the one extension step the entry stands for
                }
                else {
                    if (kind == MK_LKT16U) { nextl = q0.x; nextr = q0.y; } else { nextl = a0 + 1; nextr = a1; }
                    k = 15; ev.lkt16++; ev.skipped += 3;                  // the three extension steps the table stands for
                }
                s.i += k; s.seed_len = k;
                if (nextl <= nextr) {
                    s.l = nextl; s.r = nextr; ++s.seed_len; ++s.i;
                    if (s.i >= s.len) fin = true;
                    else if (kind == MK_LKT16U && nextl == nextr && q0.z != 0xFFFFFFFFu) {
                        // a 16-mer that occurs once: the entry says where, and what stands in front of it -- compared right away
                        s.upos = q0.z; s.uvalid = true; trev = q0.w; text_now = true;
                    }
                } else mid_close = true;
            } else if (kind == MK_OCC || kind == MK_OCC2) {
                // one backward extension: rank of the symbol at both ends of the range (one block, or two over two trips)
                const uint32_t c = s.aux;
                uint32_t k1 = s.l, k2 = s.r + 1;
                k1 -= (k1 > ix.inv_sa0); k2 -= (k2 > ix.inv_sa0);
                const uint32_t b1 = k1 / MP_OCC_SYMS, off1 = k1 - b1 * MP_OCC_SYMS;
                const uint32_t b2 = k2 / MP_OCC_SYMS, off2 = k2 - b2 * MP_OCC_SYMS;
                mp_blk_view v;
                mp_blk_match(q0, q1, q2, q3, c, v);
                const uint32_t ca = mp_blk_rank(v, kind == MK_OCC ? off1 : off2);
                uint32_t o1 = ca, o2 = ca;
                bool done = true;
                if (kind == MK_OCC) {
                    if (b1 == b2) o2 = mp_blk_rank(v, off2);
                    else { s.o1 = o1; next = MK_OCC2; s.ptr = ix.occ + b2; done = false; }
                } else o1 = s.o1;
                if (done) {
                    ev.occ += 2; ev.blk += b1 == b2 ? 1u : 2u;
                    const uint32_t cumc = c == 0 ? cum0 : c == 1 ? cum1 : c == 2 ? cum2 : cum3;     // (ix.cum[c] would be a load from the argument block)
                    const uint32_t nextl = cumc + o1 + 1, nextr = cumc + o2;
                    if (nextl <= nextr) {
                        if (s.seed_len >= P.seed_min_length && nextr - nextl < s.r - s.l) { s.last_r = s.r; s.last_l = s.l; s.last_seed_len = (uint32_t)s.seed_len; }
                        s.l = nextl; s.r = nextr; ++s.seed_len; ++s.i;
                        if (s.i >= s.len) fin = true;
                    } else mid_close = true;
                }
            } else if (kind == MK_SA) {
                // the text position of a match whose range became a single row: from here on it is extended by comparing with the text
                s.upos = a0; s.uvalid = true;
                if (a0 == 0) { mid_close = true; ev.skipped++; }           // the text starts here: what the failing rank query reports
            } else if (kind == MK_TEXT) {
                const uint32_t pend = s.upos - 1;                          // text base that must equal scan base s.i
                uint32_t t16;                                               // text bases pend-15 .. pend, LSB-first
                if (pend >= 15) t16 = (uint32_t)((((uint64_t)a1 << 32) | a0) >> (2 * ((pend - 15) & 15)));
                else t16 = a0 << (2 * (15 - pend));
                const uint32_t r = __brev(t16);
                trev = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);    // text base pend in bits 1:0, pend - 1 in 3:2, ...
                text_now = true; ev.text++;
            }
            if (text_now) {
                // The next base of the backward search is the text base in front of the match, 16 of them per word instead of one
                // rank block each.  The range cannot shrink any more (the re-seed bookkeeping stands still) and the seed is
                // reported by its text position (MP_SEED_LOCATED).
                const int avail = min(s.len - s.i, (int)min(s.upos, (uint32_t)1 << 20));     // scan bases left / text bases in front of the match
                const int nn = avail < 16 ? avail : 16;
                const uint32_t wq = (uint32_t)s.i >> 4;
                const uint32_t scan16 = (uint32_t)((((uint64_t)sl[wq + 1] << 32) | sl[wq]) >> (2 * ((uint32_t)s.i & 15)));
                const uint32_t x = scan16 ^ trev;
                int same = x ? (__ffs((int)x) - 1) >> 1 : 16;
                const bool stop = same < nn;
                if (!stop) same = nn;
                s.upos -= (uint32_t)same; s.seed_len += same; s.i += same; ev.skipped += (uint32_t)same;
                if (stop || same == avail) {
                    if (s.i >= s.len) fin = true;
                    else { mid_close = true; ev.skipped++; }               // the failing step
                }
            }
            mp_seed_rec *slab = slabs + (uint64_t)s.rs * MP_SEED_SLAB;
            const int i_end = s.i;                                     // (where a seed that ends in this trip ended: the base that did not extend it)
            if (mid_close || fin) {
                const int x = mid_close ? (s.strand == 0 ? s.len - s.i : s.i - s.seed_len) : (s.strand == 0 ? 0 : s.len - s.seed_len);
                mp_close_seed<MP_SEED_SLAB>(s, P, n, x, slab);
                if (mid_close) { ++s.i; fin = s.i >= s.len; }
            }
            // ---- a lane without a seed under way: where does the next one start?  (pre-filter, then the table look-up)
            if (!fin && next == MK_NONE && s.seed_len == 0) {
                // the scan bases from here on as a 96-bit window: enough for the six positions two pre-filter entries answer and for the
                // keys of the two entries after them
                uint32_t wq0 = (uint32_t)s.i >> 4;
                uint32_t w0 = sl[wq0], w1 = sl[wq0 + 1], w2 = sl[wq0 + 2];
                auto bases_at = [&](int pos) -> uint64_t {                   // scan bases pos .. pos + 16 at least (pos - 16 wq0 <= 31)
                    const uint32_t off = (uint32_t)pos - 16u * wq0;
                    const uint64_t win = off < 16 ? (((uint64_t)w1 << 32) | w0) : (((uint64_t)w2 << 32) | w1);
                    return win >> (2 * (off & 15));
                };
                if (kind == MK_PRE) {
                    // what just arrived, as one bit per position: is the 17-mer that starts there anywhere in the text?  s.nf = present
                    // bits (low half) and known bits (high half) of the positions pg .. pg + 15; the two entries answer three positions each,
                    // from the offsets the lane chose when it asked (kept in s.nf meanwhile)
                    uint32_t nf;
                    if (ix.kmer15x3) {
                        const unsigned long long e1 = ((unsigned long long)a1 << 32) | a0, e2 = ((unsigned long long)b1 << 32) | b0;
                        const int o1 = (int)(s.nf & 255u), o2 = (int)((s.nf >> 8) & 255u);
                        nf = ((7u << o1) | (7u << o2)) << 16;
#pragma unroll
                        for (int q = 0; q < 6; q++) {
                            const int off = (q < 3 ? o1 : o2 - 3) + q;
                            const uint64_t k17 = bases_at(s.i + off) & ((1ull << 34) - 1);
                            const int aa = q % 3;
                            const uint32_t bit = aa == 0 ? (uint32_t)(k17 & 15) : aa == 1 ? 16u + ((uint32_t)(k17 & 3) | ((uint32_t)(k17 >> 32) & 3u) << 2)
                                                                                          : 32u + (uint32_t)((k17 >> 30) & 15);
                            nf |= (uint32_t)(((q < 3 ? e1 : e2) >> bit) & 1ull) << off;
                        }
                    } else nf = ((a0 >> (uint32_t)(bases_at(s.i) & 31)) & 1u) | 0x10000u;       // (bit = the key's low five bits, whatever its length)
                    s.nf = nf; s.pg = s.i;
                }
                const int lim = s.len - P.seed_min_length;                    // the last position a seed may start at (`break` of the scan loop beyond)
                if (s.i > lim) fin = true;
                else {
                    bool start_here = true;
                    if (pf) {
                        // pre-filter: is the 17-mer that starts at a scan position anywhere in the text?  A futile start (a seed shorter than
                        // seed_min_length: nothing reported, restart one base on -- CHECK_AND_ADD_RANGE) costs no look-up.  A seed is reported
                        // only from seed_min_length bases on, so a start at p is futile when ANY of the 17-mers at p .. p + seed_min_length - 17
                        // is absent (`need` windows: six with soap4.ini's 22-base seeds, one with 17), and an absent 17-mer at z rules out every
                        // start in [z - (need - 1), z]: the scan jumps behind the highest known such z, again and again.  That is all the
                        // correctness there is to it -- WHICH positions a lane asks about is free: with six windows per start the two entries
                        // of a trip are taken six positions apart (p + 5 .. 7 and p + 11 .. 13: a read's mismatch makes 22 starts futile, two
                        // trips rule them out), and a start is tried as soon as nothing known speaks against it and something known speaks for it.
                        const int need = ix.kmer15x3 ? min(P.seed_min_length - (int)ix.kmer_k + 1, 6) : 1;
                        const uint32_t present = s.nf & 0xFFFFu, known = s.nf >> 16, zeros = known & ~present, wmask = (1u << need) - 1u;
                        int a = s.i - s.pg;
                        if (a >= 0 && a < 16) {
                            const int a_in = a;
                            for (;;) {
                                const uint32_t win = (zeros >> a) & wmask;     // known absent among the windows of a start at pg + a
                                if (!win) break;
                                a += 32 - __clz((int)win);
                            }
                            const int p = s.pg + a;
                            if (p > lim) { ev.kmer += (uint32_t)(lim - s.i + 1); s.i = lim + 1; fin = true; start_here = false; }
                            else {
                                // (not right after a seed ended: what is known then is mostly about the stretch that just matched)
                                start_here = ((present >> a) & wmask) != 0 && !(mid_close && need > 1);
                                ev.kmer += (uint32_t)(a - a_in) + (start_here ? 1u : 0u);
                                s.i = p;
                                if (a != a_in) { wq0 = (uint32_t)p >> 4; w0 = sl[wq0]; w1 = sl[wq0 + 1]; w2 = sl[wq0 + 2]; }     // the window moves with the scan
                            }
                        } else start_here = false;
                        if (!fin && !start_here) {
                            next = MK_PRE;
                            if (ix.kmer15x3) {
                                // two entries per trip.  Six windows per start: the first entry at the window that contains the base the last seed
                                // ended at (i_end - 16, when that is among this start's windows) or at the start's last window, the second six on.
                                int o1 = 0, o2 = 3;
                                if (need == 6) { o1 = mid_close ? min(max(i_end - 16 - s.i, 0), 5) : 5; o2 = o1 + 6; }
                                s.ptr = ix.kmer15x3 + ((bases_at(s.i + o1) >> 4) & ((1ull << 30) - 1));
                                s.aux = (uint32_t)((bases_at(s.i + o2) >> 4) & ((1ull << 30) - 1));
                                s.nf = (uint32_t)o1 | ((uint32_t)o2 << 8);
                                s.pg = -1000;
                                ev.kload += 2;
                            } else { s.ptr = ix.kmer17 + ((bases_at(s.i) & kmask) >> 5); s.pg = -1000; ev.kload++; }
                        }
                    }
                    if (start_here) {
                        const uint32_t key16 = (uint32_t)bases_at(s.i);
                        if (pf && ix.lkt16 && (key16 & (MP_LKT_ENTRIES - 1)) != 0) {
                            // the 17-mer is reachable, so its first 16 bases are: their SA range straight from the 16-mer table instead of
                            // the 13-mer table and three extension steps (all below seed_min_length, none observable).  Keys whose 13
                            // rightmost bases are 'A' keep the 13-mer path (kernels_index.hip, lkt16_hist_kernel).
                            if (ix.lkt16u) { next = MK_LKT16U; s.ptr = ix.lkt16u + key16; }
                            else { next = MK_LKT16; s.ptr = ix.lkt16 + (key16 - 1); }
                        } else {
                            const uint32_t key = key16 & (MP_LKT_ENTRIES - 1);
                            if (lkt14_ok) { next = MK_LKT14; s.aux = key16 & 3u; s.ptr = ix.lkt14 + ((key16 >> 2) & (MP_LKT_ENTRIES - 1)); }
                            else if (packed) { next = MK_LKT13; s.aux = 0; s.ptr = ix.lkt14 + key; }
                            else { next = MK_LKT13; s.aux = key == 0; s.ptr = ix.lkt + (key ? key - 1 : 0); }
                        }
                    }
                }
            }
            // ---- a lane with a seed under way: its next extension
            if (!fin && next == MK_NONE) {
                if (text_ok && s.l == s.r && s.l != 0) {
                    if (s.uvalid) {
                        const uint32_t pend = s.upos - 1;
                        next = MK_TEXT; s.ptr = ix.text + ((pend >= 15 ? pend - 15 : 0u) >> 4);
                    } else { next = MK_SA; s.ptr = ix.sa_dense + s.l; ev.sa++; }
                } else {
                    const uint32_t c = (sl[(uint32_t)s.i >> 4] >> (2 * ((uint32_t)s.i & 15))) & 3u;
                    uint32_t k1 = s.l;
                    k1 -= (k1 > ix.inv_sa0);
                    next = MK_OCC; s.aux = c; s.ptr = ix.occ + k1 / MP_OCC_SYMS;
                }
            }
            if (fin) {
                seeds_per_rs[s.rs] = s.n_seeds;
                hits_per_rs[s.rs] = s.n_hits;
                ev.located += s.n_located;
                if (s.n_seeds > slab_limit) {
                    uint32_t slot = (uint32_t)atomicAdd(&ctr->overflow, 1ull);
                    if (slot < overflow_cap) { overflow_rs[slot] = s.rs; over_slot[s.rs] = slot; }
                }
                have = false; next = MK_NONE;
            }
            s.kind = next;
        }
#pragma clang diagnostic pop
    }
    { uint32_t lb = lane_busy; for (int o = MP_WAVE / 2; o > 0; o >>= 1) lb += __shfl_down(lb, o);
      if (lane == 0) { atomicAdd(&ctr->n_wave_trips, (unsigned long long)wave_trips); atomicAdd(&ctr->n_lane_trips, (unsigned long long)lb); } }
    uint32_t tot[10] = {ev.occ, ev.blk, ev.lkt, ev.lkt16, ev.kmer, ev.kload, ev.text, ev.sa, ev.skipped, ev.located};
#pragma unroll
    for (int q = 0; q < 10; q++)
        for (int o = MP_WAVE / 2; o > 0; o >>= 1) tot[q] += __shfl_down(tot[q], o);
    if (lane == 0) {
        unsigned long long *dst[10] = {&ctr->n_occ, &ctr->n_occ_loaded, &ctr->n_lkt, &ctr->n_lkt16, &ctr->n_kmer, &ctr->n_kmer_loads, &ctr->n_text_words, &ctr->n_sa,
                                       &ctr->n_steps_skipped, &ctr->n_located};
#pragma unroll
        for (int q = 0; q < 10; q++) if (tot[q]) atomicAdd(dst[q], (unsigned long long)tot[q]);
    }
}

// Items with more than MP_SEED_SLAB seeds (repeat-rich or heavily mutated reads): redone with a slab that can hold the
// theoretical maximum of one seed per scan position.  Their rank queries were already counted by seed_kernel.
#define MP_SEED_BIG 384
__global__ void __launch_bounds__(64)
seed_overflow_kernel(mp_dev_index ix, const uint32_t *__restrict__ scan, const uint32_t *__restrict__ lens, uint32_t W, mp_dev_mmp P,
                     const uint32_t *__restrict__ overflow_rs, uint32_t n_over, mp_seed_rec *__restrict__ big_slabs)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_over) return;
    mp_mmp_state s;
    mp_mmp_begin(s, scan, lens, W, overflow_rs[t], ix.n);
    mp_lane_events ev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // already counted by seed_kernel: the events of the rerun add HBM traffic, not work
    mp_seed_rec *slab = big_slabs + (uint64_t)t * MP_SEED_BIG;
    while (!mp_mmp_trip<MP_SEED_BIG>(ix, P, s, slab, ev)) {}
}

// ------------------------------------------------------------------------------------------------ K2
// expand: one lane per (read, strand) lists the SA rows of its seeds as locate tasks at rs_base[rs]..  (no atomics:
// every hit has a fixed, input-determined slot, so the whole seed stage is run-to-run deterministic)
__global__ void __launch_bounds__(256)
expand_hits_kernel(const mp_seed_rec *__restrict__ slabs, const uint32_t *__restrict__ seeds_per_rs, const uint32_t *__restrict__ rs_base,
                   uint32_t n_items, const uint32_t *__restrict__ over_slot, const mp_seed_rec *__restrict__ big_slabs,
                   mp_sa_seed *__restrict__ seeds_out /* one per hit */, uint32_t *__restrict__ hit_row, mp_counters *ctr, uint32_t slab_limit)
{
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ns = rs < n_items ? seeds_per_rs[rs] : 0;
    uint32_t tot = ns;
    for (int o = MP_WAVE / 2; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    if ((threadIdx.x & (MP_WAVE - 1)) == 0 && tot) atomicAdd(&ctr->n_seeds, (unsigned long long)tot);
    if (ns == 0) return;
    const mp_seed_rec *src = slabs + (uint64_t)rs * MP_SEED_SLAB;
    if (ns > slab_limit) src = big_slabs + (uint64_t)over_slot[rs] * MP_SEED_BIG;
    uint32_t w = rs_base[rs];
    for (uint32_t j = 0; j < ns; j++) {
        const mp_seed_rec rec = src[j];
        mp_sa_seed s;
        s.read_strand = rs; s.sa_l = rec.sa_l; s.query_offset = rec.query_offset; s.seed_len = rec.seed_len; s.sa_diff = rec.sa_diff;
        if (rec.sa_diff == MP_SEED_LOCATED) { seeds_out[w] = s; hit_row[w] = rec.sa_l; w++; continue; }      // sa_l is the text position
        for (uint32_t k = 0; k <= rec.sa_diff; k++) { seeds_out[w] = s; hit_row[w] = rec.sa_l + k; w++; }
    }
}

// One lane walks TWO hits at once (t and t + half): an LF step is one dependent 64-byte gather and a wavefront that waits for ONE of them per
// lane spends 78 % of its time parked (profiles/r04_nt_pmc_sq_counters.json: 1.27 TB/s, 37 % of the gather rate the seeding kernel reaches in the
// same chunks).  Both blocks are requested before either is used; the walks are independent, so the order of the results is unchanged.
__global__ void __launch_bounds__(256)
locate_kernel(mp_dev_index ix, const mp_sa_seed *__restrict__ seeds, const uint32_t *__restrict__ hit_row, uint32_t n_tasks,
              const uint32_t *__restrict__ lens, mp_dev_mmp P, mp_seed_hit *__restrict__ hits, mp_counters *ctr)
{
    const uint32_t half = (n_tasks + 1) / 2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t my_lf = 0;
    if (t < half) {
        const uint32_t tt[2] = {t, t + half};
        const bool have[2] = {true, t + half < n_tasks};
        mp_sa_seed sd[2];
        uint32_t idx[2] = {0, 0}, steps[2] = {0, 0};
        bool located[2] = {true, true};
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (have[k]) { sd[k] = seeds[tt[k]]; idx[k] = hit_row[tt[k]]; located[k] = sd[k].sa_diff == MP_SEED_LOCATED; }    // located: the seeding kernel already knows the text position
        uint64_t sav[2] = {0, 0};
        const bool pow2 = !ix.sa_dense && ix.sa_shift != 0xFFFFFFFFu;
        if (pow2) {
            // sampled rows every 2^k: the two walks side by side
            const uint32_t mask = ix.sa_interval - 1;
            bool walk[2] = {have[0] && !located[0] && (idx[0] & mask) != 0, have[1] && !located[1] && (idx[1] & mask) != 0};
            while (walk[0] || walk[1]) {
                uint4 q[2][4];
                uint32_t sym_at[2] = {0, 0};
                bool load[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    load[k] = walk[k] && idx[k] != ix.inv_sa0;              // (the '$' row: LF = 0 without a look-up)
                    if (load[k]) {
                        uint32_t kk = idx[k] + 1;
                        kk -= (kk > ix.inv_sa0);
                        sym_at[k] = kk - 1;
                        const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + sym_at[k] / MP_OCC_SYMS);
                        q[k][0] = p[0]; q[k][1] = p[1]; q[k][2] = p[2]; q[k][3] = p[3];
                    }
                }
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (walk[k]) {
                        idx[k] = load[k] ? mp_lf_blk(ix, sym_at[k], q[k][0], q[k][1], q[k][2], q[k][3]) : 0u;
                        steps[k]++;
                        walk[k] = (idx[k] & mask) != 0;
                    }
            }
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (have[k]) {
                    const uint32_t qq = idx[k] >> ix.sa_shift;
                    // SA[0] is kept as -1 so that walking through the '$' row yields text position `steps - 1` (BWT.c:242,994-996)
                    sav[k] = located[k] ? (uint64_t)idx[k] : (qq == 0 ? ~0ull : (uint64_t)ix.sa[qq]) + steps[k];
                }
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (have[k]) sav[k] = located[k] ? (uint64_t)idx[k] : mp_sa_value(ix, idx[k], &steps[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (!have[k]) continue;
            const mp_sa_seed s = sd[k];
            const uint32_t read = s.read_strand >> 1, strand = s.read_strand & 1;
            const uint32_t len = lens[read];
            my_lf += steps[k];
            const uint32_t off = s.query_offset, seedlen = s.seed_len;
            // unsigned 64-bit arithmetic: the projection may wrap exactly as in the reference (DV-DPfunctions.cpp:2489)
            const uint64_t target = strand == 0 ? sav[k] - off : sav[k] - ((uint64_t)len - seedlen - off);
            mp_seed_hit h;
            h.offset = target;
            h.length = (uint16_t)seedlen;
            h.qo = (uint16_t)off;
            h.strand = (uint16_t)strand;
            h.mult = (seedlen >= (uint32_t)P.good_seed_len || seedlen >= len / 2 || located[k]) ? (uint16_t)1 : (uint16_t)(s.sa_diff + 1);
            hits[tt[k]] = h;
        }
    }
    // one counter update per block: a single address takes only ~100 M atomics per second chip-wide, and a launch of
    // 57 M hits has 890 k wavefronts
    __shared__ uint32_t s_lf[256 / MP_WAVE];
    for (int o = MP_WAVE / 2; o > 0; o >>= 1) my_lf += __shfl_down(my_lf, o);
    if ((threadIdx.x & (MP_WAVE - 1)) == 0) s_lf[threadIdx.x / MP_WAVE] = my_lf;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t lf = 0;
        for (int w = 0; w < 256 / MP_WAVE; w++) lf += s_lf[w];
        // an LF step is one rank query = one rank block (mp_lf).  Suffix-array reads are not counted here (one atomic per
        // block on one address would cost more than the kernel): n_sa = hits - located seeds, known after seed_kernel
        if (lf) { atomicAdd(&ctr->n_occ, (unsigned long long)lf); atomicAdd(&ctr->n_occ_loaded, (unsigned long long)lf); atomicAdd(&ctr->n_lf, (unsigned long long)lf); }
    }
    if (t == 0) atomicAdd(&ctr->n_hits, (unsigned long long)n_tasks);
}

// ------------------------------------------------------------------------------------------------ K3
// A located hit as one sortable word: strand | offset (32 bits: an index chunk holds < 2^32 bases) | qo | length | unique.
// Sorting by the word sorts by (strand, offset) -- SeedAlign order (DV-DPfunctions.cpp:2144-2160); the order among equal
// (strand, offset) does not matter to what follows (group start, union of intervals, any-unique).
__device__ __forceinline__ uint64_t hit_key(const mp_seed_hit &h, const mp_dev_mmp &P)
{
    const uint64_t uniq = ((int)h.mult <= P.uniq_threshold && (int)h.length >= P.seed_min_length) ? 1ull : 0ull;
    return ((uint64_t)h.strand << 52) | ((h.offset & 0xFFFFFFFFull) << 20) | ((uint64_t)(h.qo & 511u) << 11) | ((uint64_t)(h.length & 511u) << 2) | uniq;
}
#define MP_KEY_STRAND(k) ((uint32_t)((k) >> 52))
// The projection seed position - query offset is done in unsigned 64-bit arithmetic and may wrap below zero exactly as
// in the reference (DV-DPfunctions.cpp:2489).  Real positions are < text_len, so a 32-bit offset >= text_len can only be
// the low word of such a wrapped value (text_len <= 2^32 - MP_SEED_BIG): it is sign-extended back.
__device__ __forceinline__ uint64_t key_pos(uint64_t k, uint64_t text_len)
{
    const uint64_t o = (k >> 20) & 0xFFFFFFFFull;
    return o >= text_len ? (o | 0xFFFFFFFF00000000ull) : o;
}
#define MP_KEY_QO(k) ((int)(((k) >> 11) & 511u))
#define MP_KEY_LEN(k) ((int)(((k) >> 2) & 511u))
#define MP_KEY_UNIQ(k) ((bool)((k) & 1u))

#define MP_MERGE_BLOCK 64           // reads per block
#define MP_MERGE_THREADS 256        // four wavefronts share the block's staged keys
#define MP_MERGE_CAP 4096          // keys staged per block (32 KB)
#define MP_MERGE_SMALL 32          // insertion sort up to here when a lane sorts on its own (blocks that do not fit in LDS)
#define MP_MERGE_COOP 512          // up to here the wavefront sorts a read together, eight keys per lane
__device__ void sort_keys(uint64_t *a, int n)
{
    if (n <= MP_MERGE_SMALL) {
        for (int i = 1; i < n; i++) {
            const uint64_t t = a[i];
            int j = i - 1;
            while (j >= 0 && t < a[j]) { a[j + 1] = a[j]; j--; }
            a[j + 1] = t;
        }
        return;
    }
    // heap sort for the larger (repeat-rich) segments
    for (int start = n / 2 - 1; start >= 0; start--) {
        int root = start;
        const uint64_t v = a[root];
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            uint64_t cv = a[child];
            if (child + 1 < n) { const uint64_t c2 = a[child + 1]; if (cv < c2) { cv = c2; child++; } }
            if (!(v < cv)) break;
            a[root] = cv;
            root = child;
        }
        a[root] = v;
    }
    for (int end = n - 1; end > 0; end--) {
        const uint64_t v = a[end];
        a[end] = a[0];
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            uint64_t cv = a[child];
            if (child + 1 < end) { const uint64_t c2 = a[child + 1]; if (cv < c2) { cv = c2; child++; } }
            if (!(v < cv)) break;
            a[root] = cv;
            root = child;
        }
        a[root] = v;
    }
}

// seed merging of one read (DV-DPfunctions.cpp:2470-2560): k[0..n) are the SORTED keys of its located seed hits, out
// receives the kept seeds.
__device__ __forceinline__ void merge_read(uint64_t *k, int n, mp_seedpos *__restrict__ out, uint32_t r, const mp_dev_mmp &P, uint64_t text_len, int cov_words,
                                           uint32_t *__restrict__ n_pos, uint32_t *__restrict__ n_neg, mp_counters *ctr)
{
    uint32_t max_cov = 0;
    int n_kept = 0;
    for (int m = 0; m < n; m++) {
        const uint64_t first = k[m];
        const uint64_t pos = key_pos(first, text_len);
        const uint64_t group_end = pos + (uint64_t)P.indel_fuzz;
        bool has_unique = false;
        // covered read length = size of the union of [qo, qo+len) over the group (DV-DPfunctions.cpp:2518-2531 walks the
        // intervals in (qo, end) order and adds up the merged runs): a bitmap over the read positions, linear in the
        // group size -- repeat-rich reads have groups of hundreds of hits.  The group: while next.offset <= first.offset
        // + indelFuzz on the same strand (:2514).
        uint64_t cov[MP_SEED_BIG / 64];
#pragma unroll
        for (int w = 0; w < MP_SEED_BIG / 64; w++) cov[w] = 0;
        uint64_t cur = first;
        for (;;) {
            const int lo = MP_KEY_QO(cur), hi = lo + MP_KEY_LEN(cur);
            has_unique |= MP_KEY_UNIQ(cur);
#pragma unroll
            for (int w = 0; w < MP_SEED_BIG / 64; w++) {
                if (w >= cov_words) break;                 // reads of this batch are no longer than 64 cov_words bases
                int a = lo - 64 * w, e = hi - 64 * w;
                a = a < 0 ? 0 : (a > 64 ? 64 : a);
                e = e < 0 ? 0 : (e > 64 ? 64 : e);
                const uint64_t below_e = e >= 64 ? ~0ull : ((1ull << e) - 1ull), below_a = a >= 64 ? ~0ull : ((1ull << a) - 1ull);
                cov[w] |= below_e & ~below_a;
            }
            if (m + 1 >= n) break;
            cur = k[m + 1];
            if (MP_KEY_STRAND(cur) != MP_KEY_STRAND(first) || key_pos(cur, text_len) > group_end) break;
            m++;
        }
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < MP_SEED_BIG / 64; w++) total += (uint32_t)__popcll(cov[w]);
        if (max_cov < total) max_cov = total;
        if (has_unique || total >= (uint32_t)P.good_seed_len) {
            mp_seedpos sp;
            sp.pos = pos;
            sp.paired_seed_length = total;
            sp.strand_read_id = (r & ~1u) | (MP_KEY_STRAND(first) << 31);
            out[n_kept++] = sp;
        }
    }
    // short-seed filter against the best covered length of BOTH strands, rejected groups included (:2533,2547)
    int np = 0, nn = 0, w = 0;
    for (int j = 0; j < n_kept; j++) {
        mp_seedpos sp = out[j];
        if ((double)sp.paired_seed_length >= P.short_seed_ratio * (double)max_cov) {
            out[w++] = sp;
            if (sp.strand_read_id >> 31) nn++; else np++;
        }
    }
    n_pos[r] = (uint32_t)np;
    n_neg[r] = (uint32_t)nn;
    if (w) atomicAdd(&ctr->n_seedpos, (unsigned long long)w);      // no-return atomic; rare enough after the early exits
}

// Bitonic sort of 64 K keys held K per lane (element i lives in lane i & 63, register i >> 6): partners less than 64
// apart are exchanged with a wavefront shuffle, the others are register pairs of the same lane.  No LDS, no barrier.
template <int K>
__device__ __forceinline__ void wave_bitonic(uint64_t (&key)[K], int lane)
{
#pragma unroll
    for (int kk = 2; kk <= 64 * K; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int jr = j >> 6;
#pragma unroll
                for (int rg = 0; rg < K; rg++) {
                    if (rg & jr) continue;
                    const bool up = ((rg * 64) & kk) == 0;            // kk >= 128 here: the direction bit is a register bit
                    const uint64_t a = key[rg], b = key[rg | jr];
                    const bool sw = (a > b) == up;
                    key[rg] = sw ? b : a; key[rg | jr] = sw ? a : b;
                }
            } else {
#pragma unroll
                for (int rg = 0; rg < K; rg++) {
                    const uint64_t a = key[rg];
                    const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)a, j), phi = (uint32_t)__shfl_xor((int)(uint32_t)(a >> 32), j);
                    const uint64_t b = ((uint64_t)phi << 32) | plo;
                    const bool lower = (lane & j) == 0;
                    const bool up = (((rg * 64) | lane) & kk) == 0;
                    const bool keep_min = lower == up;
                    key[rg] = keep_min ? (a < b ? a : b) : (a > b ? a : b);
                }
            }
        }
    }
}

template <int K>
__device__ __forceinline__ void wave_sort_read(uint64_t *k, int n, int lane)
{
    uint64_t key[K];
#pragma unroll
    for (int rg = 0; rg < K; rg++) { const int i = rg * 64 + lane; key[rg] = i < n ? k[i] : ~0ull; }
    wave_bitonic<K>(key, lane);
#pragma unroll
    for (int rg = 0; rg < K; rg++) { const int i = rg * 64 + lane; if (i < n) k[i] = key[rg]; }
}

// Sort of n <= 64 keys, one per lane, by counting: every lane compares its key with each of the n keys (broadcast by v_readlane, no
// LDS round trip) and writes it to the slot of its rank.  Four instructions per key instead of the 21 dependent exchange stages of
// the bitonic network; equal keys are ranked by lane, so the ranks are a permutation.
__device__ __forceinline__ void wave_rank_sort(uint64_t *k, int n, int lane)
{
    const uint64_t key = lane < n ? k[lane] : ~0ull;
    const uint64_t sk = lane < n ? (key << 6) | (uint64_t)lane : ~0ull;       // keys use 53 bits: the lane number makes them distinct
    const uint32_t lo = (uint32_t)sk, hi = (uint32_t)(sk >> 32);
    uint32_t rank = 0;
    for (int j = 0; j < n; j++) {
        const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
        rank += kj < sk ? 1u : 0u;
    }
    if (lane < n) k[rank] = key;
}

// maximum of a value < 512 over the lanes with `active` set (wave-uniform result): nine ballots, no shuffles
__device__ __forceinline__ uint32_t wave_max9(uint32_t v, bool active)
{
    unsigned long long cand = __ballot(active);
    uint32_t mx = 0;
    if (!cand) return 0;
#pragma unroll
    for (int b = 8; b >= 0; b--) {
        const unsigned long long m = __ballot(active && ((v >> b) & 1u)) & cand;
        if (m) { cand = m; mx |= 1u << b; }
    }
    return mx;
}

// Seed merging of one read by the whole wavefront (same rules as merge_read).  k[0..n) are the read's SORTED keys in LDS.  Sixty-four
// keys at a time: every lane finds where the group that would start at its key ends (the greedy rule of DV-DPfunctions.cpp:2514 --
// while next.offset <= first.offset + indelFuzz on the same strand), a scalar walk from group start to group start marks the lanes
// that really start one, and those lanes union their group's intervals.  Time per read ~ its largest group, not its number of hits.
// Reads with more than 64 keys are swept twice (the short-seed filter needs the best covered length of the whole read first).
__device__ __forceinline__ uint32_t merge_read_wave(const uint64_t *k, int n, mp_seedpos *__restrict__ out, uint32_t r, const mp_dev_mmp &P, uint64_t text_len, int cw32,
                                                    uint32_t *__restrict__ n_pos, uint32_t *__restrict__ n_neg, int lane)
{
    const int n_chunks = (n + 63) >> 6;
    uint32_t max_cov = 0, w_out = 0, np = 0, nn = 0;
    for (int sweep = n_chunks > 1 ? 0 : 1; sweep < 2; sweep++) {
        int s = 0;                                        // next group start (wave-uniform)
        for (int c = 0; c < n_chunks; c++) {
            const int cbase = c << 6, e = cbase + lane;
            const bool valid = e < n;
            const uint64_t key = valid ? k[e] : 0;
            const uint64_t pos = key_pos(key, text_len), group_end = pos + (uint64_t)P.indel_fuzz;
            int nxt = e + 1;
            if (valid)
                while (nxt < n) {
                    const uint64_t cur = k[nxt];
                    if (MP_KEY_STRAND(cur) != MP_KEY_STRAND(key) || key_pos(cur, text_len) > group_end) break;
                    nxt++;
                }
            unsigned long long starts = 0;
            const int cend = cbase + 64 < n ? cbase + 64 : n;
            while (s < cend) { const int ls = s - cbase; starts |= 1ull << ls; s = __builtin_amdgcn_readlane(nxt, ls); }
            const bool is_start = (starts >> lane) & 1ull;
            uint32_t total = 0;
            bool has_unique = false;
            if (is_start && nxt == e + 1) { total = (uint32_t)MP_KEY_LEN(key); has_unique = MP_KEY_UNIQ(key); }      // a group of one: its own length
            else if (is_start) {
                uint32_t cov[MP_SEED_BIG / 32];
#pragma unroll
                for (int w = 0; w < MP_SEED_BIG / 32; w++) cov[w] = 0;
                for (int j = e; j < nxt; j++) {
                    const uint64_t cur = j == e ? key : k[j];
                    const int lo = MP_KEY_QO(cur), hi = lo + MP_KEY_LEN(cur);
                    has_unique |= MP_KEY_UNIQ(cur);
#pragma unroll
                    for (int w = 0; w < MP_SEED_BIG / 32; w++) {
                        if (w >= cw32) break;                  // reads of this batch are no longer than 32 cw32 bases
                        int a = lo - 32 * w, b = hi - 32 * w;
                        a = a < 0 ? 0 : (a > 32 ? 32 : a);
                        b = b < 0 ? 0 : (b > 32 ? 32 : b);
                        const uint32_t below_b = b >= 32 ? ~0u : ((1u << b) - 1u), below_a = a >= 32 ? ~0u : ((1u << a) - 1u);
                        cov[w] |= below_b & ~below_a;
                    }
                }
#pragma unroll
                for (int w = 0; w < MP_SEED_BIG / 32; w++) total += (uint32_t)__popc(cov[w]);
            }
            if (n_chunks == 1 || sweep == 0) { const uint32_t m = wave_max9(total, is_start); max_cov = m > max_cov ? m : max_cov; }
            if (sweep == 1) {
                // kept: a unique or long enough group (:2533) that also passes the short-seed filter against the best covered length of
                // BOTH strands, rejected groups included (:2547)
                const bool keep = is_start && (has_unique || total >= (uint32_t)P.good_seed_len) && (double)total >= P.short_seed_ratio * (double)max_cov;
                const unsigned long long kb = __ballot(keep);
                const uint32_t strand = MP_KEY_STRAND(key);
                if (keep) {
                    mp_seedpos sp;
                    sp.pos = pos;
                    sp.paired_seed_length = total;
                    sp.strand_read_id = (r & ~1u) | (strand << 31);
                    out[w_out + (uint32_t)__popcll(kb & ((1ull << lane) - 1ull))] = sp;
                }
                const uint32_t neg = (uint32_t)__popcll(__ballot(keep && strand));
                w_out += (uint32_t)__popcll(kb); nn += neg; np += (uint32_t)__popcll(kb) - neg;
            }
        }
    }
    if (lane == 0) { n_pos[r] = np; n_neg[r] = nn; }
    return w_out;
}

// ---- four reads per wavefront: a read with at most 16 keys (nearly all of them) is handled by one 16-lane ROW, the four rows of a wavefront
// working on four reads at once.  Same rules as merge_read_wave; instead of v_readlane broadcasts (wave-wide, one read at a time) every lane
// meets the other fifteen keys of its row through DPP row rotations (full-rate VALU, no LDS): ranks for the sort, group ends, the row's
// best covered length.  Rows whose read has more keys (or none in range) sit the call out (active = false) and are handled wave-wide afterwards.
template <int N> __device__ __forceinline__ uint32_t mp_row_ror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x120 + N, 0xF, 0xF, false); }
template <int N> __device__ __forceinline__ uint64_t mp_row_ror64(uint64_t v) { return ((uint64_t)mp_row_ror<N>((uint32_t)(v >> 32)) << 32) | mp_row_ror<N>((uint32_t)v); }
template <int N, class F> __device__ __forceinline__ void mp_row_others(uint64_t x, F &&f) { f(mp_row_ror64<N>(x)); if constexpr (N < 15) mp_row_others<N + 1>(x, f); }

__device__ __forceinline__ uint32_t merge_reads_rows(bool active, uint64_t *ks /* the row's read: its keys in LDS */, int n /* <= 16 */, mp_seedpos *__restrict__ out, uint32_t r,
                                                     const mp_dev_mmp &P, uint64_t text_len, int cw32, uint32_t *__restrict__ n_pos, uint32_t *__restrict__ n_neg, int lane)
{
    const int sl = lane & 15;
    const bool valid = active && sl < n;
    // sort: rank among the row's keys (made distinct by the lane number), scatter to LDS, read back in order
    const uint64_t key0 = valid ? ks[sl] : ~0ull;
    const uint64_t sk = valid ? (key0 << 6) | (uint64_t)sl : ~0ull;
    uint32_t rank = 0;
    mp_row_others<1>(sk, [&](uint64_t o) { rank += o < sk ? 1u : 0u; });
    if (valid) ks[rank] = key0;
    const uint64_t key = valid ? ks[sl] : 0;                            // (LDS operations of one wavefront complete in order)
    // where the group that would start at this key ends: the keys behind it on the same strand within indelFuzz (DV-DPfunctions.cpp:2514)
    const uint64_t pos = key_pos(key, text_len), group_end = pos + (uint64_t)P.indel_fuzz;
    const uint64_t cs = valid ? (key << 4) | (uint64_t)sl : ~0ull;      // keys use 53 bits
    uint32_t behind = 0;
    mp_row_others<1>(cs, [&](uint64_t o) {
        const uint64_t kj = o >> 4;
        const int j = (int)(o & 15u);
        behind += (o != ~0ull && j > sl && MP_KEY_STRAND(kj) == MP_KEY_STRAND(key) && key_pos(kj, text_len) <= group_end) ? 1u : 0u;
    });
    const uint32_t nxt = (uint32_t)sl + 1u + behind;
    // the groups of a read: from start to start (each row walks its own chain; the rows of a wavefront in step)
    uint32_t s = 0, starts = 0;
    for (;;) {
        const bool go = active && s < (uint32_t)n;
        if (!__any(go)) break;
        if (go) starts |= 1u << s;
        const uint32_t nx = (uint32_t)__shfl((int)nxt, (lane & 48) | (int)(go ? s : 0u));
        if (go) s = nx;
    }
    const bool is_start = valid && ((starts >> sl) & 1u);
    uint32_t total = 0;
    bool has_unique = false;
    if (is_start && nxt == (uint32_t)sl + 1u) { total = (uint32_t)MP_KEY_LEN(key); has_unique = MP_KEY_UNIQ(key); }      // a group of one: its own length
    else if (is_start) {
        uint32_t cov[MP_SEED_BIG / 32];
#pragma unroll
        for (int w = 0; w < MP_SEED_BIG / 32; w++) cov[w] = 0;
        for (uint32_t j = (uint32_t)sl; j < nxt; j++) {
            const uint64_t cur = j == (uint32_t)sl ? key : ks[j];
            const int lo = MP_KEY_QO(cur), hi = lo + MP_KEY_LEN(cur);
            has_unique |= MP_KEY_UNIQ(cur);
#pragma unroll
            for (int w = 0; w < MP_SEED_BIG / 32; w++) {
                if (w >= cw32) break;                  // reads of this batch are no longer than 32 cw32 bases
                int a = lo - 32 * w, b = hi - 32 * w;
                a = a < 0 ? 0 : (a > 32 ? 32 : a);
                b = b < 0 ? 0 : (b > 32 ? 32 : b);
                const uint32_t below_b = b >= 32 ? ~0u : ((1u << b) - 1u), below_a = a >= 32 ? ~0u : ((1u << a) - 1u);
                cov[w] |= below_b & ~below_a;
            }
        }
#pragma unroll
        for (int w = 0; w < MP_SEED_BIG / 32; w++) total += (uint32_t)__popc(cov[w]);
    }
    // best covered length of the read, rejected groups included (:2547): maximum over the row
    uint32_t max_cov = is_start ? total : 0u;
    { uint32_t o = mp_row_ror<8>(max_cov); max_cov = o > max_cov ? o : max_cov; o = mp_row_ror<4>(max_cov); max_cov = o > max_cov ? o : max_cov;
      o = mp_row_ror<2>(max_cov); max_cov = o > max_cov ? o : max_cov; o = mp_row_ror<1>(max_cov); max_cov = o > max_cov ? o : max_cov; }
    const bool keep = is_start && (has_unique || total >= (uint32_t)P.good_seed_len) && (double)total >= P.short_seed_ratio * (double)max_cov;
    const unsigned long long kb = __ballot(keep);
    const uint32_t strand = MP_KEY_STRAND(key);
    const unsigned long long nb = __ballot(keep && strand);
    const uint32_t row_keep = (uint32_t)(kb >> (lane & 48)) & 0xFFFFu, row_neg = (uint32_t)(nb >> (lane & 48)) & 0xFFFFu;
    if (keep) {
        mp_seedpos sp;
        sp.pos = pos;
        sp.paired_seed_length = total;
        sp.strand_read_id = (r & ~1u) | (strand << 31);
        out[__popc(row_keep & ((1u << sl) - 1u))] = sp;
    }
    if (active && sl == 0) { const uint32_t nn = (uint32_t)__popc(row_neg); n_pos[r] = (uint32_t)__popc(row_keep) - nn; n_neg[r] = nn; }
    return (uint32_t)__popcll(kb);
}

// Output: seedpos[read_base[r] .. ) holds the kept seeds of read r, '+' first; n_pos[r], n_neg[r] their counts.
// One lane per read, one wavefront per block.  The hits of a block's 64 reads are one contiguous range: it is read with
// coalesced loads, packed to 8-byte keys and staged in LDS.  Sorting: a lane-per-read sort makes the wavefront wait for
// its largest read (nine in ten wavefronts hold a read with more than 64 hits, while the median read has 20), so the
// reads are sorted one after the other by all 64 lanes together: a bitonic network over registers and wavefront
// shuffles, 1 to 8 keys per lane.  The grouping pass is linear and runs one lane per read.  A block whose range does not
// fit in LDS packs and sorts in place in global memory, lane by lane.
// The seeds of one (read, strand) as sort keys, one per SA row (what expand_hits_kernel + locate_kernel + hit_key produce), for an index
// whose whole suffix array is resident: a seed reported by its text position is one key with no look-up, a row is one load.
__device__ __forceinline__ uint32_t mp_rs_keys(const mp_dev_index &ix, const mp_dev_mmp &P, uint32_t rs, const mp_seed_rec *__restrict__ slabs, const uint32_t *__restrict__ seeds_per_rs,
                                               const uint32_t *__restrict__ over_slot, const mp_seed_rec *__restrict__ big_slabs, uint32_t slab_limit,
                                               const uint32_t *__restrict__ lens, uint64_t *dst)
{
    const uint32_t ns = seeds_per_rs[rs];
    if (ns == 0) return 0;
    const mp_seed_rec *src = slabs + (uint64_t)rs * MP_SEED_SLAB;
    if (ns > slab_limit) src = big_slabs + (uint64_t)over_slot[rs] * MP_SEED_BIG;
    const uint32_t len = lens[rs >> 1], strand = rs & 1u;
    uint32_t w = 0;
    for (uint32_t j0 = 0; j0 < ns; j0 += 4) {
        // four records per round: their loads (and the suffix-array reads of single-row seeds, the common kind that was not already
        // reported by its position) are in flight together
        mp_seed_rec r4[4];
        uint32_t sa4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (j0 + u < ns) r4[u] = src[j0 + u];
#pragma unroll
        for (int u = 0; u < 4; u++) sa4[u] = (j0 + u < ns && r4[u].sa_diff != MP_SEED_LOCATED && r4[u].sa_l != 0) ? ix.sa_dense[r4[u].sa_l] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j0 + u >= ns) break;
            const mp_seed_rec rec = r4[u];
            const bool located = rec.sa_diff == MP_SEED_LOCATED;
            const uint32_t rows = located ? 1u : rec.sa_diff + 1u;
            const uint32_t off = rec.query_offset, seedlen = rec.seed_len;
            mp_seed_hit h;
            h.length = (uint16_t)seedlen; h.qo = (uint16_t)off; h.strand = (uint16_t)strand;
            h.mult = (seedlen >= (uint32_t)P.good_seed_len || seedlen >= len / 2 || located) ? (uint16_t)1 : (uint16_t)(rec.sa_diff + 1);
            // SA[0] is kept as -1 (BWT.c:242,994-996); unsigned 64-bit arithmetic: the projection may wrap exactly as in the reference
            auto emit = [&](uint64_t sav) {
                h.offset = strand == 0 ? sav - off : sav - ((uint64_t)len - seedlen - off);
                dst[w++] = hit_key(h, P);
            };
            if (located) emit((uint64_t)rec.sa_l);
            else {
                emit(rec.sa_l == 0 ? ~0ull : (uint64_t)sa4[u]);
                // further rows of a seed with several occurrences: consecutive suffix-array entries, eight loads in flight at a time
                for (uint32_t k0 = 1; k0 < rows; k0 += 8) {
                    uint32_t v8[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) v8[q] = k0 + q < rows ? ix.sa_dense[rec.sa_l + k0 + q] : 0u;
#pragma unroll
                    for (int q = 0; q < 8; q++) if (k0 + q < rows) emit((uint64_t)v8[q]);
                }
            }
        }
    }
    return ns;
}

// FUSED (whole suffix array resident): the keys are made here from the seeding kernel's records -- no expand / locate launch, no hit
// array; `hits` is then only scratch for the blocks that do not fit in LDS (8 bytes per hit).
template <bool FUSED>
__global__ void __launch_bounds__(MP_MERGE_THREADS)
merge_kernel(mp_seed_hit *__restrict__ hits, const uint32_t *__restrict__ rs_base, uint32_t n_reads, mp_dev_mmp P, uint64_t text_len, int cov_words,
             mp_seedpos *__restrict__ seedpos, uint32_t *__restrict__ n_pos, uint32_t *__restrict__ n_neg, mp_counters *ctr,
             mp_dev_index ix, const mp_seed_rec *__restrict__ slabs, const uint32_t *__restrict__ seeds_per_rs, const uint32_t *__restrict__ over_slot,
             const mp_seed_rec *__restrict__ big_slabs, uint32_t slab_limit, const uint32_t *__restrict__ lens,
             uint32_t stage_cap /* <= MP_MERGE_CAP; smaller only to exercise the path of blocks that do not fit in LDS */,
             int rows_on /* 0: every read by a whole wavefront (test switch MP_MERGE_ROWS=0) */)
{
    __shared__ uint64_t s_key[MP_MERGE_CAP];
    __shared__ uint32_t s_base[MP_MERGE_BLOCK + 1];
    __shared__ uint32_t s_part[2 + MP_MERGE_THREADS / MP_WAVE];     // seeds turned into keys by wavefronts 0, 1; seeds kept by each wavefront
    static_assert(MP_MERGE_THREADS / MP_WAVE == 4 && 2 * MP_MERGE_BLOCK == 2 * MP_WAVE, "s_part is summed by hand");
    const uint32_t r0 = blockIdx.x * MP_MERGE_BLOCK;
    const uint32_t r1 = min(r0 + MP_MERGE_BLOCK, n_reads);
    const uint32_t b0 = rs_base[2 * r0], T = rs_base[2 * r1] - b0;
    const bool staged = T <= stage_cap;
    const int lane = threadIdx.x & (MP_WAVE - 1), wave = threadIdx.x / MP_WAVE;
    if (!staged) {
        // a block whose hits do not fit in LDS: one lane per read, in global memory (not FUSED: in place, key i overwrites bytes
        // 8i..8i+7 of the read's own records, all of them already consumed)
        const uint32_t r = r0 + threadIdx.x;
        if (threadIdx.x >= MP_MERGE_BLOCK || r >= n_reads) return;
        const uint32_t base = rs_base[2 * r];
        const int n = (int)(rs_base[2 * r + 2] - base);
        uint64_t *k;
        if (FUSED) {
            k = reinterpret_cast<uint64_t *>(hits) + base;
            uint32_t nsd = mp_rs_keys(ix, P, 2 * r, slabs, seeds_per_rs, over_slot, big_slabs, slab_limit, lens, k);
            nsd += mp_rs_keys(ix, P, 2 * r + 1, slabs, seeds_per_rs, over_slot, big_slabs, slab_limit, lens, k + (rs_base[2 * r + 1] - base));
            if (nsd) atomicAdd(&ctr->n_seeds, (unsigned long long)nsd);
        } else {
            k = reinterpret_cast<uint64_t *>(hits + base);
            for (int i = 0; i < n; i++) { const mp_seed_hit h = hits[base + i]; k[i] = hit_key(h, P); }
        }
        sort_keys(k, n);
        merge_read(k, n, seedpos + base, r, P, text_len, cov_words, n_pos, n_neg, ctr);
        return;
    }
    if (threadIdx.x <= MP_MERGE_BLOCK) { const uint32_t r = min(r0 + threadIdx.x, r1); s_base[threadIdx.x] = rs_base[2 * r] - b0; }
    if (FUSED) {
        // one lane per (read, strand) writes its keys into the block's LDS range
        uint32_t nsd = 0;
        const uint32_t rs = 2 * r0 + threadIdx.x;
        if (threadIdx.x < 2 * MP_MERGE_BLOCK && rs < 2 * r1)
            nsd = mp_rs_keys(ix, P, rs, slabs, seeds_per_rs, over_slot, big_slabs, slab_limit, lens, s_key + (rs_base[rs] - b0));
        if (threadIdx.x < 2 * MP_MERGE_BLOCK) {
            for (int o = MP_WAVE / 2; o > 0; o >>= 1) nsd += __shfl_down(nsd, o);
            if (lane == 0) s_part[wave] = nsd;                  // summed per block below: one counter update per block, not per wavefront
        }
    } else {
        // four independent 16-byte loads in flight per lane
        for (uint32_t i0 = 0; i0 < T; i0 += 4 * MP_MERGE_THREADS) {
            mp_seed_hit h4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * MP_MERGE_THREADS + threadIdx.x; if (i < T) h4[u] = hits[b0 + i]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * MP_MERGE_THREADS + threadIdx.x; if (i < T) s_key[i] = hit_key(h4[u], P); }
        }
    }
    __syncthreads();
    // ---- four reads at a time per wavefront, one 16-lane row each (merge_reads_rows); a read with more than 16 keys is then taken by the whole
    // wavefront: sort (lane = key), then groups (lane = group).  The block's wavefronts take every fourth quad of reads.
    const int cw32 = 2 * cov_words;
    uint32_t wave_kept = 0;
    static_assert(MP_MERGE_BLOCK % 4 == 0, "reads are handed out in quads");
    for (uint32_t q0 = 4u * (uint32_t)wave; r0 + q0 < r1; q0 += 4u * (MP_MERGE_THREADS / MP_WAVE)) {
        const uint32_t mine = q0 + (uint32_t)(lane >> 4);                       // this row's read
        const bool in = r0 + mine < r1;
        const uint32_t kbr = s_base[in ? mine : q0];
        const int nsr = in ? (int)(s_base[mine + 1] - kbr) : 0;
        const bool rows = rows_on != 0;
        wave_kept += merge_reads_rows(in && rows && nsr <= 16, s_key + kbr, nsr, seedpos + b0 + kbr, r0 + mine, P, text_len, cw32, n_pos, n_neg, lane);
        for (uint32_t src = q0; src < q0 + 4u && r0 + src < r1; src++) {
            const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_base[src]);          // wave-uniform: the loops below are scalar loops
            const int ns = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)s_base[src + 1]) - kb);
            if (rows && ns <= 16) continue;                                                               // done by its row
            if (ns == 0) { if (lane == 0) { n_pos[r0 + src] = 0; n_neg[r0 + src] = 0; } continue; }
            uint64_t *ks = s_key + kb;
            if (ns > 1) {
                if (ns <= 64) wave_rank_sort(ks, ns, lane);
                else if (ns <= 128) wave_sort_read<2>(ks, ns, lane);
                else if (ns <= 256) wave_sort_read<4>(ks, ns, lane);
                else if (ns <= MP_MERGE_COOP) wave_sort_read<8>(ks, ns, lane);
                else { if (lane == 0) sort_keys(ks, ns); }                 // the rare very large read (heap sort by one lane)
            }
            wave_kept += merge_read_wave(ks, ns, seedpos + b0 + kb, r0 + src, P, text_len, cw32, n_pos, n_neg, lane);
        }
    }
    // one update per block and counter: a single address takes only ~100 M atomics per second chip-wide, and a launch over 2 Mi reads has
    // 131 k wavefronts here -- per-wavefront updates of the two counters were most of this kernel's time
    if (lane == 0) s_part[2 + wave] = wave_kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t kept = s_part[2] + s_part[3] + s_part[4] + s_part[5], nsd = FUSED ? s_part[0] + s_part[1] : 0u;
        if (kept) atomicAdd(&ctr->n_seedpos, (unsigned long long)kept);
        if (nsd) atomicAdd(&ctr->n_seeds, (unsigned long long)nsd);
    }
}

// ------------------------------------------------------------------------------------------------ K4
__device__ __forceinline__ int mp_margin(int l) { return l > 100 ? 30 : 25; }

// Emits (EMIT) or only counts the candidates of one pair: pairEndMerge for both legs.
template <bool EMIT>
__device__ __forceinline__ uint32_t pair_candidates(const mp_seedpos *__restrict__ seedpos, const uint32_t *__restrict__ rs_base,
                                                    const uint32_t *__restrict__ n_pos, const uint32_t *__restrict__ n_neg,
                                                    const uint32_t *__restrict__ lens, uint32_t p, int insert_low, int insert_high,
                                                    mp_candidate *__restrict__ out)
{
    uint32_t n = 0;
    for (int leg = 0; leg < 2; leg++) {
        // leg 0: read '+' with mate '-' (readIDLeft even); leg 1: mate '+' with read '-' (readIDLeft odd)
        const uint32_t lft = 2 * p + (uint32_t)leg, rgt = lft ^ 1u;
        const mp_seedpos *L = seedpos + rs_base[2 * lft];
        const int nl = (int)n_pos[lft];
        const mp_seedpos *R = seedpos + rs_base[2 * rgt] + n_pos[rgt];
        const int nr = (int)n_neg[rgt];
        if (nl == 0 || nr == 0) continue;
        const int rlen = (int)lens[rgt];                     // length of the right-leg read (DV-DPfunctions.cpp:2030)
        const int margin = mp_margin(rlen);
        int length_low = insert_low - rlen - margin;
        if (length_low < 0) length_low = 0;
        const int length_high = insert_high - rlen + margin;
        int pre_start = 0;
        uint64_t prev_kept = 0;
        for (int a = 0; a < nl; a++) {
            const uint64_t read_loc = L[a].pos;
            // MC_Compress on the fly: skip left seeds within DP2_DIVIDE_GAP (5) of the previously kept one
            if (a > 0 && !(prev_kept + 5 < read_loc)) continue;
            prev_kept = read_loc;
            for (int b = pre_start; b < nr; b++) {
                const uint64_t mate_loc = R[b].pos;
                if (read_loc + (uint64_t)(int64_t)length_high < mate_loc) break;
                else if (read_loc + (uint64_t)(int64_t)length_low <= mate_loc) {
                    if (EMIT) { mp_candidate c; c.read_id_left = lft; c.pad = 0; c.pos0 = read_loc; c.pos1 = mate_loc; out[n] = c; }
                    n++;
                    pre_start = b;                           // App. D7: overwritten at every match
                }
            }
        }
    }
    return n;
}

// One lane per pair.  Count, reserve a contiguous range with ONE atomic per wavefront (a single counter word sustains
// only ~90 returning atomics per microsecond), then emit.
__global__ void __launch_bounds__(128)
pair_kernel(const mp_seedpos *__restrict__ seedpos, const uint32_t *__restrict__ rs_base, const uint32_t *__restrict__ n_pos,
            const uint32_t *__restrict__ n_neg, const uint32_t *__restrict__ lens, uint32_t n_pairs, int insert_low, int insert_high,
            mp_candidate *__restrict__ cands, uint32_t cand_cap, mp_counters *ctr)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (MP_WAVE - 1);
    const uint32_t mine = p < n_pairs ? pair_candidates<false>(seedpos, rs_base, n_pos, n_neg, lens, p, insert_low, insert_high, nullptr) : 0u;
    // inclusive prefix sum over the wavefront
    uint32_t incl = mine;
    for (int o = 1; o < MP_WAVE; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const uint32_t total = __shfl(incl, MP_WAVE - 1);
    uint32_t base = 0;
    if (lane == MP_WAVE - 1 && total) base = (uint32_t)atomicAdd(&ctr->n_cand, (unsigned long long)total);
    base = __shfl(base, MP_WAVE - 1);
    if (mine == 0) return;
    const uint32_t at = base + incl - mine;
    if (at + mine <= cand_cap) pair_candidates<true>(seedpos, rs_base, n_pos, n_neg, lens, p, insert_low, insert_high, cands + at);
    else atomicAdd(&ctr->overflow, 1ull);
}
