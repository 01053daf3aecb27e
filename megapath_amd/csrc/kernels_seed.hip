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
// sector per rank query, enough waves in flight to cover ~1 us of HBM latency per step, and no host round trips.
#include <hip/hip_runtime.h>
#include "mp_internal.h"

#define MP_WAVE 64

// ------------------------------------------------------------------------------------------------ FM-index primitives
__device__ __forceinline__ uint32_t mp_popc_sym(uint32_t w, uint32_t pat, int nsym)
{
    uint32_t x = w ^ pat;
    uint32_t m = ~(x | (x >> 1)) & 0x55555555u;
    if (nsym < 16) m &= (1u << (2 * nsym)) - 1u;
    return (uint32_t)__popc(m);
}

// rank_c over BWT[0,k) with the '$' row removed from the numbering (BWTOccValue, BWT.c:597-634)
__device__ __forceinline__ uint32_t mp_occ_raw(const mp_dev_index &ix, uint32_t k, uint32_t c)
{
    uint32_t b = k / MP_OCC_SYMS, off = k - b * MP_OCC_SYMS;
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.occ + b);
    uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    uint32_t cnt = c == 0 ? q0.x : c == 1 ? q0.y : c == 2 ? q0.z : q0.w;
    const uint32_t pat = c * 0x55555555u;
    const uint32_t w[12] = {q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
    for (int i = 0; i < 12; i++) {
        int ns = (int)off - 16 * i;
        if (ns > 0) cnt += mp_popc_sym(w[i], pat, ns);
    }
    return cnt;
}

__device__ __forceinline__ uint32_t mp_occ(const mp_dev_index &ix, uint32_t k, uint32_t c)
{
    k -= (k > ix.inv_sa0);
    return mp_occ_raw(ix, k, c);
}

// LF step (BWTPsiMinusValue, BWT.c:915-938): one 64-byte block gives both the symbol and its rank
__device__ __forceinline__ uint32_t mp_lf(const mp_dev_index &ix, uint32_t idx)
{
    if (idx == ix.inv_sa0) return 0;
    uint32_t k = idx + 1;
    k -= (k > ix.inv_sa0);
    uint32_t s = k - 1;
    uint32_t b = s / MP_OCC_SYMS, so = s - b * MP_OCC_SYMS;
    uint32_t c = (ix.occ[b].bwt[so >> 4] >> (2 * (so & 15))) & 3u;
    return ix.cum[c] + mp_occ_raw(ix, k, c);
}

__global__ void occ_query_kernel(mp_dev_index ix, const uint64_t *idx, const uint32_t *sym, uint64_t n, uint64_t *out)
{
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t < n) out[t] = mp_occ(ix, (uint32_t)idx[t], sym[t]);
}

__device__ __forceinline__ uint64_t mp_sa_value(const mp_dev_index &ix, uint32_t idx, uint32_t *steps_out)
{
    uint32_t steps = 0;
    while (idx % ix.sa_interval != 0) { idx = mp_lf(ix, idx); steps++; }
    uint32_t q = idx / ix.sa_interval;
    *steps_out = steps;
    // SA[0] is kept as -1 so that walking through the '$' row yields text position `steps - 1` (BWT.c:242,994-996)
    uint64_t base = q == 0 ? ~0ull : (uint64_t)ix.sa[q];
    return base + steps;
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
        out_l[t] = k ? (uint64_t)ix.lkt[k - 1] + 1 : 1;
        out_r[t] = ix.lkt[k];
    }
}

// ------------------------------------------------------------------------------------------------ scan-order reads
// scan[(read*2+strand)*W2 + w]: the read's bases in the order the strand's backward search consumes them
// (strand 0: right-to-left; strand 1: left-to-right, complemented), LSB-first, zero padded.  W2 = W + 1 words so that a
// 26-bit key can always be cut from two adjacent words.
__global__ void prep_scan_kernel(const uint32_t *reads, const uint32_t *lens, uint32_t n_reads, uint32_t W, uint32_t *scan)
{
    const uint32_t W2 = W + 1;
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)n_reads * 2 * W2;
    if (t >= total) return;
    uint32_t w = (uint32_t)(t % W2);
    uint32_t rs = (uint32_t)(t / W2);
    uint32_t r = rs >> 1, strand = rs & 1;
    uint32_t len = lens[r];
    const uint32_t *src = reads + (uint64_t)r * W;
    uint32_t out = 0;
#pragma unroll 4
    for (uint32_t j = 0; j < 16; j++) {
        uint32_t k = w * 16 + j;
        if (k < len) {
            uint32_t p = strand == 0 ? len - 1 - k : k;
            uint32_t c = (src[p >> 4] >> (2 * (p & 15))) & 3u;
            if (strand) c = 3u - c;
            out |= c << (2 * j);
        }
    }
    scan[t] = out;
}

// ------------------------------------------------------------------------------------------------ K1
__global__ void __launch_bounds__(256)
seed_kernel(mp_dev_index ix, const uint32_t *__restrict__ scan, const uint32_t *__restrict__ lens, uint32_t n_reads,
            uint32_t W, mp_dev_mmp P, mp_sa_seed *__restrict__ seeds, uint32_t seed_cap, mp_hit_task *__restrict__ hit_tasks,
            uint32_t hit_cap, uint32_t *__restrict__ hits_per_read, mp_counters *ctr)
{
    const uint32_t rs = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t my_occ = 0, my_lkt = 0;
    if (rs < n_reads * 2) {
        const uint32_t W2 = W + 1;
        const uint32_t *sc = scan + (uint64_t)rs * W2;
        const int len = (int)lens[rs >> 1];
        const int strand = (int)(rs & 1);
        const uint32_t n = ix.n;
        int i, seed_len = 0;
        uint32_t l = 0, r = n, nextl = 0, nextr = 0;
        uint32_t last_l = 0, last_r = n, last_seed_len = 0;
        uint32_t cur_w = 0xFFFFFFFFu, cur_word = 0;

        auto close_seed = [&](int x) {
            int diff = 0;
            if (seed_len >= P.seed_min_length) {
                // reseed to the last shorter seed whose SA range was larger (DV-DPfunctions.cpp:2200-2205); all
                // comparisons are unsigned as in the reference (int vs unsigned long long operands)
                if (seed_len >= P.reseed_len && (uint64_t)last_r - last_l + 1 <= (uint64_t)P.sa_size_threshold &&
                    ((uint64_t)(uint32_t)seed_len - last_seed_len <= (uint64_t)P.reseed_abs_diff ||
                     seed_len * P.reseed_rlt_ratio < (double)last_seed_len)) {
                    diff = seed_len - (int)last_seed_len;
                    l = last_l; r = last_r; seed_len = (int)last_seed_len;
                }
                uint32_t sa_diff = (r - l) < (uint32_t)P.sa_size_threshold ? (r - l) : (uint32_t)P.sa_size_threshold;
                uint32_t nh = sa_diff + 1;
                if ((uint32_t)P.sa_size_threshold + 1 < nh) nh = (uint32_t)P.sa_size_threshold + 1;
                uint32_t slot = (uint32_t)atomicAdd(&ctr->n_seeds, 1ull);
                uint32_t hbase = (uint32_t)atomicAdd(&ctr->n_hits, (unsigned long long)nh);
                if (slot < seed_cap && hbase + nh <= hit_cap) {
                    mp_sa_seed s;
                    s.read_strand = rs; s.sa_l = l; s.query_offset = (uint16_t)(x & 1023); s.seed_len = (uint16_t)seed_len; s.sa_diff = sa_diff;
                    seeds[slot] = s;
                    for (uint32_t j = 0; j < nh; j++) { mp_hit_task t; t.seed_slot = slot; t.k = l + j; hit_tasks[hbase + j] = t; }
                    atomicAdd(&hits_per_read[rs >> 1], nh);
                } else {
                    atomicAdd(&ctr->overflow, 1ull);
                }
            }
            i -= diff;
            i -= seed_len < P.seed_min_length ? seed_len : P.seed_min_length;
            l = 0; r = n; seed_len = 0;
            last_l = 0; last_r = n; last_seed_len = 0;
        };

        for (i = 0; i < len; ++i) {
            if (seed_len == 0) {
                if (len - i < P.seed_min_length) break;
                // 13-mer key: scan position i in the low bits, i+12 (the leftmost text base) in bits 25:24
                uint32_t wq = (uint32_t)i >> 4, sh = 2 * ((uint32_t)i & 15);
                uint64_t two = ((uint64_t)sc[wq + 1] << 32) | sc[wq];
                uint32_t key = (uint32_t)(two >> sh) & (MP_LKT_ENTRIES - 1);
                nextl = key ? ix.lkt[key - 1] + 1 : 1;
                nextr = ix.lkt[key];
                my_lkt++;
                i += MP_LKT_K - 1;
                seed_len = MP_LKT_K - 1;
            } else {
                uint32_t wq = (uint32_t)i >> 4;
                if (wq != cur_w) { cur_w = wq; cur_word = sc[wq]; }
                uint32_t c = (cur_word >> (2 * ((uint32_t)i & 15))) & 3u;
                nextl = ix.cum[c] + mp_occ(ix, l, c) + 1;
                nextr = ix.cum[c] + mp_occ(ix, r + 1, c);
                my_occ += 2;
            }
            if (nextl <= nextr) {
                if (seed_len >= P.seed_min_length && nextr - nextl < r - l) { last_r = r; last_l = l; last_seed_len = (uint32_t)seed_len; }
                l = nextl; r = nextr; ++seed_len;
            } else {
                close_seed(strand == 0 ? len - i : i - seed_len);
            }
        }
        close_seed(strand == 0 ? 0 : len - seed_len);
    }
    // one counter update per wave
    for (int o = MP_WAVE / 2; o > 0; o >>= 1) { my_occ += __shfl_down(my_occ, o); my_lkt += __shfl_down(my_lkt, o); }
    if ((threadIdx.x & (MP_WAVE - 1)) == 0) {
        if (my_occ) atomicAdd(&ctr->n_occ, (unsigned long long)my_occ);
        if (my_lkt) atomicAdd(&ctr->n_lkt, (unsigned long long)my_lkt);
    }
}

// ------------------------------------------------------------------------------------------------ K2
__global__ void __launch_bounds__(256)
locate_kernel(mp_dev_index ix, const mp_sa_seed *__restrict__ seeds, const mp_hit_task *__restrict__ tasks, uint32_t n_tasks,
              const uint32_t *__restrict__ lens, mp_dev_mmp P, const uint32_t *__restrict__ read_base, uint32_t *__restrict__ cursor,
              mp_seed_hit *__restrict__ hits, mp_counters *ctr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t my_occ = 0;
    if (t < n_tasks) {
        mp_hit_task task = tasks[t];
        mp_sa_seed s = seeds[task.seed_slot];
        uint32_t read = s.read_strand >> 1, strand = s.read_strand & 1;
        uint32_t len = lens[read];
        uint32_t steps;
        uint64_t sav = mp_sa_value(ix, task.k, &steps);
        my_occ = steps;
        uint32_t off = s.query_offset, seedlen = s.seed_len;
        // unsigned 64-bit arithmetic: the projection may wrap exactly as in the reference (DV-DPfunctions.cpp:2489)
        uint64_t target = strand == 0 ? sav - off : sav - ((uint64_t)len - seedlen - off);
        mp_seed_hit h;
        h.offset = target;
        h.length = (uint16_t)seedlen;
        h.qo = (uint16_t)off;
        h.strand = (uint16_t)strand;
        h.mult = (seedlen >= (uint32_t)P.good_seed_len || seedlen >= len / 2) ? (uint16_t)1 : (uint16_t)(s.sa_diff + 1);
        uint32_t slot = read_base[read] + atomicAdd(&cursor[read], 1u);
        hits[slot] = h;
    }
    for (int o = MP_WAVE / 2; o > 0; o >>= 1) my_occ += __shfl_down(my_occ, o);
    if ((threadIdx.x & (MP_WAVE - 1)) == 0 && my_occ) atomicAdd(&ctr->n_occ, (unsigned long long)my_occ);
    if (t == 0) atomicAdd(&ctr->n_sa, (unsigned long long)n_tasks);
}

// ------------------------------------------------------------------------------------------------ K3
__device__ __forceinline__ bool hit_less(const mp_seed_hit &a, const mp_seed_hit &b)
{
    if (a.strand != b.strand) return a.strand < b.strand;
    return a.offset < b.offset;
}

__device__ void sort_hits(mp_seed_hit *a, int n)
{
    if (n <= 24) {
        for (int i = 1; i < n; i++) {
            mp_seed_hit t = a[i];
            int j = i - 1;
            while (j >= 0 && hit_less(t, a[j])) { a[j + 1] = a[j]; j--; }
            a[j + 1] = t;
        }
        return;
    }
    // heap sort for the rare large (repeat-rich) segments
    for (int start = n / 2 - 1; start >= 0; start--) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && hit_less(a[child], a[child + 1])) child++;
            if (!hit_less(a[root], a[child])) break;
            mp_seed_hit t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    }
    for (int end = n - 1; end > 0; end--) {
        mp_seed_hit t = a[0]; a[0] = a[end]; a[end] = t;
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && hit_less(a[child], a[child + 1])) child++;
            if (!hit_less(a[root], a[child])) break;
            mp_seed_hit u = a[root]; a[root] = a[child]; a[child] = u;
            root = child;
        }
    }
}

// Output: seedpos[read_base[r] .. ) holds the kept seeds of read r, '+' first; n_pos[r], n_neg[r] their counts.
__global__ void __launch_bounds__(128)
merge_kernel(mp_seed_hit *__restrict__ hits, const uint32_t *__restrict__ read_base, uint32_t n_reads, mp_dev_mmp P,
             mp_seedpos *__restrict__ seedpos, uint32_t *__restrict__ n_pos, uint32_t *__restrict__ n_neg, mp_counters *ctr)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t base = read_base[r];
    const int n = (int)(read_base[r + 1] - base);
    mp_seed_hit *h = hits + base;
    mp_seedpos *out = seedpos + base;
    sort_hits(h, n);
    uint32_t max_cov = 0;
    int n_kept = 0;
    for (int m = 0; m < n; m++) {
        const uint16_t strand = h[m].strand;
        const uint64_t pos = h[m].offset;
        bool has_unique = false;
        int g0 = m;
        // group while next.offset <= first.offset + indelFuzz on the same strand (DV-DPfunctions.cpp:2514)
        while (m + 1 < n && h[m + 1].strand == strand && h[m + 1].offset <= pos + (uint64_t)P.indel_fuzz) m++;
        // covered read length = union of [qo, qo+len) over the group: intervals taken in ascending (qo, end) order
        uint32_t total = 0, cs = 0, ce = 0;
        uint32_t prev_lo = 0, prev_hi = 0;
        bool first = true;
        for (int cnt = g0; cnt <= m; cnt++) {
            // select the next interval in (qo, end) order without allocating: groups are tiny (<= 31 * few seeds)
            uint32_t best_lo = 0xFFFFFFFFu, best_hi = 0xFFFFFFFFu;
            int same = 0;
            for (int q = g0; q <= m; q++) {
                uint32_t lo = h[q].qo, hi = (uint32_t)h[q].length + h[q].qo;
                bool after_prev = first || lo > prev_lo || (lo == prev_lo && hi > prev_hi);
                if (!after_prev) continue;
                if (lo < best_lo || (lo == best_lo && hi < best_hi)) { best_lo = lo; best_hi = hi; }
            }
            if (best_lo == 0xFFFFFFFFu) break;           // remaining intervals are duplicates of ones already taken
            (void)same;
            if (best_lo >= ce) { total += ce - cs; cs = best_lo; }
            if (best_hi > ce) ce = best_hi;
            prev_lo = best_lo; prev_hi = best_hi; first = false;
        }
        total += ce - cs;
        for (int q = g0; q <= m; q++)
            has_unique |= (int)h[q].mult <= P.uniq_threshold && (int)h[q].length >= P.seed_min_length;
        if (max_cov < total) max_cov = total;
        if (has_unique || total >= (uint32_t)P.good_seed_len) {
            mp_seedpos sp;
            sp.pos = pos;
            sp.paired_seed_length = total;
            sp.strand_read_id = (r & ~1u) | ((uint32_t)strand << 31);
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
    if (w) atomicAdd(&ctr->n_seedpos, (unsigned long long)w);
}

// ------------------------------------------------------------------------------------------------ K4
__device__ __forceinline__ int mp_margin(int l) { return l > 100 ? 30 : 25; }

__global__ void __launch_bounds__(128)
pair_kernel(const mp_seedpos *__restrict__ seedpos, const uint32_t *__restrict__ read_base, const uint32_t *__restrict__ n_pos,
            const uint32_t *__restrict__ n_neg, const uint32_t *__restrict__ lens, uint32_t n_pairs, int insert_low, int insert_high,
            mp_candidate *__restrict__ cands, uint32_t cand_cap, mp_counters *ctr)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    for (int leg = 0; leg < 2; leg++) {
        // leg 0: read '+' with mate '-' (readIDLeft even); leg 1: mate '+' with read '-' (readIDLeft odd)
        const uint32_t lft = 2 * p + (uint32_t)leg, rgt = lft ^ 1u;
        const mp_seedpos *L = seedpos + read_base[lft];
        const int nl = (int)n_pos[lft];
        const mp_seedpos *R = seedpos + read_base[rgt] + n_pos[rgt];
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
                    uint32_t slot = (uint32_t)atomicAdd(&ctr->n_cand, 1ull);
                    if (slot < cand_cap) {
                        mp_candidate c; c.read_id_left = lft; c.pad = 0; c.pos0 = read_loc; c.pos1 = mate_loc;
                        cands[slot] = c;
                    } else atomicAdd(&ctr->overflow, 1ull);
                    pre_start = b;                           // App. D7: overwritten at every match
                }
            }
        }
    }
}
