/*
 * oracle/seeding.c -- CPU ORACLE (test infrastructure only): MMP seeding, SA locate, seed merge/filter.
 *
 * Restates, for one read at a time:
 *   mmp<0> / mmp<2>            soap4/DV-DPfunctions.cpp:2224-2267, 2317-2377
 *   CHECK_AND_SET_LAST         :2188-2195
 *   CHECK_AND_ADD_RANGE        :2197-2219
 *   SA locate loop             :2470-2495
 *   merge / cover / filter     :2499-2553
 * Reads are passed as one base code (0..3) per byte, 5'->3'.
 */
#include "mp_oracle.h"
#include <stdlib.h>
#include <string.h>

#define LKT_K 13

typedef struct { uint64_t offset; uint32_t mult; int32_t length; uint32_t qo; } seed_align;

static int cmp_seed_align(const void *a, const void *b)
{
    uint64_t x = ((const seed_align *)a)->offset, y = ((const seed_align *)b)->offset;
    return x < y ? -1 : x > y;
}

static int cmp_itv(const void *a, const void *b)
{
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : x[1] > y[1];
}

int mpo_mmp(const mpo_index *ix, const uint8_t *read, int len, int strand, const mpo_mmp_params *P,
            mpo_sa_seed *out, int cap, mpo_seed_counters *ctr)
{
    /* `sym(k)` is the k-th base in scan order: the forward strand is scanned right-to-left, the reverse
     * strand left-to-right with complemented bases (DV-DPfunctions.cpp:2252-2253 vs :2361-2362). */
#define SYM(k) (strand == 0 ? read[len - 1 - (k)] : (uint8_t)(3 - read[(k)]))
    const uint64_t n = mpo_text_length(ix);
    int i, seed_len = 0, n_out = 0;
    uint64_t l = 0, r = n, nextl, nextr;
    uint64_t last_l = l, last_r = r, last_seed_len = 0;

#define CLOSE_SEED(x_expr) do {                                                                          \
        int diff = 0;                                                                                    \
        int x = (x_expr);                                                                                \
        if (seed_len >= P->seed_min_length) {                                                            \
            if (seed_len >= P->reseed_len && last_r - last_l + 1 <= (uint64_t)P->sa_size_threshold &&    \
                ((uint64_t)seed_len - last_seed_len <= (uint64_t)P->reseed_abs_diff ||                   \
                 seed_len * P->reseed_rlt_ratio < (double)last_seed_len)) {                              \
                diff = seed_len - (int)last_seed_len;                                                    \
                l = last_l; r = last_r; seed_len = (int)last_seed_len;                                   \
            }                                                                                            \
            if (n_out < cap) {                                                                           \
                out[n_out].query_offset = (uint32_t)x & 1023u;      /* query_offset : 10 */              \
                out[n_out].sa_l = l;                                                                     \
                out[n_out].sa_diff = (uint32_t)((r - l) < (uint64_t)P->sa_size_threshold ? (r - l) : (uint64_t)P->sa_size_threshold); \
                out[n_out].seed_len = (uint32_t)seed_len;                                                \
            }                                                                                            \
            n_out++;                                                                                     \
            if (ctr) ctr->n_seeds++;                                                                     \
        }                                                                                                \
        i -= diff;                                                                                       \
        i -= seed_len < P->seed_min_length ? seed_len : P->seed_min_length;                              \
        l = 0; r = n; seed_len = 0;                                                                      \
        last_l = l; last_r = r; last_seed_len = 0;                                                       \
    } while (0)

    for (i = 0; i < len; ++i) {
        if (seed_len == 0) {
            if (len - i < P->seed_min_length) break;
            /* 13-mer key: scan position i lands in the low bits, i+12 (the leftmost text base) in bits 25:24 */
            uint32_t key = 0;
            for (int k = 0; k < LKT_K; k++) key |= (uint32_t)SYM(i + k) << (2 * k);
            mpo_lkt_range(ix, key, &nextl, &nextr);
            if (ctr) ctr->n_lkt++;
            i += LKT_K - 1;
            seed_len = LKT_K - 1;
        } else {
            unsigned c = SYM(i);
            nextl = mpo_cum_freq(ix, c) + mpo_occ(ix, l, c) + 1;
            nextr = mpo_cum_freq(ix, c) + mpo_occ(ix, r + 1, c);
            if (ctr) ctr->n_occ += 2;
        }
        if (nextl <= nextr) {
            if (seed_len >= P->seed_min_length && nextr - nextl < r - l) {
                last_r = r; last_l = l; last_seed_len = (uint64_t)seed_len;
            }
            l = nextl; r = nextr; ++seed_len;
        } else {
            if (strand == 0) CLOSE_SEED(len - i); else CLOSE_SEED(i - seed_len);
        }
    }
    if (strand == 0) CLOSE_SEED(0); else CLOSE_SEED(len - seed_len);
#undef CLOSE_SEED
#undef SYM
    return n_out;
}

void mpo_seed_read(const mpo_index *ix, const uint8_t *read, int len, uint32_t even_read_id,
                   const mpo_mmp_params *P, mpo_seed_pos *out_pos, int *n_pos, mpo_seed_pos *out_neg, int *n_neg,
                   int cap, mpo_seed_counters *ctr)
{
    mpo_sa_seed *seeds = (mpo_sa_seed *)malloc(sizeof(mpo_sa_seed) * (size_t)(len + 2));
    mpo_seed_pos *kept = (mpo_seed_pos *)malloc(sizeof(mpo_seed_pos) * (size_t)(len + 2) * 64);
    seed_align *sa[2];
    int n_sa[2] = {0, 0}, n_kept = 0, a;
    uint32_t max_seed_len = 0;
    *n_pos = 0; *n_neg = 0;

    for (a = 0; a < 2; a++) {
        int ns = mpo_mmp(ix, read, len, a, P, seeds, len + 2, ctr);
        sa[a] = (seed_align *)malloc(sizeof(seed_align) * (size_t)(ns * (P->sa_size_threshold + 1) + 1));
        for (int j = 0; j < ns; j++) {
            uint64_t l = seeds[j].sa_l, r = seeds[j].sa_l + seeds[j].sa_diff;
            if (r > l + (uint64_t)P->sa_size_threshold) r = (uint64_t)P->sa_size_threshold + l - 1;
            uint32_t off = seeds[j].query_offset, seedlen = seeds[j].seed_len;
            for (uint64_t k = l; k <= r; k++) {
                uint64_t steps = 0, idx = k, si = mpo_sa_interval(ix);
                while (idx % si) { idx = mpo_psi_minus(ix, idx); steps++; }
                if (ctr) { ctr->n_occ += steps; ctr->n_sa++; ctr->n_hits++; }
                uint64_t sav = mpo_sa_value(ix, k);
                uint64_t target = a == 0 ? sav - off : sav - ((uint64_t)(uint32_t)len - seedlen - off);
                seed_align *s = &sa[a][n_sa[a]++];
                s->offset = target;
                s->length = (int16_t)seedlen;
                s->qo = off & 0xFFFF;
                if (seedlen >= (uint32_t)P->good_seed_len || seedlen >= (uint32_t)len / 2) s->mult = 1;
                else s->mult = (uint32_t)(r - l + 1) & 0xFFFF;
            }
        }
        qsort(sa[a], (size_t)n_sa[a], sizeof(seed_align), cmp_seed_align);
    }

    uint32_t (*itv)[2] = (uint32_t (*)[2])malloc(sizeof(uint32_t[2]) * (size_t)(n_sa[0] + n_sa[1] + 1));
    for (a = 0; a < 2; a++) {
        for (int m = 0; m < n_sa[a]; m++) {
            seed_align *s = &sa[a][m];
            int has_unique = s->mult <= (uint32_t)P->uniq_threshold && s->length >= P->seed_min_length;
            uint64_t pos = s->offset;
            int ni = 0;
            itv[ni][0] = s->qo; itv[ni][1] = (uint32_t)(s->length + (int)s->qo); ni++;
            while (m + 1 < n_sa[a] && sa[a][m + 1].offset <= pos + (uint64_t)P->indel_fuzz) {
                ++m; s = &sa[a][m];
                has_unique |= s->mult <= (uint32_t)P->uniq_threshold && s->length >= P->seed_min_length;
                itv[ni][0] = s->qo; itv[ni][1] = (uint32_t)(s->length + (int)s->qo); ni++;
            }
            qsort(itv, (size_t)ni, sizeof(uint32_t[2]), cmp_itv);
            uint32_t total = 0, cs = 0, ce = 0;
            for (int q = 0; q < ni; q++) {
                if (itv[q][0] >= ce) { total += ce - cs; cs = itv[q][0]; }
                if (itv[q][1] > ce) ce = itv[q][1];
            }
            total += ce - cs;
            if (max_seed_len < total) max_seed_len = total;
            if (has_unique || total >= (uint32_t)P->good_seed_len) {
                kept[n_kept].pos = pos;
                kept[n_kept].paired_seed_length = total;
                kept[n_kept].strand_read_id = even_read_id | ((uint32_t)a << 31);
                n_kept++;
            }
        }
    }
    for (int j = 0; j < n_kept; j++) {
        if ((double)kept[j].paired_seed_length >= P->short_seed_ratio * (double)max_seed_len) {
            if (kept[j].strand_read_id >> 31) { if (*n_neg < cap) out_neg[*n_neg] = kept[j]; (*n_neg)++; }
            else { if (*n_pos < cap) out_pos[*n_pos] = kept[j]; (*n_pos)++; }
        }
    }
    free(itv); free(sa[0]); free(sa[1]); free(seeds); free(kept);
}
