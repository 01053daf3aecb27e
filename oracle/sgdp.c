/*
 * oracle/sgdp.c -- CPU ORACLE (test infrastructure only): semi-global affine-gap DP + traceback.
 *
 * Restates callDP -> SemiGlobalAlignment -> GenerateDPTable -> GPUBacktrack
 * (reference soap4/CPU_DP.cpp:122-619 score fill, :622-786 traceback, :788-871 driver) on plain integer
 * H/E/F matrices, with the traceback state machine kept in the reference's own
 * shape (NORMAL / D_EXT / I_EXT and the three soft-clip exits) so that tie-breaking is identical.
 * The reference's u8 delta encoding and score trimming are implementation devices that never reach
 * the outputs (App. C.3/C.4b); tests/test_oracle_dp.py fuzzes this file against the real callDP.
 *
 * Scores are fixed by the reference kernel: match +1, gap extend -1 (CPU_DP.cpp:199-200);
 * mismatch and gap open come from the ini (-2 / -3, soap4-nt2.ini:39-45).
 */
#include "mp_oracle.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#define MISMATCH (-2)
#define GAP_OPEN (-3)
#define GAP_EXT (-1)
#define NEG_INF (-30000)

int mpo_dp_cutoff(int read_len)
{
    /* std::max(DP_SCORE_THRESHOLD_RATIO * readLength, DP_SCORE_THRESHOLD_LOWER_BOUND) truncated to int
     * (definitions.h:166-167; DV-DPfunctions.cpp:2916) */
    double v = 0.2 * (double)read_len;
    if (v < 30.0) v = 30.0;
    return (int)v;
}

void mpo_sgdp(const uint8_t *ref, int m, const uint8_t *read, int L,
              int clip_l, int clip_r, int cutoff, mpo_dp_result *res, uint8_t *ops, int ops_cap)
{
    const int W = L + 1;
    short *H = (short *)malloc(sizeof(short) * (size_t)(m + 1) * W);
    uint8_t *flag = (uint8_t *)malloc((size_t)(m + 1) * W); /* 0 = soft-clipped (Z), 1 = E==H, 2 = otherwise */
    int *Ecol = (int *)malloc(sizeof(int) * W);             /* E[i-1][j] while filling row i */
    int i, j;
    res->score = 0; res->hit_loc = 0; res->max_score_count = 0; res->n_ops = 0;

    /* row 0 (CPU_DP.cpp:394-432): free start inside the left clip, then a pure insertion run */
    H[0] = 0; flag[0] = 0;
    for (j = 1; j <= L; j++) {
        if (j <= clip_l) { H[j] = 0; flag[j] = 0; }
        else { H[j] = (short)(GAP_OPEN + (j - clip_l - 1) * GAP_EXT); flag[j] = 2; }
        Ecol[j] = NEG_INF;
    }
    Ecol[0] = NEG_INF;

    int best = 0, best_i = 0, best_j = 0, count = 0;
    int min_col = L - clip_r; if (min_col < 0) min_col = 0;

    for (i = 1; i <= m; i++) {
        short *Hp = H + (size_t)(i - 1) * W, *Hc = H + (size_t)i * W;
        uint8_t *fc = flag + (size_t)i * W;
        int F = NEG_INF;
        Hc[0] = 0; fc[0] = 0;                 /* column 0: free reference prefix (CPU_DP.cpp:441-452) */
        for (j = 1; j <= L; j++) {
            int e = Hp[j] + GAP_OPEN;
            if (Ecol[j] + GAP_EXT > e) e = Ecol[j] + GAP_EXT;
            int f = Hc[j - 1] + GAP_OPEN;
            if (F + GAP_EXT > f) f = F + GAP_EXT;
            int h = Hp[j - 1] + (ref[i - 1] == read[j - 1] ? 1 : MISMATCH);
            if (e > h) h = e;
            if (f > h) h = f;
            int z = 0;
            if (j <= clip_l && h < 0) { h = 0; z = 1; }   /* CPU_DP.cpp:505-510 */
            Hc[j] = (short)h;
            fc[j] = z ? 0 : (e == h ? 1 : 2);
            Ecol[j] = e; F = f;
            /* best cell: rows <= refLen, columns >= L - clipR, strict '>' in row-major order (CPU_DP.cpp:538-591) */
            if (j >= min_col && h >= cutoff) {
                if (h > best) { best = h; best_i = i; best_j = j; count = 1; }
                else if (h == best && count < 255) count++;
            }
        }
    }

    if (best >= cutoff && best > 0) {
        int valid = 1, n = 0, hit = 0;
#define EMIT(ch) do { if (n < ops_cap) ops[n] = (uint8_t)(ch); n++; } while (0)
        if (L < cutoff) valid = 0;                          /* CPU_DP.cpp:830-831 */
        else {
            int c = best_j, r = best_i, k;
            for (k = 0; k < L - best_j; k++) EMIT('S');
            enum { NORMAL, I_EXT, D_EXT } state = NORMAL;
            int accum = 0, done = 0;
            while (c > 0 && r > 0 && !done) {
                int here = H[(size_t)r * W + c];
                int diag_diff = here - H[(size_t)(r - 1) * W + c - 1];
                int hori_diff = here - H[(size_t)r * W + c - 1];
                int vert_diff = here - H[(size_t)(r - 1) * W + c];
                int s = ref[r - 1] == read[c - 1] ? 1 : MISMATCH;
                int fl = flag[(size_t)r * W + c];
                if (state == NORMAL) {
                    if (flag[(size_t)(r - 1) * W + c - 1] == 0 && diag_diff == s && c != 1) {
                        /* DP_BT_SM_EXIT: match/mismatch, then clip the remaining c-1 read bases */
                        EMIT(s == 1 ? 'M' : 'm');
                        for (k = 0; k < c - 1; k++) EMIT('S');
                        hit = r - 1; done = 1;
                    } else if (diag_diff == s) {
                        EMIT(s == 1 ? 'M' : 'm'); r--; c--;
                    } else if (fl == 1) {
                        EMIT('D'); r--;
                        if (vert_diff != GAP_OPEN) { accum = vert_diff - GAP_EXT; state = D_EXT; }
                    } else {
                        EMIT('I'); c--;
                        if (hori_diff != GAP_OPEN) { accum = hori_diff - GAP_EXT; state = I_EXT; }
                    }
                } else if (state == D_EXT) {
                    if (flag[(size_t)(r - 1) * W + c] == 0 && vert_diff + accum == GAP_OPEN) {
                        valid = 0; done = 1;                /* DP_BT_SD_EXIT: starts with a deletion -> dropped */
                    } else {
                        EMIT('D'); r--;
                        if (vert_diff + accum == GAP_OPEN) state = NORMAL;
                        else accum += vert_diff - GAP_EXT;
                    }
                } else {
                    if (flag[(size_t)r * W + c - 1] == 0 && hori_diff + accum == GAP_OPEN) {
                        /* DP_BT_SI_EXIT */
                        EMIT('I');
                        for (k = 0; k < c - 1; k++) EMIT('S');
                        hit = r; done = 1;
                    } else {
                        EMIT('I'); c--;
                        if (hori_diff + accum == GAP_OPEN) state = NORMAL;
                        else accum += hori_diff - GAP_EXT;
                    }
                }
            }
            if (!done) {
                if (r == 0) {
                    int sc = clip_l > c ? c : clip_l;       /* CPU_DP.cpp:743-757 */
                    for (k = 0; k < c - sc; k++) EMIT('I');
                    for (k = 0; k < sc; k++) EMIT('S');
                }
                hit = r;
            }
        }
#undef EMIT
        if (valid) { res->score = best; res->hit_loc = hit; res->max_score_count = count; res->n_ops = n; }
        else { res->score = 0; res->hit_loc = 0; res->max_score_count = count; res->n_ops = 0; }
    }
    free(H); free(flag); free(Ecol);
}

int mpo_cigar_from_ops(const uint8_t *ops, int n_ops, char *out, int cap)
{
    /* CigarStringEncoder (DV-DPfunctions.h:377-427): runs are collected in traceback order and printed reversed */
    int len = 0, i = n_ops - 1;
    out[0] = 0;
    while (i >= 0) {
        int j = i;
        while (j >= 0 && ops[j] == ops[i]) j--;
        int w = snprintf(out + len, (size_t)(cap - len), "%d%c", i - j, ops[i]);
        if (w < 0 || w >= cap - len) return -1;
        len += w;
        i = j;
    }
    return len;
}
