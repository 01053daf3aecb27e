/*
 * oracle/pipeline.c -- CPU ORACLE (test infrastructure only): the whole `soap4 pair ... -F` path.
 *
 * Restates, scalar and single-threaded (SURVEY.md section 3.2 call stack):
 *   read ingest / packing quirks       soap4/QueryParser.cpp:184-234, IndexHandler.cpp:26-46
 *   insert_low adjustment              soap4/SOAP4.cpp:465-474, QueryParser.cpp:2253-2277
 *   deep-DP round                      DV-DPfunctions.cpp:2617-2690 (seeding), :1968-2119 (pairing),
 *                                      :2857-3005 (windows), :3391-3540 (result assembly)
 *   per-pair result selection          DV-DPfunctions.h:165-243, DV-DPfunctions.cpp:240-259, OutputDPResult.cpp:65-265
 *   annotated FASTQ                    BGS-IO.cpp:1312-1446, :1943-2091
 *   single-end rescue                  DV-DPForSingleReads.cpp:121-215, DV-DPfunctions.cpp:295-342, :401-445, :678-750
 *   mate rescue ("default DP")         DV-SemiDP.cpp:30-296, DV-DPfunctions.cpp:1048-1231, :1476-1747
 *   unpaired output                    SeedPool.cpp:267-319, DV-DPfunctions.cpp:841-909, PEAlgnmt.cpp:1119-1269,
 *                                      OutputDPResult.cpp:284-404, BGS-IO.cpp:1384-1446
 */
#include "mp_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NOT_ALIGNED 0xFFFFFFFFFFFFFFFFull
#define DIVIDE_GAP 5
#define MARGIN(l) ((l) > 100 ? 30 : 25)

/* ------------------------------------------------------------------ small containers */
typedef struct { void *p; size_t n, cap, esz; } vec;
static void vec_init(vec *v, size_t esz) { v->p = NULL; v->n = v->cap = 0; v->esz = esz; }
static void *vec_push(vec *v)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = realloc(v->p, v->cap * v->esz);
        if (!v->p) { fprintf(stderr, "[oracle] out of memory\n"); exit(1); }
    }
    return (char *)v->p + (v->n++) * v->esz;
}
static void vec_free(vec *v) { free(v->p); v->p = NULL; v->n = v->cap = 0; }

/* ------------------------------------------------------------------ reads */
typedef struct {
    int n;                 /* number of reads (2 per pair) */
    char **name, **comment, **qual;
    uint8_t **seq;         /* base codes */
    int *len;
} read_set;

static unsigned char g_char_map[256];
static void fill_char_map(void)
{
    /* INDEXFillCharMap (IndexHandler.cpp:26-46): everything maps to A except ACGT/acgt, U->T, N->G */
    memset(g_char_map, 0, sizeof g_char_map);
    const char *b = "ACGT";
    for (int i = 0; i < 4; i++) { g_char_map[(int)b[i]] = (unsigned char)i; g_char_map[(int)b[i] - 'A' + 'a'] = (unsigned char)i; }
    g_char_map['U'] = g_char_map['u'] = 3;
    g_char_map['N'] = g_char_map['n'] = 2;
}

static char *read_line(FILE *f, char **buf, size_t *cap)
{
    size_t n = 0; int c;
    while ((c = fgetc(f)) != EOF && c != '\n') {
        if (n + 2 > *cap) { *cap = *cap ? *cap * 2 : 256; *buf = (char *)realloc(*buf, *cap); }
        (*buf)[n++] = (char)c;
    }
    if (c == EOF && n == 0) return NULL;
    if (!*buf) { *cap = 16; *buf = (char *)malloc(16); }
    if (n && (*buf)[n - 1] == '\r') n--;
    (*buf)[n] = 0;
    return *buf;
}

static int load_fastq_pair(const char *fq1, const char *fq2, int max_read_length, int qconst, read_set *rs)
{
    FILE *f[2] = { fopen(fq1, "rb"), fopen(fq2, "rb") };
    if (!f[0] || !f[1]) { fprintf(stderr, "[oracle] cannot open read files\n"); return -1; }
    vec nm, cm, ql, sq, ln;
    vec_init(&nm, sizeof(char *)); vec_init(&cm, sizeof(char *)); vec_init(&ql, sizeof(char *));
    vec_init(&sq, sizeof(uint8_t *)); vec_init(&ln, sizeof(int));
    char *buf = NULL; size_t cap = 0;
    (void)qconst;
    for (;;) {
        int got = 0;
        for (int k = 0; k < 2; k++) {
            if (!read_line(f[k], &buf, &cap)) continue;
            if (buf[0] != '@' && buf[0] != '>') { k--; continue; }
            int fasta = buf[0] == '>';
            /* kseq: name = up to first whitespace, comment = rest (kseq.h); trim_readno (QueryParser.cpp:177) */
            char *sp = buf + 1; while (*sp && *sp != ' ' && *sp != '\t') sp++;
            char *comment = NULL;
            if (*sp) { *sp = 0; comment = sp + 1; if (!*comment) comment = NULL; }
            size_t L = strlen(buf + 1);
            if (L > 2 && buf[1 + L - 2] == '/' && buf[1 + L - 1] >= '0' && buf[1 + L - 1] <= '9') buf[1 + L - 2] = 0;
            *(char **)vec_push(&nm) = strdup(buf + 1);
            *(char **)vec_push(&cm) = comment ? strdup(comment) : NULL;
            if (!read_line(f[k], &buf, &cap)) { fprintf(stderr, "[oracle] truncated read file\n"); return -1; }
            int slen = (int)strlen(buf);
            int keep = slen > max_read_length - 1 ? max_read_length - 1 : slen;   /* QueryParser.cpp:188 */
            uint8_t *codes = (uint8_t *)malloc((size_t)keep + 1);
            for (int i = 0; i < keep; i++) codes[i] = g_char_map[(unsigned char)buf[i]];
            *(uint8_t **)vec_push(&sq) = codes;
            *(int *)vec_push(&ln) = keep;
            char *q = (char *)malloc((size_t)keep + 1);
            if (!fasta) {
                read_line(f[k], &buf, &cap);               /* '+' line */
                read_line(f[k], &buf, &cap);
                for (int i = 0; i < keep; i++) q[i] = buf[i];
            } else {
                for (int i = 0; i < keep; i++) q[i] = (char)qconst; /* no qualities: zeros + constant */
            }
            q[keep] = 0;
            *(char **)vec_push(&ql) = q;
            got++;
        }
        if (got == 0) break;
        if (got != 2) { fprintf(stderr, "Error: number of sequences of pair-end files not matched.\n"); return -1; }
    }
    free(buf); fclose(f[0]); fclose(f[1]);
    rs->n = (int)nm.n; rs->name = (char **)nm.p; rs->comment = (char **)cm.p; rs->qual = (char **)ql.p;
    rs->seq = (uint8_t **)sq.p; rs->len = (int *)ln.p;
    return 0;
}

static void free_reads(read_set *rs)
{
    for (int i = 0; i < rs->n; i++) { free(rs->name[i]); free(rs->comment[i]); free(rs->qual[i]); free(rs->seq[i]); }
    free(rs->name); free(rs->comment); free(rs->qual); free(rs->seq); free(rs->len);
}

/* GetReadLength (QueryParser.cpp:2253-2277): max over every `sample`-th length, first 10^6 samples */
static int detect_read_length(const int *lens, int n, int sample)
{
    if (n <= 0) return 100;
    int i = 0, j = 1, mx = lens[0];
    while (i < n && j < 1000000) { if (mx < lens[i]) mx = lens[i]; j++; i += sample; }
    return mx;
}

/* ------------------------------------------------------------------ DP on a text window */
typedef struct {
    int score, hit_loc, count;
    char *cigar;           /* special CIGAR, malloc'ed, or NULL */
} dp_out;

static void run_dp(const mpo_index *ix, const read_set *rs, int read_id, int strand /*1 fwd, 2 rev-comp*/,
                   uint64_t win_start, int win_len, const mpo_align_params *P, dp_out *o, mpo_align_stats *st)
{
    int L = rs->len[read_id];
    uint8_t *ref = (uint8_t *)malloc((size_t)(win_len > 0 ? win_len : 1));
    uint8_t *rd = (uint8_t *)malloc((size_t)L + 1);
    uint8_t *ops = (uint8_t *)malloc((size_t)(L + win_len + 8));
    for (int i = 0; i < win_len; i++) ref[i] = (uint8_t)mpo_text_base(ix, win_start + (uint64_t)i);
    if (strand == 1) memcpy(rd, rs->seq[read_id], (size_t)L);
    else for (int i = 0; i < L; i++) rd[i] = (uint8_t)(3 - rs->seq[read_id][L - 1 - i]);
    /* soft clips follow the strand (DV-DPfunctions.cpp:2908-2911) */
    int cl = strand == 1 ? P->clip_left : P->clip_right;
    int cr = strand == 1 ? P->clip_right : P->clip_left;
    mpo_dp_result r;
    mpo_sgdp(ref, win_len, rd, L, cl, cr, mpo_dp_cutoff(L), &r, ops, L + win_len + 8);
    o->score = r.score; o->hit_loc = r.hit_loc; o->count = r.max_score_count; o->cigar = NULL;
    if (r.score > 0) {
        o->cigar = (char *)malloc((size_t)(L + win_len) * 4 + 16);
        mpo_cigar_from_ops(ops, r.n_ops, o->cigar, (L + win_len) * 4 + 16);
    }
    if (st) { st->n_dp_instances++; st->dp_cells += (uint64_t)(win_len + 1) * (uint64_t)L; }
    free(ref); free(rd); free(ops);
}

/* ------------------------------------------------------------------ deep-DP result record (PEAlgnmt.h:509-541) */
typedef struct {
    uint32_t read_id;              /* even id of the pair */
    uint64_t algnmt_1, algnmt_2;
    int strand_1, strand_2, score_1, score_2;
    char *cigar_1, *cigar_2;
    uint64_t start_1, start_2;     /* window starts (dpInfoForReads) */
    int dplen_1, dplen_2;
    int editdist_1, editdist_2;    /* XA edit distance */
    int count_1, count_2;          /* num_sameScore */
    long long insert;              /* insertSize: only its sign reaches the BAM record */
} deep_result;

/* edit distance from the score identity (DV-DPfunctions.cpp:733-737, 1545-1548, 3466-3469); `len` is the read length the
 * reference happens to use there (the right leg's for both mates of a deep-DP pair).  Also returns D - I - S. */
static int edit_dist(const char *sp, int len, int score, int *dis)
{
    int nI = 0, nD = 0, nS = 0, gap = 0, cur = 0;
    for (const char *p = sp; *p; p++) {
        if (*p >= '0' && *p <= '9') { cur = cur * 10 + (*p - '0'); continue; }
        if (cur > 0) {
            if (*p == 'I') { nI += cur; gap += -3 + (cur - 1) * -1; }
            else if (*p == 'D') { nD += cur; gap += -3 + (cur - 1) * -1; }
            else if (*p == 'S') nS += cur;
        }
        cur = 0;
    }
    if (dis) *dis = nD - nI - nS;
    int L = len - nI - nS;
    return nI + nD + (L * 1 + gap - score) / (1 - -2);
}

static int cmp_deep(const void *a, const void *b)
{
    /* ResultCompare<DeepDPAlignResult> (DV-DPfunctions.cpp:254-257) */
    const deep_result *x = (const deep_result *)a, *y = (const deep_result *)b;
    if (x->algnmt_1 != y->algnmt_1) return x->algnmt_1 < y->algnmt_1 ? -1 : 1;
    if (x->algnmt_2 != y->algnmt_2) return x->algnmt_2 < y->algnmt_2 ? -1 : 1;
    if (x->score_1 != y->score_1) return x->score_1 < y->score_1 ? -1 : 1;
    if (x->score_2 != y->score_2) return x->score_2 < y->score_2 ? -1 : 1;
    return 0;
}

/* decideTargetChr (BGS-IO.cpp:1312-1341): sequence id if [pos, pos+readLen-1] lies inside one sequence, else -1 */
static int target_chr(const mpo_index *ix, uint64_t pos, int read_len)
{
    uint64_t tp; unsigned c1, c2;
    mpo_chr_and_pos(ix, pos, &tp, &c1);
    mpo_chr_and_pos(ix, pos + (uint64_t)read_len - 1, &tp, &c2);
    return c1 == c2 ? (int)c1 : -1;
}

typedef struct { int chr, neg_score; } chr_hit;
static int cmp_chr_hit(const void *a, const void *b)
{
    const chr_hit *x = (const chr_hit *)a, *y = (const chr_hit *)b;
    if (x->chr != y->chr) return x->chr < y->chr ? -1 : 1;
    return x->neg_score < y->neg_score ? -1 : x->neg_score > y->neg_score;
}

typedef struct { int score, s, e; } hdr_map;
/* getMappingFromHeader (BGS-IO.cpp:1348-1371), quirks included (the record filter tests the header's best score) */
static int mapping_from_header(const char *comment, hdr_map *v, int *nv, int cap, double top, double scoreT)
{
    *nv = 0;
    if (comment == NULL || strcmp(comment, "IGNORE") == 0) return 0;
    int score = atoi(comment + 6);
    if (score < scoreT) return score;
    else if (scoreT < score * top) scoreT = score * top;
    const char *p = strchr(comment + 6, ';');
    while (p && *(p + 1) != '\0') {
        hdr_map m;
        m.score = atoi(p + 1);
        m.s = (int)(p + 1 - comment);
        p = strchr(p + 1, ';');
        if (!p) break;
        m.e = (int)(p - comment);
        if (score >= scoreT && *nv < cap) v[(*nv)++] = m;
    }
    return score;
}

static void put_seq_qual(FILE *out, const read_set *rs, int id)
{
    for (int i = 0; i < rs->len[id]; i++) fputc("ACGT"[rs->seq[id][i]], out);
    fputs("\n+\n", out);
    /* qualities were stored minus the constant and are printed plus the constant (BGS-IO.cpp:1373-1383) */
    fwrite(rs->qual[id], 1, (size_t)rs->len[id], out);
    fputc('\n', out);
}

static void put_score_line(FILE *out, const mpo_index *ix, const read_set *rs, int id, int best, chr_hit *hits, int nh,
                           const hdr_map *v, int nv, double top)
{
    const char *comment = rs->comment[id];
    fprintf(out, "@%s", rs->name[id]);
    if (comment && strcmp(comment, "IGNORE") == 0) { fputs("\tIGNORE\n", out); return; }
    fprintf(out, "\tSCORE:%d;", best);
    if (best > 0) {
        for (int i = 0; i < nh; i++) {
            if (i > 0 && hits[i].chr == hits[i - 1].chr) continue;
            if (-hits[i].neg_score > 0 && -hits[i].neg_score >= best * top)
                fprintf(out, "%d,%s;", -hits[i].neg_score, mpo_seq_name(ix, hits[i].chr - 1));
        }
    }
    for (int i = 0; i < nv; i++)
        if (v[i].score >= best * top) { fwrite(comment + v[i].s, 1, (size_t)(v[i].e - v[i].s), out); fputc(';', out); }
    fputc('\n', out);
}

/* pairDeepDPOutputFastqAPI (BGS-IO.cpp:1966-2091) for the results of one pair */
static void output_pair_fastq(FILE *out, const mpo_index *ix, const read_set *rs, const mpo_align_params *P,
                              deep_result *res, int n)
{
    int id1 = (int)res[0].read_id, id2 = id1 + 1;
    int best1 = 0, best2 = 0, n1 = 0, n2 = 0;
    chr_hit *h1 = (chr_hit *)malloc(sizeof(chr_hit) * (size_t)(n + 1)), *h2 = (chr_hit *)malloc(sizeof(chr_hit) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) {
        int chr1 = -1, chr2 = -1;
        if (res[i].algnmt_1 != NOT_ALIGNED) chr1 = target_chr(ix, res[i].algnmt_1, rs->len[id1]);
        if (res[i].algnmt_2 != NOT_ALIGNED) chr2 = target_chr(ix, res[i].algnmt_2, rs->len[id2]);
        if (chr1 == -1) { res[i].algnmt_1 = NOT_ALIGNED; res[i].score_1 = 0; }
        if (chr2 == -1) { res[i].algnmt_2 = NOT_ALIGNED; res[i].score_2 = 0; }
        if (P->megapath_mode == 2 && (res[i].algnmt_1 == NOT_ALIGNED || res[i].algnmt_2 == NOT_ALIGNED)) {
            res[i].algnmt_1 = NOT_ALIGNED; res[i].score_1 = 0; res[i].algnmt_2 = NOT_ALIGNED; res[i].score_2 = 0;
        }
        /* normalizeScore (BGS-IO.cpp:1943-1961) */
        if (res[i].algnmt_1 == NOT_ALIGNED) res[i].score_1 = 0;
        if (res[i].algnmt_2 == NOT_ALIGNED) res[i].score_2 = 0;
        if (chr1 == chr2 && res[i].algnmt_1 != NOT_ALIGNED && res[i].algnmt_2 != NOT_ALIGNED) {
            int sum = res[i].score_1 + res[i].score_2;
            res[i].score_1 = sum; res[i].score_2 = sum;
        }
        if (best1 < res[i].score_1) best1 = res[i].score_1;
        if (best2 < res[i].score_2) best2 = res[i].score_2;
        if (chr1 != -1) { h1[n1].chr = chr1; h1[n1].neg_score = -res[i].score_1; n1++; }
        if (chr2 != -1) { h2[n2].chr = chr2; h2[n2].neg_score = -res[i].score_2; n2++; }
    }
    qsort(h1, (size_t)n1, sizeof(chr_hit), cmp_chr_hit);
    qsort(h2, (size_t)n2, sizeof(chr_hit), cmp_chr_hit);
    hdr_map v1[256], v2[256]; int nv1, nv2;
    int prev1 = mapping_from_header(rs->comment[id1], v1, &nv1, 256, P->top_percentage, best1 * P->top_percentage);
    int prev2 = mapping_from_header(rs->comment[id2], v2, &nv2, 256, P->top_percentage, best2 * P->top_percentage);
    if (prev1 > best1) best1 = prev1;
    if (prev2 > best2) best2 = prev2;
    put_score_line(out, ix, rs, id1, best1, h1, n1, v1, nv1, P->top_percentage);
    put_seq_qual(out, rs, id1);
    put_score_line(out, ix, rs, id2, best2, h2, n2, v2, nv2, P->top_percentage);
    put_seq_qual(out, rs, id2);
    free(h1); free(h2);
}

/* ------------------------------------------------------------------ single-end hit (PEAlgnmt.h SingleAlgnmtResult) */
typedef struct {
    uint32_t read_id;
    int strand, score, seed_len;
    uint64_t algnmt;
    char *cigar;
    int editdist, count;
} single_result;

/* unproperlypairDPOutputFastqAPI (BGS-IO.cpp:1384-1446) */
static void output_single_fastq(FILE *out, const mpo_index *ix, const read_set *rs, const mpo_align_params *P,
                                int id, single_result *hits, int n)
{
    int best = 0, nh = 0;
    if (P->megapath_mode == 2) n = 0;
    if (rs->comment[id] && strcmp(rs->comment[id], "IGNORE") == 0) {
        fprintf(out, "@%s\tIGNORE\n", rs->name[id]);
        put_seq_qual(out, rs, id);
        return;
    }
    chr_hit *h = (chr_hit *)malloc(sizeof(chr_hit) * (size_t)(n + 1));
    for (int i = 0; i < n; i++) {
        int chr = target_chr(ix, hits[i].algnmt, rs->len[id]);
        if (chr < 0) { hits[i].algnmt = NOT_ALIGNED; hits[i].score = 0; }
        else { h[nh].chr = chr; h[nh].neg_score = -hits[i].score; nh++; }
        if (best < hits[i].score) best = hits[i].score;
    }
    qsort(h, (size_t)nh, sizeof(chr_hit), cmp_chr_hit);
    hdr_map v[256]; int nv;
    int prev = mapping_from_header(rs->comment[id], v, &nv, 256, P->top_percentage, best * P->top_percentage);
    if (prev > best) best = prev;
    put_score_line(out, ix, rs, id, best, h, nh, v, nv, P->top_percentage);
    put_seq_qual(out, rs, id);
    free(h);
}

/* ------------------------------------------------------------------ pairing (pairEndMerge, DV-DPfunctions.cpp:1968-2070) */
static void pair_end_merge(vec *cands, mpo_seed_pos *left, int n_left, const mpo_seed_pos *right, int n_right,
                           int is_mate_positive, int right_leg_len, const mpo_align_params *P, uint32_t even_id)
{
    if (n_left == 0 || n_right == 0) return;
    /* MC_Compress: drop left positions within DIVIDE_GAP of the previously kept one (in place) */
    int k = 0;
    uint64_t prev = left[0].pos;
    for (int i = 1; i < n_left; i++) {
        if (prev + DIVIDE_GAP < left[i].pos) { left[++k] = left[i]; prev = left[i].pos; }
    }
    n_left = k + 1;
    int margin = MARGIN(right_leg_len);
    int length_low = P->insert_low - right_leg_len - margin;
    if (length_low < 0) length_low = 0;
    int length_high = P->insert_high - right_leg_len + margin;
    int pre_start = 0;
    for (int a = 0; a < n_left; a++) {
        uint64_t read_loc = left[a].pos;
        for (int b = pre_start; b < n_right; b++) {
            uint64_t mate_loc = right[b].pos;
            if (read_loc + (uint64_t)(int64_t)length_high < mate_loc) break;
            else if (read_loc + (uint64_t)(int64_t)length_low <= mate_loc) {
                mpo_candidate *c = (mpo_candidate *)vec_push(cands);
                c->pos[0] = read_loc; c->pos[1] = mate_loc;
                c->read_id_left = even_id + (uint32_t)is_mate_positive;
                pre_start = b;                   /* App. D7: overwritten at every match */
            }
        }
    }
}

/* ------------------------------------------------------------------ window helpers (packLeft / packRight / pack) */
static void dp_window(uint64_t seed_pos, int read_len, uint64_t text_len, uint64_t *start, int *wlen)
{
    int margin = MARGIN(read_len);
    uint64_t s = seed_pos - (uint64_t)margin;
    if (s >= text_len) s = 0;                       /* DV-DPfunctions.cpp:2871-2874 */
    uint64_t len = (uint64_t)read_len + 2 * (uint64_t)margin;
    if (s + len > text_len) len = text_len - s;
    *start = s; *wlen = (int)(uint32_t)len;
}

/* ------------------------------------------------------------------ rescue stages (defined in rescue.c part below) */
typedef struct {
    vec *pool;              /* per read: vec of pooled seeds {pos, strand(1|2), seedLength} */
} seed_pool;
typedef struct { uint64_t pos; int strand; uint32_t seed_len; } pool_seed;

typedef struct { uint32_t read_id; uint64_t pos; int strand; uint32_t seed_len; } single_cand;

static int cmp_single_cand(const void *a, const void *b)
{
    /* RadixTraitsCandidateInfo (DV-DPForSingleReads.cpp:108-118) */
    const single_cand *x = (const single_cand *)a, *y = (const single_cand *)b;
    if (x->read_id != y->read_id) return x->read_id < y->read_id ? -1 : 1;
    if (x->strand != y->strand) return x->strand < y->strand ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    if (x->seed_len != y->seed_len) return x->seed_len < y->seed_len ? -1 : 1;
    return 0;
}

/* std::sort(first, last, LongerSeedLength()) as singleMerge calls it (DV-DPfunctions.cpp:289-293,334).  The order among
 * seeds of equal length is output-visible: transferSeed keeps the first 200 candidates of a read
 * (DV-DPForSingleReads.cpp:200), so the permutation is restated here step for step.  The reference is built with
 * g++ 11 / libstdc++ (oracle/Makefile.ref), whose std::sort is (bits/stl_algo.h: __sort, __introsort_loop,
 * __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort; bits/stl_heap.h for the depth-limit
 * fallback): quicksort on the median of (first+1, middle, last-1) moved to `first`, Hoare partition over
 * [first+1, last), right part recursed, left part looped, until a range has <= 16 elements or 2*floor(log2 n) levels
 * are used up (then heap sort of that range); finally one insertion sort pass over the whole array, guarded for the
 * first 16 elements and unguarded after them.  Pinned by tests/golden/g2 (reads with > 200 tied candidates). */
#define LSL_LESS(x, y) ((x).seed_len > (y).seed_len)      /* comp(x, y): x sorts before y */
static void lsl_swap(single_cand *x, single_cand *y) { single_cand t = *x; *x = *y; *y = t; }

static void lsl_sift(single_cand *a, int hole, int len, single_cand v)      /* __adjust_heap + __push_heap */
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (LSL_LESS(a[child], a[child - 1])) child--;
        a[hole] = a[child]; hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top && LSL_LESS(a[parent], v)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
    a[hole] = v;
}

static void lsl_heap_sort(single_cand *a, int n)            /* __partial_sort(first, last, last) */
{
    if (n >= 2) for (int parent = (n - 2) / 2;; parent--) { lsl_sift(a, parent, n, a[parent]); if (parent == 0) break; }
    for (int last = n; last > 1;) { --last; single_cand v = a[last]; a[last] = a[0]; lsl_sift(a, 0, last, v); }
}

static void lsl_introsort_loop(single_cand *a, int first, int last, int depth)
{
    while (last - first > 16) {
        if (depth == 0) { lsl_heap_sort(a + first, last - first); return; }
        --depth;
        single_cand *x = &a[first + 1], *y = &a[first + (last - first) / 2], *z = &a[last - 1], *res = &a[first];
        if (LSL_LESS(*x, *y)) { if (LSL_LESS(*y, *z)) lsl_swap(res, y); else if (LSL_LESS(*x, *z)) lsl_swap(res, z); else lsl_swap(res, x); }
        else if (LSL_LESS(*x, *z)) lsl_swap(res, x);
        else if (LSL_LESS(*y, *z)) lsl_swap(res, z);
        else lsl_swap(res, y);
        int lo = first + 1, hi = last;
        for (;;) {
            while (LSL_LESS(a[lo], a[first])) ++lo;
            --hi;
            while (LSL_LESS(a[first], a[hi])) --hi;
            if (!(lo < hi)) break;
            lsl_swap(&a[lo], &a[hi]);
            ++lo;
        }
        lsl_introsort_loop(a, lo, last, depth);
        last = lo;
    }
}

static void sort_longer_seed(single_cand *a, int n)
{
    if (n <= 0) return;
    int lg = 0;
    while ((n >> (lg + 1)) != 0) lg++;
    lsl_introsort_loop(a, 0, n, 2 * lg);
    /* __final_insertion_sort */
    const int guarded = n > 16 ? 16 : n;
    for (int i = 1; i < guarded; i++) {
        single_cand v = a[i];
        if (LSL_LESS(v, a[0])) { memmove(&a[1], &a[0], sizeof(single_cand) * (size_t)i); a[0] = v; }
        else { int j = i; while (LSL_LESS(v, a[j - 1])) { a[j] = a[j - 1]; j--; } a[j] = v; }
    }
    for (int i = guarded; i < n; i++) { single_cand v = a[i]; int j = i; while (LSL_LESS(v, a[j - 1])) { a[j] = a[j - 1]; j--; } a[j] = v; }
}

/* test hook: the permutation sort_longer_seed applies to n seeds of the given lengths (perm[k] = input index at rank k) */
void mpo_sort_longer_seed_perm(const uint32_t *seed_len, int n, int32_t *perm)
{
    single_cand *a = (single_cand *)malloc(sizeof(single_cand) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { a[i].read_id = (uint32_t)i; a[i].pos = 0; a[i].strand = 1; a[i].seed_len = seed_len[i]; }
    sort_longer_seed(a, n);
    for (int i = 0; i < n; i++) perm[i] = (int32_t)a[i].read_id;
    free(a);
}

static int cmp_single_result(const void *a, const void *b)
{
    /* ResultCompare<SingleAlgnmtResult> (DV-DPfunctions.cpp:248-251) */
    const single_result *x = (const single_result *)a, *y = (const single_result *)b;
    if (x->algnmt != y->algnmt) return x->algnmt < y->algnmt ? -1 : 1;
    if (x->score != y->score) return x->score < y->score ? -1 : 1;
    return 0;
}

void mpo_default_params(mpo_align_params *p)
{
    memset(p, 0, sizeof *p);
    p->max_read_length = 120; p->insert_low = 1; p->insert_high = 500;      /* IniParam.cpp:627-648 */
    p->clip_left = 130; p->clip_right = 130;                                /* soap4-nt2.ini:83-85 */
    p->megapath_mode = 1; p->top_percentage = 0.95; p->quality_constant = 33;
    p->skip_default_dp = 0;
    p->mmp.sa_size_threshold = 30; p->mmp.seed_min_length = 17; p->mmp.uniq_threshold = 6; p->mmp.indel_fuzz = 5;
    p->mmp.good_seed_len = 27; p->mmp.reseed_len = 18; p->mmp.reseed_rlt_ratio = 0.7; p->mmp.reseed_abs_diff = 4;
    p->mmp.short_seed_ratio = 0.5;
}

#include "samout.inc"
#include "rescue.inc"

int mpo_align_fastq(const mpo_index *ix, const mpo_align_params *Pin, const char *fq1, const char *fq2,
                    const char *out_fastq, const char *out_records, mpo_align_stats *stats)
{
    return mpo_align_fastq_bam(ix, Pin, fq1, fq2, out_fastq, out_records, NULL, 0, NULL, stats);
}

int mpo_align_fastq_bam(const mpo_index *ix, const mpo_align_params *Pin, const char *fq1, const char *fq2,
                        const char *out_fastq, const char *out_records, const char *bam_text_prefix, int print_mdnm,
                        const char *read_group, mpo_align_stats *stats)
{
    mpo_align_params Pv = *Pin, *P = &Pv;
    mpo_align_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    fill_char_map();
    read_set rs;
    if (load_fastq_pair(fq1, fq2, P->max_read_length, P->quality_constant, &rs)) return -1;
    if (P->ignore_comments)                                   /* SOAP4.cpp:537-547 */
        for (int i = 0; i < rs.n; i++) { free(rs.comment[i]); rs.comment[i] = NULL; }
    FILE *out = fopen(out_fastq, "wb");
    FILE *rec = out_records ? fopen(out_records, "wb") : NULL;
    if (!out) { fprintf(stderr, "[oracle] cannot write %s\n", out_fastq); return -1; }
    sam_ctx sam_store, *sam = NULL;
    if (bam_text_prefix) {
        static const char *ext[3] = {".dpout.txt", ".gout.txt", ".unpair.txt"};
        char path[4096];
        sam = &sam_store; sam_ctx_init(sam);
        sam->print_mdnm = print_mdnm; sam->read_group = read_group ? read_group : fq1;
        for (int k = 0; k < 3; k++) {
            snprintf(path, sizeof path, "%s%s", bam_text_prefix, ext[k]);
            if (!(sam->f[k] = fopen(path, "wb"))) { fprintf(stderr, "[oracle] cannot write %s\n", path); return -1; }
        }
        /* SAMOutputHeaderConstruct (SAM.cpp:83-132); sample name and read-group option default to "default" and "" */
        snprintf(path, sizeof path, "%s.header.txt", bam_text_prefix);
        FILE *h = fopen(path, "wb");
        if (!h) return -1;
        fprintf(h, "@HD\tVN:1.3\tSO:unsorted\n@RG\tID:%s\tSM:%s\t%s\n", sam->read_group, "default", "");
        for (int i = 0; i < mpo_num_seqs(ix); i++) fprintf(h, "@SQ\tSN:%s\tLN:%u\n", mpo_seq_name(ix, i), (unsigned)mpo_seq_actual_length(ix, i));
        fclose(h);
    }
    const uint64_t text_len = mpo_text_length(ix);
    const int n_pairs = rs.n / 2;
    stats->n_pairs = (uint64_t)n_pairs;

    /* SOAP4.cpp:465-474: insert_low is raised to the detected read lengths (first batch only) */
    {
        int d1 = detect_read_length(rs.len, rs.n, 2), d2 = detect_read_length(rs.len + 1, rs.n, 2);
        if (P->insert_low < d2) P->insert_low = d2;
        if (P->insert_low < d1) P->insert_low = d1;
    }

    /* ---- seeding (every read; kept seeds also go to the pool, DV-DPfunctions.cpp:2604-2610) */
    int cap = 0;
    for (int i = 0; i < rs.n; i++) if (rs.len[i] > cap) cap = rs.len[i];
    cap = (cap + 2) * 64;
    mpo_seed_pos *sp_pos = (mpo_seed_pos *)malloc(sizeof(mpo_seed_pos) * (size_t)cap * 2);
    mpo_seed_pos *sp_neg = sp_pos + cap;
    seed_pool pool; pool.pool = (vec *)malloc(sizeof(vec) * (size_t)rs.n);
    for (int i = 0; i < rs.n; i++) vec_init(&pool.pool[i], sizeof(pool_seed));
    /* per-read kept seeds, both strands */
    vec *pos_list = (vec *)malloc(sizeof(vec) * (size_t)rs.n), *neg_list = (vec *)malloc(sizeof(vec) * (size_t)rs.n);
    for (int i = 0; i < rs.n; i++) {
        int np, nn;
        vec_init(&pos_list[i], sizeof(mpo_seed_pos)); vec_init(&neg_list[i], sizeof(mpo_seed_pos));
        mpo_seed_read(ix, rs.seq[i], rs.len[i], (uint32_t)(i & ~1), &P->mmp, sp_pos, &np, sp_neg, &nn, cap, &stats->seed);
        for (int k = 0; k < np; k++) *(mpo_seed_pos *)vec_push(&pos_list[i]) = sp_pos[k];
        for (int k = 0; k < nn; k++) *(mpo_seed_pos *)vec_push(&neg_list[i]) = sp_neg[k];
        if (i & 1) stats->n_mate_seedpos += (uint64_t)(np + nn); else stats->n_read_seedpos += (uint64_t)(np + nn);
        /* pool order: all '+' seeds, then all '-' seeds (DV-DPfunctions.cpp:2604-2610) */
        for (int k = 0; k < np; k++) { pool_seed *s = (pool_seed *)vec_push(&pool.pool[i]); s->pos = sp_pos[k].pos; s->strand = 1; s->seed_len = sp_pos[k].paired_seed_length & 0x7FFFFFFF; }
        for (int k = 0; k < nn; k++) { pool_seed *s = (pool_seed *)vec_push(&pool.pool[i]); s->pos = sp_neg[k].pos; s->strand = 2; s->seed_len = sp_neg[k].paired_seed_length & 0x7FFFFFFF; }
    }

    /* ---- pairing + deep DP, pair by pair (candidates of one pair never split across batches) */
    uint8_t *pair_done = (uint8_t *)calloc((size_t)n_pairs + 1, 1);
    for (int p = 0; p < n_pairs; p++) {
        int r = 2 * p, m = r + 1;
        vec cands; vec_init(&cands, sizeof(mpo_candidate));
        /* read '+' with mate '-' (readIDLeft even); then mate '+' with read '-' (readIDLeft odd);
         * stable sort by readIDLeft keeps exactly this order (DV-DPfunctions.cpp:2098-2111) */
        pair_end_merge(&cands, (mpo_seed_pos *)pos_list[r].p, (int)pos_list[r].n, (mpo_seed_pos *)neg_list[m].p, (int)neg_list[m].n,
                       0, rs.len[m], P, (uint32_t)r);
        pair_end_merge(&cands, (mpo_seed_pos *)pos_list[m].p, (int)pos_list[m].n, (mpo_seed_pos *)neg_list[r].p, (int)neg_list[r].n,
                       1, rs.len[r], P, (uint32_t)r);
        stats->n_candidates += cands.n;
        vec results; vec_init(&results, sizeof(deep_result));
        for (size_t c = 0; c < cands.n; c++) {
            mpo_candidate *ci = (mpo_candidate *)cands.p + c;
            int left = (int)ci->read_id_left, right = left ^ 1;
            uint64_t ws[2]; int wl[2]; dp_out o[2];
            dp_window(ci->pos[0], rs.len[left], text_len, &ws[0], &wl[0]);
            run_dp(ix, &rs, left, 1, ws[0], wl[0], P, &o[0], stats);
            if (o[0].score < mpo_dp_cutoff(rs.len[left])) { free(o[0].cigar); continue; }
            /* packRight (DV-DPfunctions.cpp:2926-3005) */
            dp_window(ci->pos[1], rs.len[right], text_len, &ws[1], &wl[1]);
            uint64_t hit_left = ws[0] + (uint64_t)o[0].hit_loc;
            uint64_t bounded = hit_left + (uint64_t)(int64_t)P->insert_high - ws[1];
            if (bounded < (uint64_t)(uint32_t)wl[1]) wl[1] = (int)(uint32_t)bounded;
            run_dp(ix, &rs, right, 2, ws[1], wl[1], P, &o[1], stats);
            if (o[1].score < mpo_dp_cutoff(rs.len[right])) { free(o[0].cigar); free(o[1].cigar); continue; }
            int read_side = left & 1, mate_side = 1 - read_side;
            deep_result *d = (deep_result *)vec_push(&results);
            d->read_id = (uint32_t)r;
            d->strand_1 = read_side == 0 ? 1 : 2; d->strand_2 = mate_side == 0 ? 1 : 2;
            d->algnmt_1 = ws[read_side] + (uint64_t)o[read_side].hit_loc;
            d->algnmt_2 = ws[mate_side] + (uint64_t)o[mate_side].hit_loc;
            d->score_1 = o[read_side].score; d->score_2 = o[mate_side].score;
            d->cigar_1 = o[read_side].cigar; d->cigar_2 = o[mate_side].cigar;
            d->start_1 = ws[read_side]; d->start_2 = ws[mate_side];
            d->dplen_1 = wl[read_side]; d->dplen_2 = wl[mate_side];
            {
                int ed[2], dis[2];
                for (int k = 0; k < 2; k++) ed[k] = edit_dist(o[k].cigar, rs.len[right], o[k].score, &dis[k]);   /* batch->lengths[i]: the right read's */
                d->editdist_1 = ed[read_side]; d->editdist_2 = ed[mate_side];
                d->count_1 = o[read_side].count; d->count_2 = o[mate_side].count;
                if (d->algnmt_1 < d->algnmt_2) d->insert = (long long)(d->algnmt_2 - d->algnmt_1) + rs.len[right] + dis[1];
                else d->insert = (long long)(d->algnmt_1 - d->algnmt_2) + rs.len[right] + dis[1];
            }
        }
        if (results.n > 0) {
            deep_result *res = (deep_result *)results.p;
            /* OutputBuffer::ready -> arrayCopyNRemoveDuplicate (DV-DPfunctions.h:165-196) */
            qsort(res, results.n, sizeof(deep_result), cmp_deep);
            size_t k = 0;
            for (size_t i = 1; i < results.n; i++) {
                if (cmp_deep(&res[k], &res[i]) < 0) res[++k] = res[i];
                else { free(res[i].cigar_1); free(res[i].cigar_2); }
            }
            size_t nres = k + 1;
            /* best = first maximum of score_1 + score_2 (OutputDPResult.cpp:156-232) */
            size_t best = 0;
            for (size_t i = 1; i < nres; i++)
                if (res[i].score_1 + res[i].score_2 > res[best].score_1 + res[best].score_2) best = i;
            if (rec) {
                for (int k2 = 0; k2 < 2; k2++) {
                    uint64_t pos = k2 ? res[best].algnmt_2 : res[best].algnmt_1, tp; unsigned chr;
                    mpo_chr_and_pos(ix, pos, &tp, &chr);
                    fprintf(rec, "%s\t%d\tdeep\t%s\t%llu\t%c\t%d\t%s\n", rs.name[r + k2], k2 + 1, mpo_seq_name(ix, (int)chr - 1),
                            (unsigned long long)tp, (k2 ? res[best].strand_2 : res[best].strand_1) == 1 ? '+' : '-',
                            k2 ? res[best].score_2 : res[best].score_1, k2 ? res[best].cigar_2 : res[best].cigar_1);
                }
            }
            output_pair_fastq(out, ix, &rs, P, res, (int)nres);
            if (sam) { sam->stage = 1; sam_output_pair(sam, 0, ix, &rs, res, (int)nres, (int)best); }
            for (size_t i = 0; i < nres; i++) { free(res[i].cigar_1); free(res[i].cigar_2); }
            /* freeSeedAlignmentsOfReadWithReadId for properly paired reads (OutputDPResult.cpp:123-127) */
            vec_free(&pool.pool[r]); vec_free(&pool.pool[m]);
            pair_done[p] = 1;
            stats->n_deep_dp_pairs++;
        }
        vec_free(&results); vec_free(&cands);
    }

    rescue_and_output(ix, P, &rs, &pool, pair_done, out, rec, sam, stats);
    if (sam) for (int k = 0; k < 3; k++) fclose(sam->f[k]);

    for (int i = 0; i < rs.n; i++) { vec_free(&pool.pool[i]); vec_free(&pos_list[i]); vec_free(&neg_list[i]); }
    free(pool.pool); free(pos_list); free(neg_list); free(sp_pos); free(pair_done);
    if (rec) fclose(rec);
    fclose(out);
    free_reads(&rs);
    return 0;
}
