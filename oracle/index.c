/*
 * oracle/index.c -- CPU ORACLE (test infrastructure only): 2BWT index reader and FM-index primitives.
 *
 * Follows the on-disk formats written by the reference builder:
 *   .bwt  2bwt-lib/BWTConstruct.c:1217-1220, read side BWT.c:132-157
 *   .fmv  BWTConstruct.c:1229-1233, BWT.c:783-797 (explicit occ), sizes BWT.c:870-889
 *   .sa   BWTConstruct.c:1380-1385, BWT.c:200-243
 *   .lkt  2bwt-flex/LT.c:37-57, builder LTConstruct.c:44-105
 *   .pac  HSP.c:551-567, TextConverter.c:447-477
 *   .ann  HSP.c:87-118      .tra  HSP.c:232-290
 * and restates BWTOccValue (BWT.c:597-634), BWTPsiMinusValue (:915-938), BWTSaValue (:968-998) and
 * getChrAndPos (BGS-IO.cpp:163-188) with plain scalar loops.
 */
#include "mp_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OCC_INTERVAL 256
#define OCC_MAJOR 65536
#define LKT_K 13

struct mpo_index {
    uint64_t n;            /* text length */
    uint64_t inv_sa0;
    uint64_t cum[5];
    uint32_t *bwt;         /* 16 symbols / word, MSB first, '$' omitted */
    uint64_t bwt_words;
    uint32_t *occ_minor;   /* 2 x u16 per word, 4 words per sample pair */
    uint64_t *occ_major;   /* 4 x u64 per 65536 positions */
    uint64_t sa_interval;
    uint64_t *sa;          /* sa[0] = (uint64)-1 as in BWT.c:242 */
    uint64_t sa_count;
    uint8_t *pac;          /* 4 bases per byte, first base in bits 7-6 */
    uint64_t *lkt;         /* 4^13 inclusive right bounds */
    int num_seq;
    char **seq_name;
    uint64_t *seq_start, *seq_len;
    uint64_t *seq_actual_len; /* .tra tail: length with the removed N-runs (SAM header LN, SAM.cpp:117) */
    uint64_t n_removed;       /* .tra header: numOfRemovedSegment */
    /* translate (.tra) */
    uint32_t n_grid;
    uint32_t *grid;
    uint32_t n_trans;
    uint64_t *tr_start, *tr_corr;
    uint32_t *tr_chr;
};

static void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "[oracle] out of memory (%zu bytes)\n", n); exit(1); }
    return p;
}

static FILE *open_ext(const char *prefix, const char *ext, int must)
{
    char path[4096];
    snprintf(path, sizeof path, "%s%s", prefix, ext);
    FILE *f = fopen(path, "rb");
    if (!f && must) { fprintf(stderr, "[oracle] cannot open %s\n", path); exit(1); }
    return f;
}

static void xread(void *dst, size_t sz, size_t cnt, FILE *f, const char *what)
{
    if (cnt && fread(dst, sz, cnt, f) != cnt) { fprintf(stderr, "[oracle] short read on %s\n", what); exit(1); }
}

static void build_lkt(mpo_index *ix)
{
    /* LTConstruct.c:44-105: count every 13-mer window of (text + 12 x 'A'), then prefix-sum. */
    uint64_t size = 1ull << (2 * LKT_K), mask = size - 1, w = 0, i;
    uint64_t *t = (uint64_t *)calloc(size, sizeof(uint64_t));
    if (!t) { fprintf(stderr, "[oracle] out of memory for LKT\n"); exit(1); }
    for (i = 0; i < ix->n + LKT_K - 1; i++) {
        unsigned c = i < ix->n ? mpo_text_base(ix, i) : 0;
        w = ((w << 2) | c) & mask;
        if (i >= LKT_K - 1) t[w]++;
    }
    for (i = 1; i < size; i++) t[i] += t[i - 1];
    ix->lkt = t;
}

mpo_index *mpo_index_open(const char *prefix)
{
    mpo_index *ix = (mpo_index *)calloc(1, sizeof *ix);
    uint64_t hdr[5], i;
    FILE *f = open_ext(prefix, ".bwt", 1);
    xread(hdr, 8, 5, f, ".bwt header");
    ix->inv_sa0 = hdr[0];
    ix->cum[0] = 0;
    for (i = 1; i <= 4; i++) ix->cum[i] = hdr[i];
    ix->n = ix->cum[4];
    {
        uint64_t in_file = (ix->n + 15) / 16;
        uint64_t resident = ((ix->n + OCC_INTERVAL - 1) / OCC_INTERVAL * OCC_INTERVAL + 15) / 16 + 8;
        ix->bwt = (uint32_t *)calloc(resident + 8, 4);
        ix->bwt_words = resident;
        xread(ix->bwt, 4, in_file, f, ".bwt payload");
        /* BWTClearTrailingBwtCode (BWT.c:891-912): symbols past n read as 'A' but are never counted */
        if (ix->n % 16) ix->bwt[ix->n / 16] &= ~0u << (32 - 2 * (ix->n % 16));
    }
    fclose(f);

    f = open_ext(prefix, ".fmv", 1);
    xread(hdr, 8, 5, f, ".fmv header");
    if (hdr[0] != ix->inv_sa0 || hdr[4] != ix->n) { fprintf(stderr, "[oracle] .fmv header mismatch\n"); exit(1); }
    {
        uint64_t n_occ = (ix->n + OCC_INTERVAL - 1) / OCC_INTERVAL + 1;
        uint64_t minor_words = (n_occ + 1) / 2 * 4;
        uint64_t major_words = (n_occ + (OCC_MAJOR / OCC_INTERVAL) - 1) / (OCC_MAJOR / OCC_INTERVAL) * 4;
        ix->occ_minor = (uint32_t *)xmalloc(minor_words * 4);
        ix->occ_major = (uint64_t *)xmalloc(major_words * 8);
        xread(ix->occ_minor, 4, minor_words, f, ".fmv minor");
        xread(ix->occ_major, 8, major_words, f, ".fmv major");
    }
    fclose(f);

    f = open_ext(prefix, ".sa", 1);
    xread(hdr, 8, 5, f, ".sa header");
    xread(&ix->sa_interval, 8, 1, f, ".sa interval");
    ix->sa_count = (ix->n + ix->sa_interval) / ix->sa_interval;
    ix->sa = (uint64_t *)xmalloc(ix->sa_count * 8);
    xread(ix->sa, 8, ix->sa_count, f, ".sa payload");
    ix->sa[0] = (uint64_t)-1;
    fclose(f);

    f = open_ext(prefix, ".pac", 1);
    {
        uint64_t bytes = ix->n / 4 + 2;
        ix->pac = (uint8_t *)calloc(bytes + 8, 1);
        if (fread(ix->pac, 1, bytes, f) < ix->n / 4 + 1) { fprintf(stderr, "[oracle] short .pac\n"); exit(1); }
    }
    fclose(f);

    f = open_ext(prefix, ".ann", 1);
    {
        unsigned long long n2; int ns; unsigned seed;
        if (fscanf(f, "%llu %d %u\n", &n2, &ns, &seed) != 3 || n2 != ix->n) { fprintf(stderr, "[oracle] bad .ann\n"); exit(1); }
        ix->num_seq = ns;
        ix->seq_name = (char **)xmalloc(sizeof(char *) * ns);
        ix->seq_start = (uint64_t *)xmalloc(8 * ns);
        ix->seq_len = (uint64_t *)xmalloc(8 * ns);
        for (int s = 0; s < ns; s++) {
            unsigned gi; char line[2048]; unsigned long long a, b; int z;
            if (fscanf(f, "%u ", &gi) != 1 || !fgets(line, sizeof line, f)) { fprintf(stderr, "[oracle] bad .ann entry\n"); exit(1); }
            size_t L = strlen(line);
            if (L && line[L - 1] == '\n') line[L - 1] = 0;
            ix->seq_name[s] = strdup(line);
            if (fscanf(f, "%llu %llu %d\n", &a, &b, &z) != 3) { fprintf(stderr, "[oracle] bad .ann offsets\n"); exit(1); }
            ix->seq_start[s] = a; ix->seq_len[s] = b;
        }
    }
    fclose(f);

    f = open_ext(prefix, ".tra", 1);
    {
        unsigned long long n2; int ns; unsigned rem, grid;
        if (fscanf(f, "%llu %d %u %u\n", &n2, &ns, &rem, &grid) != 4) { fprintf(stderr, "[oracle] bad .tra\n"); exit(1); }
        ix->n_grid = grid;
        ix->grid = (uint32_t *)xmalloc(4 * (size_t)grid);
        for (unsigned g = 0; g < grid; g++) { unsigned v; if (fscanf(f, "%u\n", &v) != 1) exit(1); ix->grid[g] = v; }
        ix->n_trans = ns + rem;
        ix->tr_start = (uint64_t *)xmalloc(8 * (size_t)ix->n_trans);
        ix->tr_corr = (uint64_t *)xmalloc(8 * (size_t)ix->n_trans);
        ix->tr_chr = (uint32_t *)xmalloc(4 * (size_t)ix->n_trans);
        for (unsigned t = 0; t < ix->n_trans; t++) {
            unsigned long long a, c; unsigned b;
            if (fscanf(f, "%llu %u %llu\n", &a, &b, &c) != 3) { fprintf(stderr, "[oracle] bad .tra row\n"); exit(1); }
            ix->tr_start[t] = a; ix->tr_chr[t] = b; ix->tr_corr[t] = c;
        }
        ix->n_removed = rem;
        ix->seq_actual_len = (uint64_t *)xmalloc(8 * (size_t)ns);
        for (int t = 0; t < ns; t++) {
            unsigned long long a, b;
            /* HSP.c:286-302: older indexes lack these rows; the packed lengths are used then */
            if (fscanf(f, "%llu %llu\n", &a, &b) == 2) ix->seq_actual_len[t] = b; else ix->seq_actual_len[t] = ix->seq_len[t];
        }
    }
    fclose(f);

    f = open_ext(prefix, ".lkt", 0);
    if (f) {
        int k;
        xread(&k, 4, 1, f, ".lkt header");
        if (k != LKT_K) { fprintf(stderr, "[oracle] lookup table size %d unsupported\n", k); exit(1); }
        ix->lkt = (uint64_t *)xmalloc(8ull << (2 * LKT_K));
        xread(ix->lkt, 8, 1ull << (2 * LKT_K), f, ".lkt payload");
        fclose(f);
    } else {
        build_lkt(ix);
    }
    return ix;
}

void mpo_index_close(mpo_index *ix)
{
    if (!ix) return;
    free(ix->bwt); free(ix->occ_minor); free(ix->occ_major); free(ix->sa); free(ix->pac); free(ix->lkt);
    for (int i = 0; i < ix->num_seq; i++) free(ix->seq_name[i]);
    free(ix->seq_name); free(ix->seq_start); free(ix->seq_len); free(ix->seq_actual_len);
    free(ix->grid); free(ix->tr_start); free(ix->tr_corr); free(ix->tr_chr);
    free(ix);
}

uint64_t mpo_text_length(const mpo_index *ix) { return ix->n; }
uint64_t mpo_inverse_sa0(const mpo_index *ix) { return ix->inv_sa0; }
uint64_t mpo_sa_interval(const mpo_index *ix) { return ix->sa_interval; }
uint64_t mpo_cum_freq(const mpo_index *ix, int c) { return ix->cum[c]; }
int mpo_num_seqs(const mpo_index *ix) { return ix->num_seq; }
const char *mpo_seq_name(const mpo_index *ix, int i) { return ix->seq_name[i]; }
const uint64_t *mpo_lkt_table(const mpo_index *ix) { return ix->lkt; }

static inline unsigned bwt_sym(const mpo_index *ix, uint64_t k)
{
    return (ix->bwt[k >> 4] >> (2 * (15 - (k & 15)))) & 3;
}

/* rank with '$' already removed from the numbering: # of c in BWT'[0, k) */
static uint64_t occ_raw(const mpo_index *ix, uint64_t k, unsigned c)
{
    uint64_t e = (k + OCC_INTERVAL / 2 - 1) / OCC_INTERVAL;   /* bidirectional: nearest explicit sample */
    uint64_t at = e * OCC_INTERVAL;
    uint32_t w = ix->occ_minor[e / 2 * 4 + c];
    uint64_t v = ix->occ_major[at / OCC_MAJOR * 4 + c] + ((e & 1) ? (w & 0xFFFF) : (w >> 16));
    uint64_t j;
    if (at < k) { for (j = at; j < k; j++) v += bwt_sym(ix, j) == c; }
    else { for (j = k; j < at; j++) v -= bwt_sym(ix, j) == c; }
    return v;
}

uint64_t mpo_occ(const mpo_index *ix, uint64_t idx, unsigned c)
{
    idx -= (idx > ix->inv_sa0);
    return occ_raw(ix, idx, c);
}

uint64_t mpo_psi_minus(const mpo_index *ix, uint64_t idx)
{
    if (idx == ix->inv_sa0) return 0;
    uint64_t k = idx + 1;
    k -= (k > ix->inv_sa0);
    unsigned c = bwt_sym(ix, k - 1);
    return ix->cum[c] + occ_raw(ix, k, c);
}

uint64_t mpo_sa_value(const mpo_index *ix, uint64_t idx)
{
    uint64_t skipped = 0;
    while (idx % ix->sa_interval != 0) { skipped++; idx = mpo_psi_minus(ix, idx); }
    return ix->sa[idx / ix->sa_interval] + skipped;
}

void mpo_lkt_range(const mpo_index *ix, uint32_t key, uint64_t *l, uint64_t *r)
{
    *l = key ? ix->lkt[key - 1] + 1 : 1;
    *r = ix->lkt[key];
}

unsigned mpo_text_base(const mpo_index *ix, uint64_t p)
{
    return (ix->pac[p >> 2] >> (6 - 2 * (p & 3))) & 3;
}

void mpo_chr_and_pos(const mpo_index *ix, uint64_t ambPos, uint64_t *tp, unsigned *chr)
{
    uint64_t g = ambPos >> 18;
    if (g >= ix->n_grid) g = ix->n_grid - 1;   /* the reference reads out of bounds here; positions past the text end */
    uint32_t v = ix->grid[g];
    while (v > 0 && ix->tr_start[v] > ambPos) v--;
    *tp = ambPos - ix->tr_corr[v];
    *chr = ix->tr_chr[v];
}

/* getChrAndPos with its return value (BGS-IO.cpp:163-188): the last packed position of the translate segment, computed
 * as the reference does -- `row < numOfRemovedSegment - 1 ? next row's start - 1 : dnaLength` in unsigned 64-bit, so that
 * with no removed segment every row takes the first branch.  For the last row that branch reads one element past the
 * reference's array (undefined there); the text length is returned here, which gives the same trimming decisions
 * unless that stray value happened to fall inside the read. */
uint64_t mpo_chr_and_pos_seg(const mpo_index *ix, uint64_t ambPos, uint64_t *tp, unsigned *chr)
{
    uint64_t g = ambPos >> 18;
    if (g >= ix->n_grid) g = ix->n_grid - 1;
    uint32_t v = ix->grid[g];
    while (v > 0 && ix->tr_start[v] > ambPos) v--;
    *tp = ambPos - ix->tr_corr[v];
    *chr = ix->tr_chr[v];
    if ((uint64_t)v < ix->n_removed - 1 && v + 1 < ix->n_trans) return ix->tr_start[v + 1] - 1;
    return ix->n;
}
uint64_t mpo_seq_start(const mpo_index *ix, int i) { return ix->seq_start[i]; }
uint64_t mpo_seq_length(const mpo_index *ix, int i) { return ix->seq_len[i]; }
uint64_t mpo_seq_actual_length(const mpo_index *ix, int i) { return ix->seq_actual_len[i]; }
