/*
 * oracle/mp_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain, scalar, single-threaded C restatement of the reference's soap4 paired-end alignment path
 * (SURVEY.md section 8a rows a1-a18).  It exists only so that the HIP path can be checked on the GPU box,
 * where /root/reference never travels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (megapath_amd/) must never link or call it.
 *
 * Parity pin: every stage is checked in tests/ against the *real* reference built by oracle/Makefile.ref
 * (oracle/_ref/libref_bwt.so, libref_dp.so, soap4) and against golden vectors generated from it
 * (tests/golden/, generator tests/golden/make_golden.py).
 */
#ifndef MP_ORACLE_H
#define MP_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ index */
typedef struct mpo_index mpo_index;

/* Opens <prefix>.{bwt,fmv,sa,pac,ann,tra}; <prefix>.lkt is read when present, otherwise the 13-mer table is
 * rebuilt from .pac exactly as the reference builder does (2bwt-lib/LTConstruct.c:44-105). */
mpo_index *mpo_index_open(const char *prefix);
void mpo_index_close(mpo_index *ix);
uint64_t mpo_text_length(const mpo_index *ix);
uint64_t mpo_inverse_sa0(const mpo_index *ix);
uint64_t mpo_sa_interval(const mpo_index *ix);
uint64_t mpo_cum_freq(const mpo_index *ix, int c);
int mpo_num_seqs(const mpo_index *ix);
const char *mpo_seq_name(const mpo_index *ix, int i);
/* rank_c over BWT[0, idx)  == BWTOccValue (2bwt-lib/BWT.c:597-634) */
uint64_t mpo_occ(const mpo_index *ix, uint64_t idx, unsigned c);
/* LF step == BWTPsiMinusValue (BWT.c:915-938) */
uint64_t mpo_psi_minus(const mpo_index *ix, uint64_t idx);
/* locate == BWTSaValue (BWT.c:968-998) */
uint64_t mpo_sa_value(const mpo_index *ix, uint64_t idx);
/* SA range of a 13-mer key as used at DV-DPfunctions.cpp:2245-2246 */
void mpo_lkt_range(const mpo_index *ix, uint32_t key, uint64_t *l, uint64_t *r);
/* base code (0..3) at text position p  (HSP.c:551-567 / TextConverter.c:447-477 layout) */
unsigned mpo_text_base(const mpo_index *ix, uint64_t p);
/* packed-position -> (1-based position in sequence, 1-based sequence id); getChrAndPos (BGS-IO.cpp:163-188) */
void mpo_chr_and_pos(const mpo_index *ix, uint64_t ambPos, uint64_t *tp, unsigned *chr);
const uint64_t *mpo_lkt_table(const mpo_index *ix);
/* getChrAndPos including its return value: last packed position of the translate segment (BGS-IO.cpp:163-188) */
uint64_t mpo_chr_and_pos_seg(const mpo_index *ix, uint64_t ambPos, uint64_t *tp, unsigned *chr);
uint64_t mpo_seq_start(const mpo_index *ix, int i);          /* seqOffset[i].startPos (.ann) */
uint64_t mpo_seq_length(const mpo_index *ix, int i);         /* packed length (.ann) */
uint64_t mpo_seq_actual_length(const mpo_index *ix, int i);  /* seqActualOffset length (.tra tail) = SAM header LN */

/* ------------------------------------------------------------------ DP (rows a9, a10) */
typedef struct {
    int score;          /* best score if >= cutoff and traceback valid, else 0 */
    int hit_loc;        /* 0-based start offset of the alignment inside the window */
    int max_score_count;/* number of eligible cells with the best score, saturating at 255 */
    int n_ops;          /* decoded op string length (ops are in traceback order: end -> start) */
} mpo_dp_result;

/* One instance.  ref[0..ref_len) and read[0..read_len) hold base codes 0..3.
 * ops receives one byte per column consumed, in traceback order, alphabet "MmIDS"
 * (M match, m mismatch, I insertion = read base only, D deletion = window base only, S soft clip).
 * Restates callDP / GenerateDPTable / GPUBacktrack (CPU_DP.cpp:122-978) in the plain Gotoh form of
 * SURVEY App. C.4b/C.4c. */
void mpo_sgdp(const uint8_t *ref, int ref_len, const uint8_t *read, int read_len,
              int clip_l, int clip_r, int cutoff, mpo_dp_result *res, uint8_t *ops, int ops_cap);
/* the reference's DP cutoff: (int) max(0.2 * len, 30.0)  (definitions.h:166-167, DV-DPfunctions.cpp:2916) */
int mpo_dp_cutoff(int read_len);
/* special CIGAR (M m I D S runs) produced by CigarStringEncoder (DV-DPfunctions.h:377-427) from traceback-order ops */
int mpo_cigar_from_ops(const uint8_t *ops, int n_ops, char *out, int cap);

/* ------------------------------------------------------------------ seeding (rows a3-a6) */
typedef struct {
    int sa_size_threshold;  /* mmpSeedSAsizeThreshold = 30 */
    int seed_min_length;    /* mmpSeedMinLength 17 (nt2) / 22 */
    int uniq_threshold;     /* 6 */
    int indel_fuzz;         /* 5 */
    int good_seed_len;      /* 27 */
    int reseed_len;         /* 18 / 23 */
    double reseed_rlt_ratio;/* 0.7 */
    int reseed_abs_diff;    /* 4 */
    double short_seed_ratio;/* 0.5 */
} mpo_mmp_params;

typedef struct {
    uint32_t query_offset;
    uint64_t sa_l;
    uint32_t sa_diff;
    uint32_t seed_len;
} mpo_sa_seed;

typedef struct {
    uint64_t pos;
    uint32_t strand_read_id;     /* bit31 = reverse strand, low 31 = even read id */
    uint32_t paired_seed_length; /* covered length */
} mpo_seed_pos;

typedef struct {
    uint64_t n_occ;   /* rank queries: 2 per backward step, 1 per LF step of locate */
    uint64_t n_lkt;   /* LKT (re)starts */
    uint64_t n_sa;    /* sampled-SA reads */
    uint64_t n_seeds; /* SA-range seeds emitted */
    uint64_t n_hits;  /* located hits */
} mpo_seed_counters;

/* mmp<0> (strand 0) / mmp<2> (strand 1) for one read (DV-DPfunctions.cpp:2224-2377). Returns #seeds. */
int mpo_mmp(const mpo_index *ix, const uint8_t *read, int len, int strand, const mpo_mmp_params *p,
            mpo_sa_seed *out, int cap, mpo_seed_counters *ctr);
/* full per-read seeding: mmp both strands + locate + merge/filter (DV-DPfunctions.cpp:2470-2553).
 * out_pos / out_neg receive SeedPos of the forward / reverse strand in ascending pos. */
void mpo_seed_read(const mpo_index *ix, const uint8_t *read, int len, uint32_t even_read_id,
                   const mpo_mmp_params *p, mpo_seed_pos *out_pos, int *n_pos, mpo_seed_pos *out_neg, int *n_neg,
                   int cap, mpo_seed_counters *ctr);

/* ------------------------------------------------------------------ whole path (rows a1-a18) */
typedef struct {
    int max_read_length;     /* -L  (reads are truncated to L-1, QueryParser.cpp:188) */
    int insert_low;          /* -v  (default 1) */
    int insert_high;         /* -u  */
    int clip_left, clip_right; /* ini MaxFrontLenClipped / MaxEndLenClipped */
    int megapath_mode;       /* 1 = -F, 2 = -P */
    double top_percentage;   /* -top N  -> N/100 */
    int quality_constant;    /* 33 */
    int skip_default_dp;     /* ini SkipDefaultDP */
    mpo_mmp_params mmp;
    int ignore_comments;     /* -nc: drop every FASTQ comment before alignment (SOAP4.cpp:537-547) */
} mpo_align_params;

typedef struct {
    uint64_t n_pairs;
    uint64_t n_candidates;        /* after pairEndMerge */
    uint64_t n_deep_dp_pairs;     /* pairs output by the deep-DP round */
    uint64_t n_read_seedpos, n_mate_seedpos;
    uint64_t n_dp_instances;      /* DP instances run (left + right + rescue) */
    uint64_t dp_cells;            /* sum (refLen+1)*readLen */
    uint64_t n_single_dp_reads, n_default_dp_pairs, n_unpaired_pairs;
    mpo_seed_counters seed;
} mpo_align_stats;

/* Aligns all pairs of two plain FASTQ files; writes the annotated interleaved FASTQ (the reference's stdout in
 * -F / -P mode) to out_fastq and, when out_records != NULL, one TSV line per output mate:
 *   name  mate(1|2)  stage  seqId  pos(1-based)  strand  score  specialCIGAR
 * for the best alignment (what the BAM record would carry).  Returns 0 on success. */
int mpo_align_fastq(const mpo_index *ix, const mpo_align_params *p, const char *fq1, const char *fq2,
                    const char *out_fastq, const char *out_records, mpo_align_stats *stats);

/* Same with the BAM side of `-b` (row a15) restated as text: writes <bam_text_prefix>.dpout.txt / .gout.txt / .unpair.txt,
 * one line per BAM record in the form tests/bamtext.py prints for a decoded BAM file, and <bam_text_prefix>.header.txt
 * (SAMOutputHeaderConstruct, SAM.cpp:83-132).  print_mdnm = -p; read_group = -D or the first read file as given. */
int mpo_align_fastq_bam(const mpo_index *ix, const mpo_align_params *p, const char *fq1, const char *fq2,
                        const char *out_fastq, const char *out_records, const char *bam_text_prefix, int print_mdnm,
                        const char *read_group, mpo_align_stats *stats);

/* stage-level access used by tests: seed + pair one batch of reads given as codes */
typedef struct {
    uint32_t read_id_left;
    uint64_t pos[2];
} mpo_candidate;

void mpo_default_params(mpo_align_params *p);
/* std::sort(.., LongerSeedLength()) of singleMerge (DV-DPfunctions.cpp:334) as libstdc++ 11 permutes it (test hook) */
void mpo_sort_longer_seed_perm(const uint32_t *seed_len, int n, int32_t *perm);

#ifdef __cplusplus
}
#endif
#endif
