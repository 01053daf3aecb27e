/*
 * include/megapath_hip.h -- the C ABI of the MI355X-native soap4 hot path (libmegapath_hip.so).
 *
 * The reference has no plugin/FFI surface: its boundary is the `soap4` process plus two internal
 * seams.  This header exports those seams as plain-C entry points (pointers and sizes only, no C++/torch types) so that
 * a soap4-compatible host front end -- ours in megapath_amd/csrc/host/, or the reference's own SOAP4.cpp with the
 * binding shown in INTEGRATION.md -- can drive the HIP kernels.  Every function returns 0 on success or a negative
 * MP_E* code; nothing throws across the boundary; all buffers named `out_*` are caller-owned.
 *
 * One mp_index per GPU (read-only, shared); one mp_ctx per host thread / HIP stream.
 */
#ifndef MEGAPATH_HIP_H
#define MEGAPATH_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_EINVAL (-22)   /* bad argument                                   */
#define MP_ENOMEM (-12)   /* host or device allocation failed               */
#define MP_EIO (-5)       /* index file missing / malformed                 */
#define MP_ENODEV (-19)   /* no HIP device / HIP runtime error              */
#define MP_ERANGE (-34)   /* input outside what the kernels support         */

typedef struct mp_index mp_index;
typedef struct mp_ctx mp_ctx;

/* ---------------------------------------------------------------------------------------------------------------
 * Index.  Replaces INDEXLoad (reference soap4/IndexHandler.cpp:49-137: BWTLoad 2bwt-lib/BWT.c:100, HSPLoad HSP.c:57,
 * LTLoad 2bwt-flex/LT.c:37).  Reads <prefix>.{bwt,fmv,sa,pac,ann,tra} (+ .lkt when present, else rebuilt from .pac as
 * LTConstruct.c:44-105 does) and re-lays them out in HBM for coalesced access (DESIGN.md "Data layout").
 * Text length must be < 2^32 - 2 (the reference itself needs index chunks < 4 Gbp: DV-DPfunctions.cpp:1516). */
int mp_index_open(const char *prefix, int device, mp_index **out);
/* Same, from arrays already in host memory (used by the in-repo index builder; layouts as in the index files):
 * bwt_words = 2-bit BWT, 16 symbols per word MSB-first, '$' omitted; sa_samples[k] = SA[k * sa_interval] with
 * sa_samples[0] ignored; pac_bytes = 4 bases per byte, first base in bits 7-6.  lkt may be NULL (rebuilt). */
int mp_index_from_arrays(int device, uint64_t text_length, uint64_t inverse_sa0, const uint64_t cum_freq[5],
                         const uint32_t *bwt_words, const uint64_t *sa_samples, uint32_t sa_interval,
                         const uint8_t *pac_bytes, const uint64_t *lkt,
                         uint32_t num_seq, const char *const *seq_names, const uint64_t *seq_starts, const uint64_t *seq_lengths,
                         mp_index **out);
/* A second copy of a resident index in the same GPU's HBM (device-to-device copies of every array; the annotation is duplicated):
 * how the benchmark fills the HBM with as many index chunks as the full NT database has (runMegaPath.sh:191-227 loops over them). */
int mp_index_clone(const mp_index *src, mp_index **out);
int mp_index_close(mp_index *ix);

typedef struct {
    uint64_t text_length, inverse_sa0;
    uint64_t cum_freq[5];
    uint32_t sa_interval, num_seq;
    uint64_t device_bytes;          /* HBM held by the index */
    uint32_t prefilter_k;           /* 17 when the k-mer presence bitmap of the seeding kernel is resident (2 GiB), else 0 */
    uint32_t reserved;
} mp_index_info;
int mp_index_get_info(const mp_index *ix, mp_index_info *out);
const char *mp_index_seq_name(const mp_index *ix, uint32_t seq_id_1based);
/* getChrAndPos (BGS-IO.cpp:163-188): packed position -> (1-based position in sequence, 1-based sequence id) */
int mp_index_chr_and_pos(const mp_index *ix, uint64_t amb_pos, uint64_t *out_pos, uint32_t *out_seq_id);
int mp_index_chr_and_pos_batch(const mp_index *ix, const uint64_t *amb_pos, uint64_t n, uint64_t *out_pos, uint32_t *out_seq_id);
/* Same, also returning getChrAndPos's own return value: the last packed position of the translate segment
 * (BGS-IO.cpp:186-187), which BoundaryCheckDP (BGS-IO.cpp:219) trims BAM records against. */
int mp_index_chr_pos_segment(const mp_index *ix, uint64_t amb_pos, uint64_t *out_pos, uint32_t *out_seq_id, uint64_t *out_segment_end);
/* decideTargetChr (BGS-IO.cpp:1312-1341): the 1-based sequence an alignment of `len` bases at amb_pos lies in; 0 when its first and last
 * base belong to different sequences (such an alignment is dropped by the annotated-FASTQ writers, BGS-IO.cpp:1406-1411, 1993-2001). */
int mp_index_target_chr(const mp_index *ix, uint64_t amb_pos, uint32_t len, uint32_t *out_seq_id);
/* HSP seqOffset[i].startPos / packed length (.ann) and seqActualOffset length (.tra tail = the BAM header's LN, SAM.cpp:117) */
int mp_index_seq_info(const mp_index *ix, uint32_t seq_id_1based, uint64_t *out_start, uint64_t *out_packed_len, uint64_t *out_actual_len);
/* Text bases (codes 0..3) of n_ranges ranges, gathered on the device from the resident .pac: range r lands in
 * out[r*stride .. r*stride + min(len[r], stride)).  Replaces the host-side hsp->packedDNA reads of getMisInfoForDP (PE.cpp:459). */
int mp_index_fetch_text(const mp_index *ix, const uint64_t *pos, const uint32_t *len, uint32_t n_ranges, uint8_t *out, uint32_t stride);

/* Unit-level device queries (tests; mirror BWTOccValue BWT.c:597, BWTSaValue BWT.c:968, LKT use at
 * DV-DPfunctions.cpp:2245-2246). All arrays are host pointers; n entries each. */
int mp_index_occ(const mp_index *ix, const uint64_t *idx, const uint32_t *sym, uint64_t n, uint64_t *out);
int mp_index_sa(const mp_index *ix, const uint64_t *idx, uint64_t n, uint64_t *out);
int mp_index_lkt(const mp_index *ix, const uint32_t *key, uint64_t n, uint64_t *out_l, uint64_t *out_r);
/* The resident 13-mer table (4^13 inclusive right SA bounds, LT.h:49-55) -- read from <prefix>.lkt or rebuilt on the device
 * at load time; lets a caller write the .lkt file the reference needs without recomputing it on the host. */
int mp_index_copy_lkt(const mp_index *ix, uint32_t *out);
/* The seeding kernel's pre-filter (mp_index_info.prefilter_k = 17): out[i] = 1 when backward search can reach length 17
 * with the 17-mer keys[i] (34 bits, first scanned base in the low bits as for mp_index_lkt), i.e. when the 13-mer table
 * range of its low 26 bits survives the four further extension steps.  MP_EINVAL when the bitmap is not resident. */
int mp_index_prefilter(const mp_index *ix, const uint64_t *keys, uint64_t n, uint8_t *out);
/* Test aid.  When the index carries the per-15-mer arrangement of the pre-filter bits (8.6 GB, one load answers three
 * consecutive probe positions): out[i] = the three answers for 17-mer key i (bit 0: as first, bit 1: as second, bit 2: as
 * third position of a group); each must equal mp_index_prefilter's bit.  MP_EINVAL when the table is not resident. */
int mp_index_prefilter3(mp_index *index, const uint64_t *keys, uint64_t n, uint8_t *out);
/* Test aid.  When the index carries the 16-mer look-up table (17 GB, built next to the pre-filter when HBM allows; the
 * seeding kernel then reads the SA range of a seed start's first 16 bases directly): compares the table with the index's
 * own procedure -- 13-mer table plus three backward-search steps -- for the 16-mer at every text position and for
 * n_random arbitrary keys.  MP_EINVAL when the table is not resident. */
int mp_index_check_lkt16(mp_index *index, uint64_t n_random, uint64_t *out_checked, uint64_t *out_mismatches);

/* ---------------------------------------------------------------------------------------------------------------
 * Context and reads. */
int mp_ctx_create(mp_index *ix, mp_ctx **out);
int mp_ctx_destroy(mp_ctx *ctx);
/* Binds the context to another index resident on the same GPU -- the next chunk of a database kept in HBM whole (the NT loop
 * of runMegaPath.sh:191-227 walks the chunks one soap4 process after the other).  The reads of the current batch stay set;
 * seeds and hit lists of the previous index are dropped.  MP_EINVAL if the index lives on another device. */
int mp_ctx_set_index(mp_ctx *ctx, mp_index *ix);

/* Reads of one batch, 2 per pair (read 2p, mate 2p+1), as the reference stores them after appendToQueryArrays
 * (QueryParser.cpp:184-234) except that each read's words are contiguous: base i of read r at bits 2*(i%16) of
 * word r*words_per_read + i/16 (codes A0 C1 G2 T3).  Lengths already truncated to maxReadLength-1. */
int mp_set_reads(mp_ctx *ctx, const uint32_t *packed, uint32_t words_per_read, const uint32_t *lengths, uint32_t n_reads);
/* Same, but the arrays are already in this GPU's memory (device pointers) -- no PCIe copy. */
int mp_set_reads_device(mp_ctx *ctx, const uint32_t *d_packed, uint32_t words_per_read, const uint32_t *d_lengths, uint32_t n_reads,
                        uint32_t max_length);

/* ---------------------------------------------------------------------------------------------------------------
 * Seeding.  Replaces PairEndSeedingBatch::mmpSeeding (DV-DPfunctions.cpp:2404-2615): mmp<0>/mmp<2> (:2224-2377),
 * SA locate (:2481-2495), merge/filter (:2499-2553).  Parameters = MmpProperties (PEAlgnmt.h:374-384, ini [MMP]). */
typedef struct {
    int32_t sa_size_threshold, seed_min_length, uniq_threshold, indel_fuzz, good_seed_len, reseed_len;
    double reseed_rlt_ratio;
    int32_t reseed_abs_diff;
    double short_seed_ratio;
} mp_mmp_params;

typedef struct {              /* DeepDP_Space::SeedPos (SeedPool.h:52-57) */
    uint64_t pos;
    uint32_t strand_read_id;  /* bit31 = reverse strand, low 31 = even read id of the pair */
    uint32_t paired_seed_length;
} mp_seedpos;

typedef struct {
    /* Event counts of the seed stage (K1 + K2), emitted by the kernels.  Only events that touch memory are counted here:
     *   n_occ        rank queries issued: 2 per backward-extension step carried out on the rank blocks, 1 per LF step of locate
     *   n_occ_loaded 64-byte rank blocks loaded for them (the two queries of a step share one block when the SA range is narrow)
     *   n_lkt        13-mer table look-ups (LT.h:49-55), n_lkt16: look-ups in the optional 16-mer table
     *   n_sa         suffix-array entries read (BWTSaValue, BWT.c:968-998; and the position of a single-row match)
     * With every optional table off (mp_ctx_set_prefilter(ctx, 0)) these are the reference algorithm's own counts. */
    uint64_t n_occ, n_lkt, n_sa, n_seeds, n_hits, n_seedpos;
    float ms_seed, ms_locate, ms_merge;                        /* HIP-event kernel times of this call */
    float reserved;
    uint64_t n_prefilter;                                      /* pre-filter probes answered (0 when the pre-filter is off) */
    uint64_t n_occ_loaded;
    uint64_t n_lkt16;
    uint64_t n_prefilter_loads;    /* table entries loaded for the probes (one 8-byte entry per 15-mer answers up to three) */
    uint64_t n_text_words;         /* 16-base text words compared in place of rank queries (single-row matches) */
    uint64_t n_steps_skipped;      /* extension steps never carried out: 3 per 16-mer look-up + 1 per base matched in the text;
                                    * n_occ + 2 * n_steps_skipped = rank queries of the same walk without those short cuts */
    uint64_t n_sa_seed_kernel;     /* the part of n_sa read by the seeding kernel itself (position of a single-row match); the rest, one per
                                    * hit that was not already reported by its text position, is read by locate -- or, when the whole suffix
                                    * array is resident, by the merge kernel, which then makes its keys straight from the seed records
                                    * (ms_locate = 0: no expand / locate launch) */
    uint64_t n_lf_steps;           /* LF steps of the locate kernel (one rank block each; they are part of n_occ and n_occ_loaded): what is left of
                                    * n_occ_loaded are the seeding kernel's own blocks */
} mp_seed_stats;

int mp_seed_batch(mp_ctx *ctx, const mp_mmp_params *params, mp_seed_stats *out_stats);
/* Measurement aid: on = 0 makes the seeding kernel ignore the index's 17-mer pre-filter, i.e. issue every table look-up
 * and rank query the reference algorithm issues (the event counts SURVEY 8d prices the stage with).  Same seeds either way. */
int mp_ctx_set_prefilter(mp_ctx *ctx, int on);
/* Copies the kept seeds of the last mp_seed_batch to the host: out_offsets has n_reads+1 entries; read r owns
 * out_seeds[out_offsets[r] .. out_offsets[r+1]), '+' strand seeds first then '-', each in ascending pos. */
int mp_get_seeds(mp_ctx *ctx, mp_seedpos *out_seeds, uint64_t cap, uint64_t *out_offsets);

/* ---------------------------------------------------------------------------------------------------------------
 * Semi-global DP + traceback.  Replaces SemiGlobalAligner::performAlignment -> callDP (CPU_DPfunctions.h:127-135,
 * CPU_DP.h:10-24, CPU_DP.cpp:122-978).  An instance names a read of the current batch, the strand it is aligned in
 * (1 = as is, 2 = reverse complement) and a window of the indexed text; the gather (packRead/repackDNA,
 * DV-DPfunctions.cpp:3009-3073) happens on the device from the resident reads and .pac. */
typedef struct {
    uint32_t read_id;
    uint32_t strand;
    uint64_t win_start;
    uint32_t win_len;
    uint32_t reserved;
} mp_dp_task;

typedef struct {
    int32_t score;            /* best score if >= cutoff and traceback valid, else 0 */
    uint32_t hit_loc;         /* 0-based start of the alignment inside the window     */
    uint32_t max_score_count; /* ties of the best score, saturating at 255            */
    uint32_t n_ops;           /* ops written (traceback order, alphabet "MmIDS")      */
} mp_dp_result;

/* cutoff = (int)max(0.2*len, 30) is applied per read as the reference hard-wires it (definitions.h:166-167).
 * out_ops may be NULL (scores/positions only); otherwise instance t owns out_ops[t*ops_stride ..). */
int mp_sgdp_batch(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n_tasks, int clip_left, int clip_right,
                  mp_dp_result *out_results, uint8_t *out_ops, uint32_t ops_stride, float *out_ms_fill, float *out_ms_trace);

/* ---------------------------------------------------------------------------------------------------------------
 * The deep-DP round for all pairs of the batch.  Replaces DPForUnalignPairs2 (DV-DPForBothUnalign.cpp:242) =
 * performMmpSeeding (:2617) + mergeAndPairPairedEnd (:2088) + PairEndAlignmentEngine::performAlignment (:3147). */
typedef struct {
    int32_t insert_low, insert_high;     /* after the SOAP4.cpp:465-474 adjustment */
    int32_t clip_left, clip_right;       /* ini MaxFrontLenClipped / MaxEndLenClipped */
    mp_mmp_params mmp;
    int32_t want_tie_counts;             /* 1: fill mp_pair_hit.count_* (num_sameScore, used only by the BAM MAPQ,
                                            BGS-IO.cpp:679-979); 0: MegaPath -F/-P text mode, counters left 0 */
    int32_t reserved;
} mp_align_params;

typedef struct {              /* one DeepDPAlignResult (PEAlgnmt.h:509-541), both mates passed the cutoff */
    uint32_t pair_id;
    uint32_t left_is_mate;    /* readIDLeft & 1 */
    uint64_t algnmt_1, algnmt_2;
    int32_t score_1, score_2;
    uint32_t start_1, start_2;        /* window starts (dpInfoForReads) */
    uint16_t dplen_1, dplen_2;
    uint16_t count_1, count_2;
} mp_pair_hit;

typedef struct {
    mp_seed_stats seed;
    uint64_t n_candidates, n_dp_left, n_dp_right, n_pair_hits;
    uint64_t dp_cells;
    float ms_pair, ms_dp_fill, ms_dp_trace, ms_total;
} mp_round_stats;

/* Runs seeding (unless already done for this batch), pairing and both DP legs on the device.  The hits are kept in the
 * context until the next call; *out_n_hits receives their number. */
int mp_deep_dp_round(mp_ctx *ctx, const mp_align_params *params, uint64_t *out_n_hits, mp_round_stats *out_stats);
/* Measurement / test aid.  The deep-DP legs of a round run a banded fill kernel that decides
 * every instance whose best alignment provably stays inside the band and leaves the rest to the full-table kernels; the
 * results are identical either way.  on = 0 sends every instance through the full-table kernels.  The counters report
 * how many instances went through the band kernel and how many of those were re-run, since the context was created. */
int mp_ctx_set_dp_band(mp_ctx *ctx, int on);
/* test hook: caller-given instances through the band scheme of the deep-DP round (results must equal mp_sgdp_batch's); MP_ERANGE where it does not apply */
int mp_test_dp_band(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n, int clip_left, int clip_right, mp_dp_result *out_results);
int mp_ctx_dp_band_stats(mp_ctx *ctx, uint64_t *out_n_band, uint64_t *out_n_full);
/* Measurement aid.  While on, every DP launch of the rounds also adds up the cells it computes (one small extra kernel per
 * launch, so leave it off in timed runs): `cells_full_table` = (window + 1) x read length of every instance submitted --
 * the work SURVEY 8d defines and the reference's callDP does -- next to what the kernels here really compute: band kernels
 * band-width x read length per pass (score pass, traceback pass), the full-table kernels all of it (score-only filter,
 * fill with traceback).  Instances the block-match pre-filter finishes as exact matches compute nothing.  Switching it on
 * clears the counters. */
typedef struct {
    uint64_t instances, cells_full_table;
    uint64_t cells_band_score, cells_band_trace, cells_full_score, cells_full_trace;
} mp_dp_census;
/* The certificate in front of the rescue filter (kernels_thermo.hip): instances that provably cannot reach their cutoff skip the filter.
 * 0 switches it off, 2 keeps it to the single-end windows of 101 .. 320-base reads (round 3's coverage), 1 (default) every shape it has a variant
 * for: reads of 30 .. 100 bases and the windows of the mate rescue too (test aid: the results are the same either way); stats: [0] certified,
 * [1] overflow, [2] bound too high, [3] not eligible. */
int mp_ctx_set_dp_thermo(mp_ctx *ctx, int on);
int mp_ctx_dp_thermo_stats(mp_ctx *ctx, uint64_t out[4]);
int mp_test_dp_thermo(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n, uint8_t *out_cert, uint64_t out_stats[4]);
/* the same for one given variant: kind 0 windows around a seed of reads of <= 100 bases, 1 of 101 .. 160, 2 of 161 .. 320, 3 mate-rescue windows (no seed, <= 768 rows) of reads
 * of <= 160 bases, 4 of <= 320; -1 = 1 or 2 by the longest read */
int mp_test_dp_thermo_kind(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n, int kind, uint8_t *out_cert, uint64_t out_stats[4]);
int mp_ctx_set_dp_census(mp_ctx *ctx, int on);
int mp_ctx_get_dp_census(mp_ctx *ctx, mp_dp_census *out);
/* The hit list is grouped by pair, pairs in ascending order (the hits of one pair in no particular order). */
int mp_get_pair_hits(mp_ctx *ctx, mp_pair_hit *out_hits, uint64_t cap);
/* Same without a pageable copy: the hits land in a pinned buffer owned by the context (valid until the next round).  The
 * transfer is started by mp_deep_dp_round itself on a copy stream; this call waits for it, so a mp_rescue_round issued in
 * between overlaps with it. */
int mp_get_pair_hits_view(mp_ctx *ctx, const mp_pair_hit **out_hits, uint64_t *out_n);

/* ---------------------------------------------------------------------------------------------------------------
 * Rescue passes for the pairs the deep-DP round left without a hit.  Replaces DPForUnalignSingle2
 * (DV-DPForSingleReads.cpp:321, alignment.cpp:182) and semiGlobalDPForSingleEndDpAlignment (DV-SemiDP.cpp:197,
 * alignment.cpp:250): candidate selection, window arithmetic and both DP batches run on the device; the per-pair
 * bookkeeping that turns the hits into output records stays with the caller.  Call after mp_deep_dp_round of the same batch
 * (MP_EINVAL otherwise: the pairs to rescue are the ones that round left without a hit). */
typedef struct {              /* SingleAlgnmtResult (PEAlgnmt.h) of a read whose pair failed the deep-DP round */
    uint32_t read_id;
    uint32_t strand;          /* 1 '+', 2 '-' */
    uint64_t algnmt;
    int32_t score;
    uint32_t count;           /* num_sameScore */
    uint64_t win_start;       /* startPos: orders the hits for mate rescue (DV-SemiDP.cpp:37-44) */
    uint32_t win_len;         /* refDpLength: with win_start it names the DP instance again (mp_sgdp_batch gives its ops) */
    uint32_t reserved;
} mp_single_hit;

typedef struct {              /* one successful mate rescue: the aligned mate's single hit + the DP hit of the other mate */
    uint32_t pair_id;
    uint32_t aligned_read;    /* read id (2p or 2p+1) of the mate that had the single-end hit */
    uint64_t aligned_pos, dp_pos;
    int32_t aligned_score, dp_score;
    uint32_t aligned_strand, dp_strand;
    uint64_t aligned_win_start, dp_win_start;   /* DP windows of both mates (startPos / refDpLength of the reference) */
    uint32_t aligned_win_len, dp_win_len;
    uint32_t aligned_count, dp_count;           /* num_sameScore */
} mp_rescue_hit;

typedef struct {
    uint64_t n_failed_pairs, n_single_tasks, n_single_hits, n_rescue_tasks, n_rescue_hits;
    float ms_tasks, ms_dp_fill, ms_dp_trace, ms_total;
    uint64_t n_floor_pairs, n_floor_tasks;    /* mp_ctx_set_score_floor: failed pairs whose rescue work in this chunk was left out, and their single-end instances */
    uint64_t n_floor_mate_tried, n_floor_mate_pairs;   /* of those: pairs kept by the first half of the rule for the rescued mate's length alone, and how many of them the bound on the mate's windows released */
} mp_rescue_stats;

int mp_rescue_round(mp_ctx *ctx, const mp_align_params *params, int skip_default_dp, mp_rescue_stats *out_stats);
/* The passing instances only, in task order (grouped by read, candidates in the order of singleMerge): they are compacted
 * on the device and copied to pinned host memory by mp_rescue_round itself, on a copy stream.  The _view calls wait for that
 * copy and hand out the context's buffers (valid until the next round or mp_set_reads*); the other two copy into
 * caller-owned arrays (out_hits may be NULL to ask for the count). */
int mp_get_single_hits_view(mp_ctx *ctx, const mp_single_hit **out_hits, uint64_t *out_n);
int mp_get_rescue_hits_view(mp_ctx *ctx, const mp_rescue_hit **out_hits, uint64_t *out_n);
int mp_get_single_hits(mp_ctx *ctx, mp_single_hit *out_hits, uint64_t cap, uint64_t *out_n);
int mp_get_rescue_hits(mp_ctx *ctx, mp_rescue_hit *out_hits, uint64_t cap, uint64_t *out_n);

/* Test aid.  The candidate order of single-end rescue: std::sort(.., LongerSeedLength()) of singleMerge (DV-DPfunctions.cpp:334)
 * exactly as the reference's libstdc++ permutes seeds of equal length -- the cut at 200 candidates per read
 * (DV-DPForSingleReads.cpp:200) makes that order output-visible.  Sorts n_arrays independent lists on the device
 * (list t = seed_len[off[t] .. off[t+1])); out_perm[off[t] + k] = index inside list t of the seed at rank k. */
int mp_test_sort_longer_seed(int device, const uint32_t *seed_len, const uint32_t *off, uint32_t n_arrays, uint32_t *out_perm);

/* Measurement aid: `lanes` lanes each load `per_lane` random 64-byte rank blocks (independent loads); reports the mean
 * launch time and the exact bytes one launch requests.  This is the random-64 B-gather ceiling the seed stage is priced
 * against and the calibration workload for the FETCH_SIZE counter. */
int mp_bench_gather64(const mp_index *ix, uint32_t lanes, uint32_t per_lane, uint32_t repeats, float *out_ms, uint64_t *out_bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * Coverage per reference and the spike filter.  Replaces the file pipeline of runMegaPath.sh:211-222:
 * samtools merge/sort | bedtools genomecov -bga | cc/genomeCovFilter (mean + k * stdev of the depth per sequence,
 * cc/genomeCovFilter.cpp:58-92) | bedtools bamtobed + annotate (fraction of each alignment on spike bases).  Intervals are
 * what the BAM side files hold: 1-based sequence id, 0-based start in the sequence's own coordinates, reference span. */
/* ---------------------------------------------------------------------------------------------------------------
 * Per-reference best scores of a batch, reduced on the device over every index chunk the batch is aligned against
 * (SURVEY 8f-2).  What the annotated-FASTQ writers keep of the hit lists is one score per mate and reference sequence
 * (pairDeepDPOutputFastqAPI BGS-IO.cpp:1966-2091, unproperlypairDPOutputFastqAPI :1384-1446, decideTargetChr :1312-1341,
 * normalizeScore :1943-1961), and the chained runs of the NT loop merge those per chunk through the header comments
 * (getMappingFromHeader :1348-1371, runMegaPath.sh:191-227).  Instead of three hit lists per chunk, the host gets ONE sorted
 * list per batch: a 64-bit key per (read, chunk, sequence),
 *     read id << 42 | (n_chunks - 1 - chunk) << 32 | 1-based sequence id << 10 | (1023 - score)
 * so that a read's entries come in the order the final comment lists them (last chunk first, sequences ascending).
 *   mp_reduce_begin(ctx, n_chunks)     after mp_set_reads
 *   per chunk: mp_ctx_set_index, mp_deep_dp_round, mp_rescue_round (when pairs are left), mp_reduce_chunk(ctx, mode, chunk)
 *   mp_reduce_finish(ctx, &keys, &n)   keys in a pinned buffer of the context, valid until the next mp_reduce_begin
 * megapath_mode: 1 = -F, 2 = -P (IniParam.cpp:879-884).  Limits: 2^22 reads per batch, 1024 chunks, 2^22 - 1 sequences per chunk,
 * scores < 1024 (MP_ERANGE otherwise: use the hit lists). */
typedef struct { uint64_t n_deep_pairs, n_rescued_pairs, n_unpaired_pairs, n_entries; } mp_reduce_stats;
int mp_ctx_set_hit_copies(mp_ctx *ctx, int on);     /* 0: the rounds leave their hit lists on the device (only the reduced list is wanted) */
int mp_reduce_begin(mp_ctx *ctx, uint32_t n_chunks);
/* out_pair_class (may be NULL): per pair of the batch how THIS chunk aligned it -- 0 deep-DP round, 1 mate rescue, 2 neither: the order
 * the reference writes a batch in (alignment.cpp:105-330); the caller's array of n_reads / 2 bytes is filled when the call returns. */
int mp_reduce_chunk(mp_ctx *ctx, int megapath_mode, uint32_t chunk_index, mp_reduce_stats *out_stats, uint8_t *out_pair_class);
/* The -top rule across resident chunks (kernels_floor.hip; getMappingFromHeader BGS-IO.cpp:1348-1371 keeps what reaches top% of a read's best score over ALL
 * chunks): with every chunk resident a batch goes through them in two passes --
 *   pass 1, chunks 0 .. n-2:  mp_ctx_set_index, mp_deep_dp_round, mp_reduce_chunk_part(.., 1, ..) [the pair hits; raises the best score known per read], mp_chunk_stash(c)
 *   the last chunk as ever:   mp_ctx_set_index, mp_deep_dp_round, mp_rescue_round, mp_reduce_chunk (its classes are the sections of the output; its lists feed -b)
 *   pass 2, chunks 0 .. n-2:  mp_ctx_set_index, mp_chunk_unstash(c), mp_ctx_set_score_floor(top), mp_rescue_round, mp_reduce_chunk_part(.., 2, ..)
 * -- and mp_rescue_round leaves out the rescue work of pairs for which NOTHING it could produce in that chunk can reach top% of the best score already known
 * (decided per pair from a sound upper bound on every single-end instance; the results that remain are the reference's own, the entries left out are ones the
 * printer's final filter drops anyway).  mp_reduce_finish's list is the same with the floor on and off. */
int mp_reduce_chunk_part(mp_ctx *ctx, int megapath_mode, uint32_t chunk_index, int parts /* 1 pair hits, 2 rescue round's hits + counts + classes, 3 both */,
                         mp_reduce_stats *out_stats, uint8_t *out_pair_class);
int mp_ctx_set_score_floor(mp_ctx *ctx, int top_percent);      /* 0 = off (default): every rescue round does all its work */
int mp_test_floor_bound(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n, uint16_t *out_ub);   /* test hook: the rule's upper bound per instance (>= its exact DP score) */
int mp_test_floor_mate_bound(mp_ctx *ctx, const mp_dp_task *tasks, uint32_t n, int insert_high, uint16_t *out_ub);   /* test hook: the bound on the mate rescued from a hit inside instance t's window (tasks grouped by read, ascending; reads 2k and 2k + 1 are mates) */
int mp_chunk_stash(mp_ctx *ctx, uint32_t slot);               /* after mp_deep_dp_round: keep what the chunk's rescue round needs (seed positions, aligned pairs) */
int mp_chunk_unstash(mp_ctx *ctx, uint32_t slot);             /* after mp_ctx_set_index of the same chunk: put it back; mp_rescue_round may follow */
int mp_reduce_finish(mp_ctx *ctx, const uint64_t **out_keys, uint64_t *out_n);

typedef struct mp_coverage mp_coverage;
int mp_coverage_create(const mp_index *ix, mp_coverage **out);
int mp_coverage_destroy(mp_coverage *cov);
int mp_coverage_add(mp_coverage *cov, const uint32_t *seq_id_1based, const uint64_t *start0, const uint32_t *ref_span, uint64_t n);
/* depth = prefix sum of the events; per sequence mean and population variance of the depth over all its bases;
 * max_depth = mean + stdev_k * sqrt(variance); spike bases = depth > max_depth.  No more intervals can be added afterwards. */
int mp_coverage_finish(mp_coverage *cov, double stdev_k);
int mp_coverage_seq_stats(const mp_coverage *cov, uint32_t seq_id_1based, double *out_mean, double *out_variance, double *out_max_depth);
/* the maximal runs of spike bases (the union of the lines cc/genomeCovFilter prints); out_* may be NULL to ask for the count */
int mp_coverage_spike_intervals(mp_coverage *cov, uint32_t *out_seq_id, uint64_t *out_start0, uint64_t *out_end0, uint64_t cap, uint64_t *out_n);
/* per interval the number of its bases that are spike bases (bedtools annotate's numerator) */
int mp_coverage_spike_bases(mp_coverage *cov, const uint32_t *seq_id_1based, const uint64_t *start0, const uint32_t *ref_span, uint64_t n, uint32_t *out_covered);

/* ---------------------------------------------------------------------------------------------------------------
 * Index construction.  Replaces the compute of `2bwt-builder` (2BWT-Builder.c:199-393: BWTConstruct.c builds
 * the BWT incrementally on one core, then the sampled suffix array by inverse-psi walks): from the n base codes (0..3, host) of the
 * text, the arrays the reference's files hold -- BWT words with the '$' row left out (16 symbols per word, first in the top bits:
 * .bwt), that row's number (inverseSa0), cumulative symbol counts, SA[k * sa_interval] for k = 0 .. n / sa_interval (.sa; SA[0] = n),
 * and the packed text (.pac: 4 bases per byte, first in bits 7-6; (n + 3) / 4 bytes).  Suffix array by prefix doubling in HBM
 * (33 bytes per base while it runs); n < 2^32 - 3.  out_rounds / out_ms_suffix_sort may be NULL. */
int mp_build_index_arrays(int device, const uint8_t *codes, uint64_t n, uint32_t sa_interval, uint32_t *out_bwt_words, uint64_t *out_sa_samples,
                          uint8_t *out_pac, uint64_t *out_inv_sa0, uint64_t *out_cum /* [5] */, uint32_t *out_rounds, float *out_ms_suffix_sort);

/* Library / device probe: fails loudly (MP_ENODEV) when no gfx950 device is usable. */
int mp_device_check(int device, char *out_name, size_t name_cap);
int mp_device_memory(int device, uint64_t *out_free, uint64_t *out_total);   /* bytes of the GPU's memory free / in all, right now */
const char *mp_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
