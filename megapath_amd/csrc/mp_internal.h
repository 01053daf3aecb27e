// megapath_amd/csrc/mp_internal.h -- shared host/device definitions of libmegapath_hip (product code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <mutex>
#include <memory>
#include "../../include/megapath_hip.h"

// ------------------------------------------------------------------------------------------------ HBM index layout
// One 64-byte block per 192 BWT symbols: running counts of A,C,G,T before the block, then the symbols themselves as two bit
// planes (symbol s of the block: its low bit at bit s%32 of lo[s/32], its high bit at the same place of hi).  One rank query =
// one aligned 64 B load; with the planes the matches of a symbol are one dense mask per 32 symbols (two logic operations), their
// running popcounts six more, and a rank is a select and one masked popcount -- a quarter of the instructions the 2-bit-per-symbol
// packing needs.  Counts are 32-bit: a text is < 2^32-2 bases (the reference's own per-chunk limit).
#define MP_OCC_SYMS 192u
struct mp_occ_block {
    uint32_t cnt[4];
    uint32_t lo[6];
    uint32_t hi[6];
};
static_assert(sizeof(mp_occ_block) == 64, "occ block must be one 64-byte sector");

#define MP_LKT_K 13
#define MP_LKT_ENTRIES (1u << (2 * MP_LKT_K))

struct mp_dev_index {          // passed by value to kernels
    const mp_occ_block *occ;   // ceil((n+2)/192)+1 blocks
    const uint32_t *lkt;       // 4^13 inclusive right SA bounds (nullptr when lkt14 stands for it)
    const uint2 *lkt14;        // optional, IN PLACE of lkt (MP_LKT14=1; the lean layout of many resident chunks): 4^13 + 1 entries of 8 bytes, one per 13-mer Q =
                               // the first 13 text bases of four 14-mers: {x = lkt[Q - 1] (0 for Q = 0; entry 4^13: lkt[4^13 - 1]), y = how many suffixes
                               // continue Q with A, C, G, T, one byte each}: the SA range of the 14-mer Q.d is x + (counts below d) + 1 .. + count of d, i.e.
                               // the 13-mer look-up AND the first extension step of a seed in one 8-byte load (lkt14_build_kernel makes the counts by that
                               // very step); a byte of 255 = no short cut for this entry (a count that does not fit, or an entry whose step does not
                               // begin at x + 1): the 13-mer path, which reads lkt[k] as entry[k + 1].x
    const uint32_t *sa;        // SA[k*interval]; entry 0 unused (handled as -1, BWT.c:242)
    const uint32_t *text;      // 2-bit text, LSB-first, 16 bases per word
    const uint32_t *lkt16;     // optional: 4^16 inclusive right SA bounds of the 16-mers (nullptr = off); used behind the 17-mer filter
    const uint4 *lkt16u;       // optional: per 16-mer one 16-byte entry {l, r, pos, prev16} (69 GB): the SA range, and for a 16-mer that occurs once
                               // its text position and the 16 text bases in front of it (scan order) -- a seed start then costs ONE sector instead of
                               // three (16-mer table, suffix array, text); pos = 0xFFFFFFFF: no short cut (match within 16 bases of the text start)
    const unsigned long long *kmer15x3;   // optional: per 15-mer, the presence bits of the 48 17-mers around it (three alignments, see
                               // kmer15x3_build_kernel): one load answers the pre-filter for three consecutive scan positions
    const uint32_t *kmer17;    // optional: one bit per K-mer (K = kmer_k: 17, or 16 for 512 MiB instead of 2 GiB), set when backward search can reach length K with it (nullptr = off)
    uint32_t kmer_k;
    const uint32_t *sa_dense;  // optional: SA[k << sa_dense_shift], filled on the device from the file's samples (nullptr = off)
    uint32_t sa_dense_shift;   // 0: the whole suffix array, 1: every 2nd row, 2: every 4th
    uint32_t n;                // text length
    uint32_t inv_sa0;
    uint32_t cum[5];
    uint32_t sa_interval;
    uint32_t sa_shift;         // log2(sa_interval) when it is a power of two (the builder default 16), else 0xFFFFFFFF
};

struct mp_translate_row { uint64_t start; uint32_t chr; uint64_t corr; };

struct mp_index {
    int device = 0;
    mp_dev_index d{};
    void *d_lkt14 = nullptr;
    void *d_occ = nullptr, *d_lkt = nullptr, *d_sa = nullptr, *d_text = nullptr, *d_kmer = nullptr, *d_sa_dense = nullptr, *d_lkt16 = nullptr, *d_kmer15 = nullptr, *d_lkt16u = nullptr;
    void *d_tr_grid = nullptr, *d_tr_start = nullptr, *d_tr_chr = nullptr;      // the .tra table on the device (kernels_reduce.hip), uploaded on first use
    uint64_t device_bytes = 0;
    size_t bytes_occ = 0, bytes_lkt = 0, bytes_sa = 0, bytes_text = 0, bytes_kmer = 0, bytes_sa_dense = 0, bytes_lkt16 = 0, bytes_kmer15 = 0, bytes_lkt16u = 0, bytes_lkt14 = 0;   // per array (mp_index_clone)
    uint64_t n = 0, inv_sa0 = 0, cum[5] = {0, 0, 0, 0, 0};
    uint32_t sa_interval = 0;
    std::vector<std::string> seq_names;
    std::vector<uint64_t> seq_start, seq_len;
    std::vector<uint64_t> seq_actual_len;   // .tra tail: length including removed N-runs (SAM header LN)
    uint64_t n_removed = 0;                 // .tra header: numOfRemovedSegment
    std::vector<uint32_t> grid;
    std::vector<mp_translate_row> translate;
    // host mirror of the 2-bit text, copied back from the device once, the first time somebody asks for reference bases
    // (mp_index_fetch_text: MD / NM tags of -b -p); shared_ptr so that the struct stays copyable (mp_index_clone)
    struct HostText { std::mutex mu; std::vector<uint32_t> words; bool ready = false; };
    std::shared_ptr<HostText> host_text = std::make_shared<HostText>();
};

// host-side staging of an index before upload
struct mp_host_index {
    uint64_t n = 0, inv_sa0 = 0, cum[5] = {0, 0, 0, 0, 0};
    uint32_t sa_interval = 0;
    // raw arrays in the reference's file layouts; the HBM layouts (rank blocks, LSB-first text, LKT when no .lkt file
    // is present) are built on the device by upload_index (api.hip)
    std::vector<uint32_t> bwt_words;   // 16 symbols per word, MSB first, '$' omitted
    std::vector<uint8_t> pac;          // 4 bases per byte, first base in bits 7-6
    std::vector<uint32_t> lkt;         // 4^13 inclusive right bounds, or empty -> rebuilt from the text
    std::vector<uint32_t> sa;
    std::vector<std::string> seq_names;
    std::vector<uint64_t> seq_start, seq_len;
    std::vector<uint64_t> seq_actual_len;
    uint64_t n_removed = 0;
    std::vector<uint32_t> grid;
    std::vector<mp_translate_row> translate;
};

// index_host.cpp
int mp_host_index_load(const char *prefix, mp_host_index &out, std::string &err);
int mp_host_index_from_arrays(uint64_t n, uint64_t inv_sa0, const uint64_t cum[5], const uint32_t *bwt_words,
                              const uint64_t *sa_samples, uint32_t sa_interval, const uint8_t *pac, const uint64_t *lkt,
                              uint32_t num_seq, const char *const *names, const uint64_t *starts, const uint64_t *lens,
                              mp_host_index &out, std::string &err);

// ------------------------------------------------------------------------------------------------ seeding records
struct mp_sa_seed {            // SeedSAalign (DV-DPfunctions.cpp:2162-2172) + owner
    uint32_t read_strand;      // read_id*2 + strand
    uint32_t sa_l;
    uint16_t query_offset;
    uint16_t seed_len;
    uint32_t sa_diff;
};

struct mp_seed_hit {           // SeedAlign (DV-DPfunctions.cpp:2144-2160), strand folded into the sort key
    uint64_t offset;
    uint16_t mult, length, qo, strand;
};

struct mp_dev_mmp {
    int32_t sa_size_threshold, seed_min_length, uniq_threshold, indel_fuzz, good_seed_len, reseed_len, reseed_abs_diff;
    double reseed_rlt_ratio, short_seed_ratio;
};

struct mp_counters {           // device-side event counters
    unsigned long long n_occ, n_lkt, n_sa, n_seeds, n_hits, n_seedpos, n_cand, overflow, n_kmer, n_occ_loaded;
    unsigned long long n_lkt16, n_kmer_loads, n_text_words, n_steps_skipped, n_located, n_wave_trips, n_lane_trips, n_lf;
};

struct mp_hit_task { uint32_t seed_slot; uint32_t k; };                         // one SA row to locate
struct mp_candidate { uint32_t read_id_left; uint32_t pad; uint64_t pos0, pos1; }; // CandidateInfo (DV-DPfunctions.h:1233-1241)
struct mp_fill_result { int32_t score; uint16_t row, col; uint32_t count; };    // best cell of one DP instance
