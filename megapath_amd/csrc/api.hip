// megapath_amd/csrc/api.hip -- product code: the C ABI of libmegapath_hip (include/megapath_hip.h), HBM residency and
// stage orchestration.  One translation unit: the kernels are #included so that hipcc sees a single code object.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <new>
#include "mp_internal.h"
#include "kernels_index.hip"
#include "kernels_seed.hip"
#include "kernels_dp.hip"
#include "kernels_dp_band.hip"
#include "kernels_rescue.hip"
#include "kernels_thermo.hip"
#include "kernels_cov.hip"
#include "kernels_reduce.hip"
#include "kernels_floor.hip"
#include "kernels_build.hip"

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fprintf(stderr, "[megapath_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return MP_ENODEV; } } while (0)

// MP_TRACE_KERNELS=1: after every kernel launch of the rounds, wait for it and name it on stderr -- a faulting kernel is then
// the last one named (debugging aid; serialises everything)
static const bool g_trace_kernels = getenv("MP_TRACE_KERNELS") != nullptr;
#define MP_TRACE(stream, what) do { if (g_trace_kernels) { hipError_t te_ = hipStreamSynchronize(stream); \
    fprintf(stderr, "[megapath_hip] %s: %s\n", what, te_ == hipSuccess ? "ok" : hipGetErrorString(te_)); fflush(stderr); } } while (0)

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;       // elements
    int reserve(size_t n)
    {
        if (n <= cap) return MP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) { p = nullptr; return MP_ENOMEM; }
        cap = want;
        return MP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

enum { CENSUS_INSTANCES = 0, CENSUS_FULL_EQUIV = 1, CENSUS_BAND_SCORE = 2, CENSUS_BAND_TRACE = 3, CENSUS_FULL_SCORE = 4, CENSUS_FULL_TRACE = 5, CENSUS_SLOTS = 6 };

inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

struct mp_ctx {
    mp_index *ix = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // device-to-host copy of the pair hits, overlapping whatever runs next on `stream`
    hipEvent_t ev_hits_ready = nullptr, ev_hits_copied = nullptr;
    bool hits_copy_started = false;
    hipEvent_t ev[10] = {};
    std::vector<hipEvent_t> ev_pool;       // per-chunk DP timing without per-chunk host syncs
    // reads
    DevBuf<uint32_t> reads, lens, scan;
    const uint32_t *d_reads = nullptr, *d_lens = nullptr;   // either our buffers or caller-owned device memory
    uint32_t n_reads = 0, W = 0, max_len = 0, min_len = 0;      // (min_len 0: not known -- reads handed over in device memory)
    size_t slab_budget = 0;          // bytes of traceback flags per DP launch (decided at the first DP call)
    bool use_prefilter = true;       // mp_ctx_set_prefilter
    bool use_band = true;            // mp_ctx_set_dp_band
    int use_thermo = 1;              // mp_ctx_set_dp_thermo: 0 off, 1 every shape it has a variant for, 2 the single-end windows of 101 .. 320-base reads only (round 3)
    bool seeded = false;
    // seeding
    DevBuf<mp_seed_rec> slabs, big_slabs;
    DevBuf<mp_sa_seed> seeds;             // one per located hit
    DevBuf<uint32_t> hit_row, seeds_per_rs, hits_per_rs, rs_base, overflow_rs, over_slot, n_pos, n_neg;
    DevBuf<unsigned int> queue;
    DevBuf<mp_seed_hit> hits;
    DevBuf<mp_seedpos> seedpos;
    DevBuf<uint8_t> cub_tmp;
    DevBuf<mp_counters> ctr;
    mp_seed_stats last_seed{};
    uint64_t n_hits_total = 0;
    // pairing + DP
    DevBuf<mp_candidate> cands;
    DevBuf<mp_dp_task> tasks_l, tasks_r, tasks_tmp;
    DevBuf<mp_fill_result> fill;
    DevBuf<mp_dp_result> res_l, res_r, res_tmp;
    DevBuf<uint8_t> flags, ops;
    DevBuf<uint8_t> th_cert;                     // dp_thermo_kernel: instances certified unable to reach their cutoff
    DevBuf<unsigned long long> th_stats;
    unsigned long long th_total[4] = {0, 0, 0, 0};
    int th_rescue_skip = 0;                      // calls the mate-rescue variant of the certificate still sits out (run_dp_filtered)
    DevBuf<uint32_t> sel_flag, sel_idx;          // score-only pre-pass of the rescue DP batches
    DevBuf<mp_dp_task> sel_tasks;
    DevBuf<mp_dp_result> sel_res;
    DevBuf<uint32_t> band_flag, band_idx;        // instances the band kernel left to the full kernels
    DevBuf<uint32_t> near_flag, near_idx, near_list, far_flag, far_idx, far_list;   // classes of the second band pass
    DevBuf<uint32_t> pre_cls, narrow_list, wide_list, narrow15_list, score_list;                               // dp_prefilter_kernel's classes: first band pass over 27 / 87 diagonals
    DevBuf<uint32_t> mid_flag, mid_idx, mid_list;                                   // scores L-9 .. L-6: second pass over 15 diagonals
    DevBuf<uint32_t> vfar_flag, vfar_idx, vfar_list, und_list[2];                   // second tier; instances still undecided after a tier
    DevBuf<mp_dp_task> band_tasks;
    DevBuf<mp_dp_result> band_res;
    DevBuf<mp_fill_result> band_fill;
    uint64_t n_band = 0, n_band_full = 0;        // instances run through the band kernel / re-run by the full kernels
    DevBuf<mp_pair_hit> pair_hits;
    DevBuf<uint32_t> pair_cnt, pair_off;   // hits per pair and where each pair's group starts in pair_hits
    DevBuf<uint32_t> hit_flag, hit_idx, cand_first;   // per candidate: makes a hit / hits in front of it; per pair: its first candidate
    DevBuf<unsigned long long> n_pair_hits;
    uint64_t n_pair_hits_host = 0;
    mp_pair_hit *h_pair_hits = nullptr;    // pinned staging for mp_get_pair_hits_view
    size_t h_pair_hits_cap = 0;
    // rescue passes
    DevBuf<uint32_t> pair_done, failed_flag, failed_off, failed_reads, n_scand, s_off, rsrc;
    DevBuf<mp_single_cand> scand;
    DevBuf<mp_dp_task> stasks, rtasks, rtasks_c;     // rtasks_c / rres_c / rlist: the filled mate-rescue slots, gathered
    DevBuf<mp_dp_result> sres, rres, rres_c;
    DevBuf<uint32_t> rlist;
    uint32_t n_failed_reads = 0, n_stasks = 0;
    bool rescued = false;
    bool census_on = false;                      // mp_ctx_set_dp_census: count the cells each DP kernel family computes
    DevBuf<unsigned long long> census;           // [instances, full-table cells, band score, band trace, full score, full trace]
    bool deep_done = false;                      // mp_deep_dp_round ran for the reads now set (mp_rescue_round needs its hits)
    hipEvent_t ev_round[3] = {};                 // whole-call timing of mp_deep_dp_round / mp_rescue_round
    // passing instances of the rescue passes, compacted on the device and copied to pinned memory behind the compute stream
    DevBuf<uint32_t> sh_flag, sh_slot, rh_flag, rh_slot;
    DevBuf<mp_single_hit> d_single_hits;
    DevBuf<mp_rescue_hit> d_rescue_hits;
    mp_single_hit *h_single_hits = nullptr;
    mp_rescue_hit *h_rescue_hits = nullptr;
    size_t h_single_cap = 0, h_rescue_cap = 0;
    uint64_t n_single_hits_host = 0, n_rescue_hits_host = 0;
    hipEvent_t ev_rescue_ready = nullptr, ev_rescue_copied = nullptr;
    bool rescue_copy_started = false;
    // per-reference best scores of the batch, accumulated over the chunks it is aligned against (mp_reduce_*)
    unsigned long long *red_keys = nullptr, *red_sorted = nullptr, *red_out = nullptr;
    size_t red_cap = 0, red_sorted_cap = 0, red_out_cap = 0, red_n = 0;
    DevBuf<uint32_t> resc_done, red_flag, red_slot;
    DevBuf<uint8_t> red_cls;
    DevBuf<unsigned long long> red_counts;
    unsigned long long *h_red = nullptr;       // pinned
    size_t h_red_cap = 0;
    uint32_t red_chunks = 0;
    bool copy_hits = true;                     // mp_ctx_set_hit_copies: the rounds send their hit lists to pinned host memory
    // the -top rule across resident chunks (kernels_floor.hip): best score known per read (raised by every mp_reduce_chunk call), the top percentage
    // (0 = the rescue rounds do all their work), what mp_chunk_stash keeps of a chunk between its deep-DP round and its rescue round
    DevBuf<uint32_t> best_known;
    int floor_top = 0;
    DevBuf<uint16_t> floor_ub;
    DevBuf<unsigned long long> floor_counts;
    DevBuf<uint8_t> floor_hope;                 // [failed pair] kept by floor_decide_kernel for the mate's length alone
    struct Stash { size_t off = 0, bytes = 0; uint64_t n_hits = 0; uint32_t n_reads = 0; const mp_index *ix = nullptr; mp_seed_stats seed{}; bool used = false, packed = false; };
    std::vector<Stash> stash;
    uint8_t *stash_arena = nullptr;
    size_t stash_cap = 0, stash_used = 0;
};

// ------------------------------------------------------------------------------------------------ misc
extern "C" const char *mp_strerror(int code)
{
    switch (code) {
        case MP_OK: return "ok";
        case MP_EINVAL: return "invalid argument";
        case MP_ENOMEM: return "out of host or device memory";
        case MP_EIO: return "index file missing or malformed";
        case MP_ENODEV: return "no usable HIP device / HIP runtime error";
        case MP_ERANGE: return "input outside the supported range";
        default: return "unknown error";
    }
}

// ------------------------------------------------------------------------------------------------ index construction
extern "C" int mp_build_index_arrays(int device, const uint8_t *codes, uint64_t n, uint32_t sa_interval, uint32_t *out_bwt_words, uint64_t *out_sa_samples,
                                     uint8_t *out_pac, uint64_t *out_inv_sa0, uint64_t *out_cum, uint32_t *out_rounds, float *out_ms_suffix_sort)
{
    if (!codes || !out_bwt_words || !out_sa_samples || !out_pac || !out_inv_sa0 || !out_cum || sa_interval == 0) return MP_EINVAL;
    if (n < 1 || n + 3 >= (1ull << 32)) return MP_ERANGE;                       // an index chunk holds < 2^32 - 3 bases (the reference's limit)
    HIPCHK(hipSetDevice(device));
    const uint64_t m = n + 1;                                                   // suffixes 0 .. n, suffix n = '$'
    DevBuf<uint8_t> d_codes, d_tmp, d_pac;
    DevBuf<uint64_t> key_a, key_b;
    DevBuf<uint32_t> idx_a, idx_b, rank, row_rank, d_words;
    DevBuf<unsigned long long> d_small, d_samples;
    struct Free { DevBuf<uint8_t> &a, &b, &c; DevBuf<uint64_t> &d, &e; DevBuf<uint32_t> &f, &g, &h, &i, &j; DevBuf<unsigned long long> &k, &l;
                  ~Free() { a.release(); b.release(); c.release(); d.release(); e.release(); f.release(); g.release(); h.release(); i.release(); j.release(); k.release(); l.release(); } }
        guard{d_codes, d_tmp, d_pac, key_a, key_b, idx_a, idx_b, rank, row_rank, d_words, d_small, d_samples};
    auto exact = [](auto &buf, size_t count) -> int {                           // no growth margin: these are tens of GB
        using T = typename std::remove_reference<decltype(*buf.p)>::type;
        if (hipMalloc((void **)&buf.p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { buf.p = nullptr; return MP_ENOMEM; }
        buf.cap = count;
        return MP_OK;
    };
    if (exact(d_codes, n) || exact(key_a, m) || exact(key_b, m) || exact(idx_a, m) || exact(idx_b, m) || exact(rank, m) || exact(row_rank, m) || exact(d_small, 8))
        return MP_ENOMEM;
    hipStream_t s = nullptr;
    HIPCHK(hipMemcpy(d_codes.p, codes, n, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(d_small.p, 0, 8 * sizeof(unsigned long long), s));
    struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
    HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    HIPCHK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(build_init_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, (const uint8_t *)d_codes.p, n, key_a.p, idx_a.p);
    hipcub::DoubleBuffer<uint64_t> keys(key_a.p, key_b.p);
    hipcub::DoubleBuffer<uint32_t> vals(idx_a.p, idx_b.p);
    int rank_bits = 1;
    while ((1ull << rank_bits) <= m) rank_bits++;
    auto sort_pairs = [&](int end_bit) -> int {
        size_t bytes = 0;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, vals, m, 0, end_bit, s) != hipSuccess) return MP_ENODEV;
        if (bytes > d_tmp.cap) { d_tmp.release(); if (exact(d_tmp, bytes + 256)) return MP_ENOMEM; }
        if (hipcub::DeviceRadixSort::SortPairs(d_tmp.p, bytes, keys, vals, m, 0, end_bit, s) != hipSuccess) return MP_ENODEV;
        return MP_OK;
    };
    int rc = sort_pairs(3 * MP_BUILD_K);
    if (rc) return rc;
    uint32_t rounds = 0;
    for (uint64_t h = MP_BUILD_K;; h *= 2) {
        rounds++;
        hipLaunchKernelGGL(build_flag_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, (const uint64_t *)keys.Current(), m, row_rank.p);
        {
            size_t bytes = 0;
            if (hipcub::DeviceScan::InclusiveSum(nullptr, bytes, row_rank.p, row_rank.p, m, s) != hipSuccess) return MP_ENODEV;
            if (bytes > d_tmp.cap) { d_tmp.release(); if (exact(d_tmp, bytes + 256)) return MP_ENOMEM; }
            if (hipcub::DeviceScan::InclusiveSum(d_tmp.p, bytes, row_rank.p, row_rank.p, m, s) != hipSuccess) return MP_ENODEV;
        }
        uint32_t top = 0;
        HIPCHK(hipMemcpyAsync(&top, row_rank.p + (m - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if ((uint64_t)top == m - 1) break;                                      // all ranks differ: vals is the suffix array
        if (h >= m) return MP_ENODEV;                                             // cannot happen: suffixes differ within n + 1 symbols
        hipLaunchKernelGGL(build_scatter_rank_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, (const uint32_t *)vals.Current(), (const uint32_t *)row_rank.p, m, rank.p);
        hipLaunchKernelGGL(build_next_key_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, (const uint32_t *)vals.Current(), (const uint32_t *)row_rank.p,
                           (const uint32_t *)rank.p, m, h, keys.Current());
        rc = sort_pairs(32 + rank_bits);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(e1, s));
    const uint32_t *sa = vals.Current();
    // ---- what the builder writes: BWT words, '$' row, counts, sampled suffix array, packed text
    key_a.release(); key_b.release(); rank.release(); row_rank.release();      // (the suffix array lives in one of the idx buffers)
    const uint64_t n_words = (n + 15) / 16, n_pac = (n + 3) / 4, n_samples = (m + sa_interval - 1) / sa_interval;
    if (exact(d_words, n_words) || exact(d_pac, n_pac) || exact(d_samples, n_samples)) return MP_ENOMEM;
    hipLaunchKernelGGL(build_find_row0_kernel, dim3(grid_for(m, 256)), dim3(256), 0, s, sa, m, d_small.p + 4);
    unsigned long long small[8];
    HIPCHK(hipMemcpyAsync(small, d_small.p, sizeof small, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint64_t inv_sa0 = small[4];
    hipLaunchKernelGGL(build_bwt_words_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, s, (const uint8_t *)d_codes.p, sa, n, inv_sa0, d_words.p);
    hipLaunchKernelGGL(build_sa_samples_kernel, dim3(grid_for(n_samples, 256)), dim3(256), 0, s, sa, m, sa_interval, n_samples, d_samples.p);
    hipLaunchKernelGGL(build_pac_kernel, dim3(grid_for(n_pac, 256)), dim3(256), 0, s, (const uint8_t *)d_codes.p, n, d_pac.p, d_small.p);
    HIPCHK(hipGetLastError());                                                  // a launch that failed (configuration, resources) must not pass for an index
    HIPCHK(hipMemcpyAsync(out_bwt_words, d_words.p, n_words * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_sa_samples, d_samples.p, n_samples * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_pac, d_pac.p, n_pac, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(small, d_small.p, sizeof small, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *out_inv_sa0 = inv_sa0;
    out_cum[0] = 0;
    for (int c = 0; c < 4; c++) out_cum[c + 1] = out_cum[c] + small[c];
    if (out_rounds) *out_rounds = rounds;
    if (out_ms_suffix_sort) (void)hipEventElapsedTime(out_ms_suffix_sort, e0, e1);
    return MP_OK;
}

extern "C" int mp_device_check(int device, char *out_name, size_t name_cap)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device || device < 0) return MP_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MP_ENODEV;
    if (out_name && name_cap) snprintf(out_name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MP_ENODEV;   // this library carries gfx950 code only
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ index
// Uploads the raw index arrays and builds the HBM layouts with streaming kernels (kernels_index.hip).
static int upload_index(int device, mp_host_index &h, mp_index **out)
{
    // positions are 32-bit on the device; the topmost MP_SEED_BIG values stay free so that a seed projection that wrapped
    // below zero (DV-DPfunctions.cpp:2489) can be told from a real position (merge_kernel)
    if (h.n == 0 || h.n > 0xFFFFFFFFull - 2 * MP_SEED_BIG) return MP_ERANGE;
    HIPCHK(hipSetDevice(device));
    mp_index *ix = new (std::nothrow) mp_index();
    if (!ix) return MP_ENOMEM;
    ix->device = device;
    ix->n = h.n; ix->inv_sa0 = h.inv_sa0; ix->sa_interval = h.sa_interval;
    for (int i = 0; i < 5; i++) ix->cum[i] = h.cum[i];
    ix->seq_names = std::move(h.seq_names); ix->seq_start = std::move(h.seq_start); ix->seq_len = std::move(h.seq_len);
    ix->grid = std::move(h.grid); ix->translate = std::move(h.translate);
    ix->seq_actual_len = std::move(h.seq_actual_len); ix->n_removed = h.n_removed;
    const uint64_t n = h.n;
    const uint32_t n_blocks = (uint32_t)((n + 2) / MP_OCC_SYMS + 2);   // ranks are asked up to n+1: one spare block keeps loads in bounds
    const uint64_t text_words = n / 16 + 4;
    const size_t occ_bytes = (size_t)n_blocks * sizeof(mp_occ_block), lkt_bytes = (size_t)MP_LKT_ENTRIES * 4, sa_bytes = h.sa.size() * 4,
                 text_bytes = (size_t)text_words * 4;
    void *d_bwt = nullptr, *d_pac = nullptr, *d_cnt = nullptr, *d_run = nullptr, *d_tmp = nullptr;
    auto fail = [&](int rc) { (void)hipFree(d_bwt); (void)hipFree(d_pac); (void)hipFree(d_cnt); (void)hipFree(d_run); (void)hipFree(d_tmp); mp_index_close(ix); return rc; };
    if (hipMalloc(&ix->d_occ, occ_bytes) != hipSuccess || hipMalloc(&ix->d_lkt, lkt_bytes) != hipSuccess || hipMalloc(&ix->d_sa, sa_bytes) != hipSuccess ||
        hipMalloc(&ix->d_text, text_bytes) != hipSuccess) return fail(MP_ENOMEM);
    ix->device_bytes = occ_bytes + lkt_bytes + sa_bytes + text_bytes;
    ix->bytes_occ = occ_bytes; ix->bytes_lkt = lkt_bytes; ix->bytes_sa = sa_bytes; ix->bytes_text = text_bytes;
    const size_t bwt_in = ((size_t)n_blocks * 12 + 16) * 4, pac_in = (size_t)text_words * 4 + 16;
    if (hipMalloc(&d_bwt, bwt_in) != hipSuccess || hipMalloc(&d_pac, pac_in) != hipSuccess || hipMalloc(&d_cnt, (size_t)n_blocks * 16) != hipSuccess ||
        hipMalloc(&d_run, (size_t)n_blocks * 16) != hipSuccess) return fail(MP_ENOMEM);
    if (hipMemset(d_bwt, 0, bwt_in) != hipSuccess || hipMemset(d_pac, 0, pac_in) != hipSuccess) return fail(MP_ENODEV);
    if (hipMemcpy(d_bwt, h.bwt_words.data(), std::min(bwt_in, h.bwt_words.size() * 4), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_pac, h.pac.data(), std::min(pac_in, h.pac.size()), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_sa, h.sa.data(), sa_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(MP_ENODEV);
    // rank blocks: per-block symbol counts -> exclusive scans -> blocks with running counts and LSB-first payload
    hipLaunchKernelGGL(occ_count_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, 0, (const uint32_t *)d_bwt, n, n_blocks, (uint32_t *)d_cnt);
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (uint32_t *)d_cnt, (uint32_t *)d_run, (int)n_blocks, 0);
    size_t tmp2 = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp2, (uint32_t *)ix->d_lkt, (uint32_t *)ix->d_lkt, (int)MP_LKT_ENTRIES, 0);
    tmp_bytes = std::max(tmp_bytes, tmp2) + 256;
    if (hipMalloc(&d_tmp, tmp_bytes) != hipSuccess) return fail(MP_ENOMEM);
    for (int c = 0; c < 4; c++)
        (void)hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, (uint32_t *)d_cnt + (size_t)c * n_blocks, (uint32_t *)d_run + (size_t)c * n_blocks, (int)n_blocks, 0);
    hipLaunchKernelGGL(occ_build_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, 0, (const uint32_t *)d_bwt, n, n_blocks, (const uint32_t *)d_run,
                       (mp_occ_block *)ix->d_occ);
    hipLaunchKernelGGL(text_build_kernel, dim3(grid_for(text_words, 256)), dim3(256), 0, 0, (const uint8_t *)d_pac, n, text_words, (uint32_t *)ix->d_text);
    if (!h.lkt.empty()) {
        if (hipMemcpy(ix->d_lkt, h.lkt.data(), lkt_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(MP_ENODEV);
    } else {
        // LTConstruct.c:44-105: count every 13-mer window of (text + 12 x 'A'), then an inclusive prefix sum
        if (hipMemset(ix->d_lkt, 0, lkt_bytes) != hipSuccess) return fail(MP_ENODEV);
        if (n) hipLaunchKernelGGL(lkt_hist_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_text, n, (uint32_t *)ix->d_lkt);
        (void)hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_bytes, (uint32_t *)ix->d_lkt, (uint32_t *)ix->d_lkt, (int)MP_LKT_ENTRIES, 0);
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail(MP_ENODEV);
    (void)hipFree(d_bwt); (void)hipFree(d_pac); (void)hipFree(d_cnt); (void)hipFree(d_run); (void)hipFree(d_tmp);
    d_bwt = d_pac = d_cnt = d_run = d_tmp = nullptr;
    {
        // 17-mer pre-filter of the seeding kernel (kernels_index.hip): 2 GiB; worth it for genome-sized chunks when the
        // HBM is not needed for more index replicas.  MP_KMER_FILTER=0 / 1 overrides the automatic choice.
        const char *e = getenv("MP_KMER_FILTER");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        bool on = n >= (1ull << 25) && free_b >= (12ull << 30);
        uint32_t K = 17;
        if (e) { on = atoi(e) != 0; if (atoi(e) == 16) K = 16; }          // MP_KMER_FILTER=16: one bit per 16-mer, 512 MiB (the lean layout of many chunks)
        const size_t kb = (size_t)1 << (2 * K - 3);
        ix->d.kmer_k = 0;
        if (on && n >= 17 && hipMalloc(&ix->d_kmer, kb) == hipSuccess) {
            if (hipMemset(ix->d_kmer, 0, kb) != hipSuccess) return fail(MP_ENODEV);
            hipLaunchKernelGGL(kmer17_build_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_text, n, (uint32_t *)ix->d_kmer, K);
            if (hipDeviceSynchronize() != hipSuccess) return fail(MP_ENODEV);
            ix->device_bytes += kb; ix->bytes_kmer = kb;
            ix->d.kmer_k = K;
        }
    }
    if (ix->d_kmer && ix->d.kmer_k == 17) {
        // the pre-filter's bits once more, per 15-mer (kernels_index.hip): 8.6 GB, one load per three probed positions;
        // automatic when that leaves >= 16 GiB free, MP_KMER15=0 / 1 overrides
        const char *e = getenv("MP_KMER15");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t bytes = ((size_t)1 << 30) * 8;
        bool on = n >= (1ull << 25) && free_b >= bytes + (16ull << 30);
        if (e) on = atoi(e) != 0;
        if (on && n >= 17 && hipMalloc(&ix->d_kmer15, bytes) == hipSuccess) {
            if (hipMemset(ix->d_kmer15, 0, bytes) != hipSuccess) return fail(MP_ENODEV);
            hipLaunchKernelGGL(kmer15x3_build_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_text, n, (unsigned long long *)ix->d_kmer15);
            if (hipDeviceSynchronize() != hipSuccess) return fail(MP_ENODEV);
            ix->device_bytes += bytes; ix->bytes_kmer15 = bytes;
        }
    }
    if (ix->d_kmer && ix->d.kmer_k == 17) {
        // 16-mer look-up table behind the pre-filter (kernels_index.hip): 17 GB, three backward-search steps less per seed
        // start; automatic when that leaves >= 16 GiB free, MP_LKT16=0 / 1 overrides
        const char *e = getenv("MP_LKT16");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t entries = (size_t)1 << 32, bytes = entries * 4;
        bool on = n >= (1ull << 25) && free_b >= bytes + (16ull << 30);
        if (e) on = atoi(e) != 0;
        if (on && hipMalloc(&ix->d_lkt16, bytes) == hipSuccess) {
            uint32_t *t16 = (uint32_t *)ix->d_lkt16;
            if (hipMemset(t16, 0, bytes) != hipSuccess) return fail(MP_ENODEV);
            hipLaunchKernelGGL(lkt16_hist_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_text, n, t16);
            // inclusive prefix sum over 2^32 entries: four scans of 2^30 (the library takes int counts), then the carries
            const size_t part = (size_t)1 << 30;
            size_t tb = 0;
            (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb, t16, t16, (int)part, 0);
            void *tmp16 = nullptr;
            if (hipMalloc(&tmp16, tb + 256) != hipSuccess) return fail(MP_ENOMEM);
            uint32_t carry = 0;
            for (int c = 0; c < 4; c++) {
                (void)hipcub::DeviceScan::InclusiveSum(tmp16, tb, t16 + c * part, t16 + c * part, (int)part, 0);
                if (carry) hipLaunchKernelGGL(add_offset_kernel, dim3(grid_for(part, 256)), dim3(256), 0, 0, t16 + c * part, (uint64_t)part, carry);
                uint32_t last = 0;
                if (hipMemcpy(&last, t16 + (c + 1) * part - 1, 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(tmp16); return fail(MP_ENODEV); }
                carry = last;
            }
            (void)hipFree(tmp16);
            if (hipDeviceSynchronize() != hipSuccess) return fail(MP_ENODEV);
            ix->device_bytes += bytes; ix->bytes_lkt16 = bytes;
        }
    }
    ix->d.occ = (const mp_occ_block *)ix->d_occ; ix->d.lkt = (const uint32_t *)ix->d_lkt; ix->d.lkt16 = (const uint32_t *)ix->d_lkt16;
    ix->d.sa = (const uint32_t *)ix->d_sa; ix->d.text = (const uint32_t *)ix->d_text; ix->d.kmer17 = (const uint32_t *)ix->d_kmer; ix->d.kmer15x3 = (const unsigned long long *)ix->d_kmer15;
    ix->d.n = (uint32_t)h.n; ix->d.inv_sa0 = (uint32_t)h.inv_sa0; ix->d.sa_interval = h.sa_interval;
    ix->d.sa_shift = 0xFFFFFFFFu;
    for (uint32_t b = 0; b < 31; b++) if ((1u << b) == h.sa_interval) ix->d.sa_shift = b;
    for (int i = 0; i < 5; i++) ix->d.cum[i] = (uint32_t)h.cum[i];
    {
        // denser SA samples, derived here from the file's: the whole suffix array (4n bytes, locate needs no LF step)
        // when that leaves >= 16 GiB free, else every 2nd or every 4th row (1.5 LF steps on average instead of 7.5 with
        // the file's interval 16); same automatic rule and override as the pre-filter (MP_SA_DENSE=0 / 1;
        // MP_SA_DENSE_SPACING=1|2|4 forces a spacing)
        const char *e = getenv("MP_SA_DENSE"), *sp = getenv("MP_SA_DENSE_SPACING");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        bool on = n >= (1ull << 25) && free_b >= n + (12ull << 30);
        if (e) on = atoi(e) != 0;
        uint32_t shift = free_b >= 4 * n + (16ull << 30) ? 0u : free_b >= 2 * n + (14ull << 30) ? 1u : 2u;
        if (e && atoi(e) != 0 && !sp && n < (1ull << 25)) shift = 0;                  // forced on for a small text: the whole array
        if (sp) shift = atoi(sp) <= 1 ? 0u : atoi(sp) == 2 ? 1u : 2u;
        if (h.sa_interval <= (1u << shift)) on = false;
        const uint64_t rows64 = (n >> shift) + 1;
        if (on && rows64 < 0xFFFFFFFFull && hipMalloc(&ix->d_sa_dense, (size_t)rows64 * 4) == hipSuccess) {
            const uint32_t rows = (uint32_t)rows64;
            ix->d.sa_dense_shift = shift;
            hipLaunchKernelGGL(sa_dense_build_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, 0, ix->d, rows, shift, (uint32_t *)ix->d_sa_dense);
            if (hipDeviceSynchronize() != hipSuccess) { mp_index_close(ix); return MP_ENODEV; }
            ix->d.sa_dense = (const uint32_t *)ix->d_sa_dense;
            ix->device_bytes += (size_t)rows * 4; ix->bytes_sa_dense = (size_t)rows * 4;
        }
    }
    if (ix->d_lkt16 && ix->d_sa_dense && ix->d.sa_dense_shift == 0) {
        // the 16-mer table with 16-byte entries {l, r, pos, prev16} (kernels_index.hip): 69 GB; a seed start behind the pre-filter
        // then touches one sector instead of three.  Automatic when that leaves >= 16 GiB free, MP_LKT16U=0 / 1 overrides.
        const char *e = getenv("MP_LKT16U");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t entries = (size_t)1 << 32, bytes = entries * sizeof(uint4);
        bool on = n >= (1ull << 25) && free_b >= bytes + (16ull << 30);
        if (e) on = atoi(e) != 0;
        if (on && hipMalloc(&ix->d_lkt16u, bytes) == hipSuccess) {
            hipLaunchKernelGGL(lkt16u_init_kernel, dim3(1u << 20), dim3(256), 0, 0, (uint4 *)ix->d_lkt16u, (uint64_t)entries);
            if (hipGetLastError() != hipSuccess) { mp_index_close(ix); return MP_ENODEV; }
            hipLaunchKernelGGL(lkt16u_build_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_text, n, (const uint32_t *)ix->d_sa_dense,
                               (const uint32_t *)ix->d_lkt16, (uint4 *)ix->d_lkt16u);
            if (hipDeviceSynchronize() != hipSuccess) { mp_index_close(ix); return MP_ENODEV; }
            ix->d.lkt16u = (const uint4 *)ix->d_lkt16u;
            ix->device_bytes += bytes; ix->bytes_lkt16u = bytes;
        }
    }
    if (const char *e = getenv("MP_LKT14")) {
        // the 13-mer table as 8-byte entries with a seed's first extension step in them (mp_dev_index::lkt14, lkt14_build_kernel): 512 MiB IN PLACE of
        // the table's 256 -- a seed start of the lean layout then costs one rank block less (3.8 -> 2.6 per start against a 2.95 Gbp chunk).  Explicit only
        // (the NT stage's many resident chunks: bench.py); the seeds are the same either way.
        if (atoi(e) != 0) {
            const size_t bytes = ((size_t)MP_LKT_ENTRIES + 1) * sizeof(uint2);
            unsigned long long *d_n = nullptr, h_n = 0;
            if (hipMalloc(&ix->d_lkt14, bytes) != hipSuccess || hipMalloc((void **)&d_n, 8) != hipSuccess || hipMemset(d_n, 0, 8) != hipSuccess) { if (d_n) (void)hipFree(d_n); mp_index_close(ix); return MP_ENOMEM; }
            hipLaunchKernelGGL(lkt14_build_kernel, dim3(grid_for((uint64_t)MP_LKT_ENTRIES + 1, 256)), dim3(256), 0, 0, ix->d, (uint2 *)ix->d_lkt14, d_n);
            const bool ok = hipMemcpy(&h_n, d_n, 8, hipMemcpyDeviceToHost) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            (void)hipFree(d_n);
            if (!ok) { mp_index_close(ix); return MP_ENODEV; }
            if (getenv("MP_DEBUG")) fprintf(stderr, "[megapath_hip] lkt14: %llu of %u entries keep the 13-mer path\n", h_n, MP_LKT_ENTRIES);
            (void)hipFree(ix->d_lkt); ix->d_lkt = nullptr; ix->d.lkt = nullptr;
            ix->d.lkt14 = (const uint2 *)ix->d_lkt14;
            ix->device_bytes += bytes; ix->device_bytes -= ix->bytes_lkt; ix->bytes_lkt14 = bytes; ix->bytes_lkt = 0;
        }
    }
    *out = ix;
    return MP_OK;
}

extern "C" int mp_index_open(const char *prefix, int device, mp_index **out)
{
    if (!prefix || !out) return MP_EINVAL;
    int rc = mp_device_check(device, nullptr, 0);
    if (rc) { fprintf(stderr, "[megapath_hip] no gfx950 device %d: the HIP path cannot run (there is no CPU fallback)\n", device); return rc; }
    mp_host_index h;
    std::string err;
    rc = mp_host_index_load(prefix, h, err);
    if (rc) { fprintf(stderr, "[megapath_hip] %s\n", err.c_str()); return rc; }
    return upload_index(device, h, out);
}

extern "C" int mp_index_from_arrays(int device, uint64_t text_length, uint64_t inverse_sa0, const uint64_t cum_freq[5],
                                    const uint32_t *bwt_words, const uint64_t *sa_samples, uint32_t sa_interval,
                                    const uint8_t *pac_bytes, const uint64_t *lkt, uint32_t num_seq, const char *const *seq_names,
                                    const uint64_t *seq_starts, const uint64_t *seq_lengths, mp_index **out)
{
    if (!out) return MP_EINVAL;
    int rc = mp_device_check(device, nullptr, 0);
    if (rc) { fprintf(stderr, "[megapath_hip] no gfx950 device %d: the HIP path cannot run (there is no CPU fallback)\n", device); return rc; }
    mp_host_index h;
    std::string err;
    rc = mp_host_index_from_arrays(text_length, inverse_sa0, cum_freq, bwt_words, sa_samples, sa_interval, pac_bytes, lkt,
                                   num_seq, seq_names, seq_starts, seq_lengths, h, err);
    if (rc) { fprintf(stderr, "[megapath_hip] %s\n", err.c_str()); return rc; }
    return upload_index(device, h, out);
}

// A second copy of a resident index in the same HBM: every array copied device-to-device (milliseconds per GB instead of the
// host parse + upload + re-layout of mp_index_open), the host-side annotation duplicated.  What a benchmark uses to fill the HBM
// with as many index chunks as a full database has; the copies are ordinary indexes (mp_index_close frees them).
extern "C" int mp_index_clone(const mp_index *src, mp_index **out)
{
    if (!src || !out) return MP_EINVAL;
    HIPCHK(hipSetDevice(src->device));
    mp_index *ix = new (std::nothrow) mp_index(*src);                 // annotation, sizes and the kernel view (pointers fixed below)
    if (!ix) return MP_ENOMEM;
    ix->d_occ = ix->d_lkt = ix->d_sa = ix->d_text = ix->d_kmer = ix->d_sa_dense = ix->d_lkt16 = ix->d_kmer15 = ix->d_lkt16u = ix->d_lkt14 = nullptr;
    ix->host_text = std::make_shared<mp_index::HostText>();
    ix->d_tr_grid = ix->d_tr_start = ix->d_tr_chr = nullptr;          // uploaded again on first use
    struct Arr { void *const *from; void **to; size_t bytes; };
    const Arr arrs[] = {{&src->d_occ, &ix->d_occ, src->bytes_occ}, {&src->d_lkt, &ix->d_lkt, src->bytes_lkt}, {&src->d_sa, &ix->d_sa, src->bytes_sa},
                        {&src->d_text, &ix->d_text, src->bytes_text}, {&src->d_kmer, &ix->d_kmer, src->bytes_kmer}, {&src->d_sa_dense, &ix->d_sa_dense, src->bytes_sa_dense},
                        {&src->d_lkt16, &ix->d_lkt16, src->bytes_lkt16}, {&src->d_kmer15, &ix->d_kmer15, src->bytes_kmer15}, {&src->d_lkt16u, &ix->d_lkt16u, src->bytes_lkt16u}, {&src->d_lkt14, &ix->d_lkt14, src->bytes_lkt14}};
    for (const Arr &a : arrs) {
        if (!*a.from || a.bytes == 0) continue;
        if (hipMalloc(a.to, a.bytes) != hipSuccess) { mp_index_close(ix); return MP_ENOMEM; }
        if (hipMemcpy(*a.to, *a.from, a.bytes, hipMemcpyDeviceToDevice) != hipSuccess) { mp_index_close(ix); return MP_ENODEV; }
    }
    if (hipDeviceSynchronize() != hipSuccess) { mp_index_close(ix); return MP_ENODEV; }
    ix->d.occ = (const mp_occ_block *)ix->d_occ; ix->d.lkt = (const uint32_t *)ix->d_lkt; ix->d.sa = (const uint32_t *)ix->d_sa; ix->d.text = (const uint32_t *)ix->d_text;
    ix->d.kmer17 = (const uint32_t *)ix->d_kmer; ix->d.sa_dense = (const uint32_t *)ix->d_sa_dense; ix->d.lkt16 = (const uint32_t *)ix->d_lkt16;
    ix->d.kmer15x3 = (const unsigned long long *)ix->d_kmer15; ix->d.lkt16u = (const uint4 *)ix->d_lkt16u; ix->d.lkt14 = (const uint2 *)ix->d_lkt14;
    *out = ix;
    return MP_OK;
}

extern "C" int mp_index_close(mp_index *ix)
{
    if (!ix) return MP_OK;
    (void)hipSetDevice(ix->device);
    if (ix->d_occ) (void)hipFree(ix->d_occ);
    if (ix->d_lkt) (void)hipFree(ix->d_lkt);
    if (ix->d_lkt14) (void)hipFree(ix->d_lkt14);
    if (ix->d_sa) (void)hipFree(ix->d_sa);
    if (ix->d_text) (void)hipFree(ix->d_text);
    if (ix->d_kmer) (void)hipFree(ix->d_kmer);
    if (ix->d_lkt16) (void)hipFree(ix->d_lkt16);
    if (ix->d_kmer15) (void)hipFree(ix->d_kmer15);
    if (ix->d_sa_dense) (void)hipFree(ix->d_sa_dense);
    if (ix->d_lkt16u) (void)hipFree(ix->d_lkt16u);
    if (ix->d_tr_grid) (void)hipFree(ix->d_tr_grid);
    if (ix->d_tr_start) (void)hipFree(ix->d_tr_start);
    if (ix->d_tr_chr) (void)hipFree(ix->d_tr_chr);
    delete ix;
    return MP_OK;
}

extern "C" int mp_index_get_info(const mp_index *ix, mp_index_info *out)
{
    if (!ix || !out) return MP_EINVAL;
    out->text_length = ix->n; out->inverse_sa0 = ix->inv_sa0;
    for (int i = 0; i < 5; i++) out->cum_freq[i] = ix->cum[i];
    out->sa_interval = ix->sa_interval; out->num_seq = (uint32_t)ix->seq_names.size();
    out->device_bytes = ix->device_bytes;
    out->prefilter_k = ix->d_kmer ? ix->d.kmer_k : 0u; out->reserved = 0;
    return MP_OK;
}

extern "C" const char *mp_index_seq_name(const mp_index *ix, uint32_t id)
{
    if (!ix || id == 0 || id > ix->seq_names.size()) return "";
    return ix->seq_names[id - 1].c_str();
}

extern "C" int mp_index_chr_and_pos(const mp_index *ix, uint64_t amb_pos, uint64_t *out_pos, uint32_t *out_seq)
{
    if (!ix || !out_pos || !out_seq || ix->grid.empty()) return MP_EINVAL;
    uint64_t g = amb_pos >> 18;                         // GRID_SAMPLING_FACTOR_2_POWER (HSP.h:68)
    if (g >= ix->grid.size()) g = ix->grid.size() - 1;
    uint32_t v = ix->grid[g];
    while (v > 0 && ix->translate[v].start > amb_pos) v--;
    *out_pos = amb_pos - ix->translate[v].corr;
    *out_seq = ix->translate[v].chr;
    return MP_OK;
}

// decideTargetChr (BGS-IO.cpp:1312-1341): the sequence an alignment of `len` bases at amb_pos lies in, 0 when its two ends fall
// into different sequences.  One table walk when both ends share a row of the translation table (nearly always).
extern "C" int mp_index_target_chr(const mp_index *ix, uint64_t amb_pos, uint32_t len, uint32_t *out_seq)
{
    if (!ix || !out_seq || ix->grid.empty()) return MP_EINVAL;
    uint64_t g = amb_pos >> 18;
    if (g >= ix->grid.size()) g = ix->grid.size() - 1;
    uint32_t v = ix->grid[g];
    while (v > 0 && ix->translate[v].start > amb_pos) v--;
    const uint32_t c1 = ix->translate[v].chr;
    const uint64_t last = amb_pos + len - 1;
    uint32_t c2 = c1;
    if ((size_t)v + 1 < ix->translate.size() && ix->translate[v + 1].start <= last) {
        uint64_t g2 = last >> 18;
        if (g2 >= ix->grid.size()) g2 = ix->grid.size() - 1;
        uint32_t w = ix->grid[g2];
        while (w > 0 && ix->translate[w].start > last) w--;
        c2 = ix->translate[w].chr;
    }
    *out_seq = c1 == c2 ? c1 : 0u;
    return MP_OK;
}

// getChrAndPos with its return value (BGS-IO.cpp:163-188).  The reference evaluates `row < numOfRemovedSegment - 1` in
// unsigned 64-bit arithmetic, so with no removed segment every row takes the "next row" branch; for the last row that
// is a read past its array, for which the text length is returned here.
extern "C" int mp_index_chr_pos_segment(const mp_index *ix, uint64_t amb_pos, uint64_t *out_pos, uint32_t *out_seq, uint64_t *out_segment_end)
{
    if (!ix || !out_pos || !out_seq || !out_segment_end || ix->grid.empty()) return MP_EINVAL;
    uint64_t g = amb_pos >> 18;
    if (g >= ix->grid.size()) g = ix->grid.size() - 1;
    uint32_t v = ix->grid[g];
    while (v > 0 && ix->translate[v].start > amb_pos) v--;
    *out_pos = amb_pos - ix->translate[v].corr;
    *out_seq = ix->translate[v].chr;
    if ((uint64_t)v < ix->n_removed - 1 && (size_t)v + 1 < ix->translate.size()) *out_segment_end = ix->translate[v + 1].start - 1;
    else *out_segment_end = ix->n;
    return MP_OK;
}

extern "C" int mp_index_seq_info(const mp_index *ix, uint32_t seq_id_1based, uint64_t *out_start, uint64_t *out_packed_len, uint64_t *out_actual_len)
{
    if (!ix || seq_id_1based == 0 || seq_id_1based > ix->seq_names.size()) return MP_EINVAL;
    if (out_start) *out_start = ix->seq_start[seq_id_1based - 1];
    if (out_packed_len) *out_packed_len = ix->seq_len[seq_id_1based - 1];
    if (out_actual_len) *out_actual_len = ix->seq_actual_len[seq_id_1based - 1];
    return MP_OK;
}

// Reference bases for the MD / NM tags of the BAM writer (-b -p).  Served from a host mirror of the packed text, copied back from
// the device once: the callers are the printer's formatting threads, thousands of small requests per batch, and a device round
// trip for each (allocation, two copies, a launch on the null stream) stalled every context of the GPU.
extern "C" int mp_index_fetch_text(const mp_index *ix, const uint64_t *pos, const uint32_t *len, uint32_t n_ranges, uint8_t *out, uint32_t stride)
{
    if (!ix || !pos || !len || !out || stride == 0) return MP_EINVAL;
    if (n_ranges == 0) return MP_OK;
    mp_index::HostText &ht = *ix->host_text;
    {
        std::lock_guard<std::mutex> lk(ht.mu);
        if (!ht.ready) {
            HIPCHK(hipSetDevice(ix->device));
            ht.words.resize(ix->bytes_text / 4);
            HIPCHK(hipMemcpy(ht.words.data(), ix->d_text, ix->bytes_text, hipMemcpyDeviceToHost));
            ht.ready = true;
        }
    }
    const uint32_t *text = ht.words.data();
    const uint64_t n = ix->n;
    for (uint32_t r = 0; r < n_ranges; r++) {
        const uint64_t p0 = pos[r];
        const uint32_t L = len[r] < stride ? len[r] : stride;
        uint8_t *o = out + (size_t)r * stride;
        for (uint32_t i = 0; i < L; i++) {
            const uint64_t p = p0 + i;
            o[i] = p < n ? (uint8_t)((text[p >> 4] >> (2 * (p & 15))) & 3u) : (uint8_t)0;
        }
    }
    return MP_OK;
}

extern "C" int mp_index_chr_and_pos_batch(const mp_index *ix, const uint64_t *amb_pos, uint64_t n, uint64_t *out_pos, uint32_t *out_seq)
{
    if (!ix || !amb_pos || !out_pos || !out_seq) return MP_EINVAL;
    for (uint64_t i = 0; i < n; i++) {
        int rc = mp_index_chr_and_pos(ix, amb_pos[i], &out_pos[i], &out_seq[i]);
        if (rc) return rc;
    }
    return MP_OK;
}

template <typename F>
static int run_query(const mp_index *ix, uint64_t n, F &&body)
{
    if (!ix) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    int rc = body();
    HIPCHK(hipDeviceSynchronize());
    return rc;
}

extern "C" int mp_index_occ(const mp_index *ix, const uint64_t *idx, const uint32_t *sym, uint64_t n, uint64_t *out)
{
    if (!ix || !idx || !sym || !out) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint64_t *d_idx, *d_out; uint32_t *d_sym;
    HIPCHK(hipMalloc((void **)&d_idx, n * 8 + 8)); HIPCHK(hipMalloc((void **)&d_out, n * 8 + 8)); HIPCHK(hipMalloc((void **)&d_sym, n * 4 + 8));
    HIPCHK(hipMemcpy(d_idx, idx, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_sym, sym, n * 4, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(occ_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, ix->d, d_idx, d_sym, n, d_out);
    HIPCHK(hipMemcpy(out, d_out, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_idx); (void)hipFree(d_out); (void)hipFree(d_sym);
    return MP_OK;
}

extern "C" int mp_index_sa(const mp_index *ix, const uint64_t *idx, uint64_t n, uint64_t *out)
{
    if (!ix || !idx || !out) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint64_t *d_idx, *d_out;
    HIPCHK(hipMalloc((void **)&d_idx, n * 8 + 8)); HIPCHK(hipMalloc((void **)&d_out, n * 8 + 8));
    HIPCHK(hipMemcpy(d_idx, idx, n * 8, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(sa_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, ix->d, d_idx, n, d_out);
    HIPCHK(hipMemcpy(out, d_out, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_idx); (void)hipFree(d_out);
    return MP_OK;
}

extern "C" int mp_index_lkt(const mp_index *ix, const uint32_t *key, uint64_t n, uint64_t *out_l, uint64_t *out_r)
{
    if (!ix || !key || !out_l || !out_r) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint32_t *d_key; uint64_t *d_l, *d_r;
    HIPCHK(hipMalloc((void **)&d_key, n * 4 + 8)); HIPCHK(hipMalloc((void **)&d_l, n * 8 + 8)); HIPCHK(hipMalloc((void **)&d_r, n * 8 + 8));
    HIPCHK(hipMemcpy(d_key, key, n * 4, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(lkt_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, ix->d, d_key, n, d_l, d_r);
    HIPCHK(hipMemcpy(out_l, d_l, n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out_r, d_r, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_key); (void)hipFree(d_l); (void)hipFree(d_r);
    return MP_OK;
}

extern "C" int mp_index_copy_lkt(const mp_index *ix, uint32_t *out)
{
    if (!ix || !out || (!ix->d_lkt && !ix->d_lkt14)) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    if (ix->d_lkt) { HIPCHK(hipMemcpy(out, ix->d_lkt, (size_t)MP_LKT_ENTRIES * 4, hipMemcpyDeviceToHost)); return MP_OK; }
    // the table lives in the 8-byte entries (mp_dev_index::lkt14): lkt[k] = entry[k + 1].x
    std::vector<uint2> e;
    try { e.resize((size_t)MP_LKT_ENTRIES + 1); } catch (...) { return MP_ENOMEM; }
    HIPCHK(hipMemcpy(e.data(), ix->d_lkt14, e.size() * sizeof(uint2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < MP_LKT_ENTRIES; k++) out[k] = e[k + 1].x;
    return MP_OK;
}

extern "C" int mp_index_prefilter(const mp_index *ix, const uint64_t *keys, uint64_t n, uint8_t *out)
{
    if (!ix || !keys || !out) return MP_EINVAL;
    if (!ix->d_kmer || ix->d.kmer_k != 17) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint64_t *d_k = nullptr; uint8_t *d_o = nullptr;
    HIPCHK(hipMalloc((void **)&d_k, n * 8 + 8)); HIPCHK(hipMalloc((void **)&d_o, n + 8));
    HIPCHK(hipMemcpy(d_k, keys, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kmer17_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const uint32_t *)ix->d_kmer, d_k, n, d_o);
    HIPCHK(hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost));
    (void)hipFree(d_k); (void)hipFree(d_o);
    return MP_OK;
}

extern "C" int mp_index_prefilter3(mp_index *ix, const uint64_t *keys, uint64_t n, uint8_t *out)
{
    if (!ix || !keys || !out) return MP_EINVAL;
    if (!ix->d_kmer15) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint64_t *d_k = nullptr; uint8_t *d_o = nullptr;
    if (hipMalloc((void **)&d_k, n * 8) != hipSuccess || hipMalloc((void **)&d_o, n) != hipSuccess) { if (d_k) (void)hipFree(d_k); return MP_ENOMEM; }
    hipError_t e = hipMemcpy(d_k, keys, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kmer15x3_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, (const unsigned long long *)ix->d_kmer15, d_k, n, d_o);
        e = hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_k); (void)hipFree(d_o);
    return e == hipSuccess ? MP_OK : MP_ENODEV;
}

extern "C" int mp_index_check_lkt16(mp_index *ix, uint64_t n_random, uint64_t *out_checked, uint64_t *out_mismatches)
{
    if (!ix || !out_checked || !out_mismatches) return MP_EINVAL;
    if (!ix->d_lkt16) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    HIPCHK(hipMalloc((void **)&d_out, 16));
    HIPCHK(hipMemset(d_out, 0, 16));
    const uint64_t n_text = ix->n >= 4 ? ix->n - 3 : 0;
    hipLaunchKernelGGL(lkt16_check_kernel, dim3(grid_for(n_text + n_random, 256)), dim3(256), 0, 0, ix->d, n_text, n_random, d_out);
    hipError_t e = hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return MP_ENODEV;
    *out_checked = h_out[0]; *out_mismatches = h_out[1];
    return MP_OK;
}

extern "C" int mp_bench_gather64(const mp_index *ix, uint32_t lanes, uint32_t per_lane, uint32_t repeats, float *out_ms, uint64_t *out_bytes)
{
    if (!ix || !out_ms || !out_bytes || lanes == 0 || per_lane == 0 || repeats == 0) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    uint32_t *sink; hipEvent_t a, b;
    HIPCHK(hipMalloc((void **)&sink, 64)); HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const uint32_t n_blocks = (uint32_t)((ix->n + 2) / MP_OCC_SYMS + 1);
    hipLaunchKernelGGL(gather64_kernel, dim3(grid_for(lanes, 256)), dim3(256), 0, 0, ix->d, n_blocks, per_lane, 1u, sink);   // warm-up
    HIPCHK(hipEventRecord(a, 0));
    for (uint32_t r = 0; r < repeats; r++)
        hipLaunchKernelGGL(gather64_kernel, dim3(grid_for(lanes, 256)), dim3(256), 0, 0, ix->d, n_blocks, per_lane, 77u + r, sink);
    HIPCHK(hipEventRecord(b, 0)); HIPCHK(hipEventSynchronize(b));
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    *out_ms = ms / repeats;
    *out_bytes = (uint64_t)grid_for(lanes, 256) * 256ull * per_lane * 64ull;
    (void)hipFree(sink); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ context
extern "C" int mp_ctx_create(mp_index *ix, mp_ctx **out)
{
    if (!ix || !out) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    mp_ctx *c = new (std::nothrow) mp_ctx();
    if (!c) return MP_ENOMEM;
    c->ix = ix;
    bool ok = hipStreamCreate(&c->stream) == hipSuccess && hipStreamCreate(&c->copy_stream) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_hits_ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_hits_copied, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_rescue_ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_rescue_copied, hipEventDisableTiming) == hipSuccess;
    for (auto &e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    for (auto &e : c->ev_round) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) { fprintf(stderr, "[megapath_hip] mp_ctx_create: stream / event creation failed: %s\n", hipGetErrorString(hipGetLastError())); mp_ctx_destroy(c); return MP_ENODEV; }
    if (c->ctr.reserve(1) || c->n_pair_hits.reserve(1)) { mp_ctx_destroy(c); return MP_ENOMEM; }
    *out = c;
    return MP_OK;
}

extern "C" int mp_ctx_destroy(mp_ctx *c)
{
    if (!c) return MP_OK;
    (void)hipSetDevice(c->ix->device);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_hits_ready) (void)hipEventDestroy(c->ev_hits_ready);
    if (c->ev_hits_copied) (void)hipEventDestroy(c->ev_hits_copied);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_round) if (e) (void)hipEventDestroy(e);
    if (c->ev_rescue_ready) (void)hipEventDestroy(c->ev_rescue_ready);
    if (c->ev_rescue_copied) (void)hipEventDestroy(c->ev_rescue_copied);
    if (c->h_single_hits) (void)hipHostFree(c->h_single_hits);
    if (c->h_rescue_hits) (void)hipHostFree(c->h_rescue_hits);
    c->census.release();
    c->sh_flag.release(); c->sh_slot.release(); c->rh_flag.release(); c->rh_slot.release(); c->d_single_hits.release(); c->d_rescue_hits.release();
    c->band_flag.release(); c->band_idx.release(); c->near_flag.release(); c->near_idx.release(); c->near_list.release(); c->far_flag.release(); c->far_idx.release();
    c->far_list.release(); c->pre_cls.release(); c->narrow_list.release(); c->wide_list.release(); c->narrow15_list.release(); c->score_list.release(); c->mid_flag.release(); c->mid_idx.release(); c->mid_list.release();
    c->vfar_flag.release(); c->vfar_idx.release(); c->vfar_list.release(); c->und_list[0].release(); c->und_list[1].release(); c->band_tasks.release();
    c->band_res.release(); c->band_fill.release();
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    c->reads.release(); c->lens.release(); c->scan.release(); c->seeds.release(); c->slabs.release(); c->big_slabs.release(); c->hits.release();
    c->seedpos.release(); c->hit_row.release(); c->seeds_per_rs.release(); c->hits_per_rs.release(); c->rs_base.release(); c->overflow_rs.release(); c->over_slot.release();
    c->queue.release(); c->n_pos.release(); c->n_neg.release(); c->cub_tmp.release(); c->ctr.release(); c->cands.release(); c->tasks_l.release(); c->tasks_r.release();
    c->tasks_tmp.release(); c->fill.release(); c->res_l.release(); c->res_r.release(); c->res_tmp.release(); c->flags.release();
    c->ops.release(); c->th_cert.release(); c->th_stats.release(); c->pair_hits.release(); c->pair_cnt.release(); c->pair_off.release(); c->hit_flag.release(); c->hit_idx.release(); c->cand_first.release(); c->n_pair_hits.release(); c->sel_flag.release(); c->sel_idx.release(); c->sel_tasks.release(); c->sel_res.release();
    if (c->h_pair_hits) (void)hipHostFree(c->h_pair_hits);
    if (c->red_keys) (void)hipFree(c->red_keys);
    if (c->red_sorted) (void)hipFree(c->red_sorted);
    if (c->red_out) (void)hipFree(c->red_out);
    if (c->h_red) (void)hipHostFree(c->h_red);
    c->resc_done.release(); c->red_flag.release(); c->red_slot.release(); c->red_counts.release(); c->red_cls.release();
    c->best_known.release(); c->floor_ub.release(); c->floor_counts.release(); c->floor_hope.release();
    if (c->stash_arena) (void)hipFree(c->stash_arena);
    c->pair_done.release(); c->failed_flag.release(); c->failed_off.release(); c->failed_reads.release(); c->n_scand.release(); c->s_off.release();
    c->rsrc.release(); c->scand.release(); c->stasks.release(); c->rtasks.release(); c->sres.release(); c->rres.release(); c->rtasks_c.release(); c->rres_c.release(); c->rlist.release();
    delete c;
    return MP_OK;
}

// A new batch: nothing computed for the previous one may leak into it (seeds, hit lists, failed-pair lists), and copies
// of its hits still in flight must have landed before their buffers are reused.
static int forget_batch(mp_ctx *c)
{
    if (c->hits_copy_started) { HIPCHK(hipEventSynchronize(c->ev_hits_copied)); c->hits_copy_started = false; }
    if (c->rescue_copy_started) { HIPCHK(hipEventSynchronize(c->ev_rescue_copied)); c->rescue_copy_started = false; }
    c->seeded = false; c->deep_done = false; c->rescued = false;
    c->n_pair_hits_host = 0; c->n_hits_total = 0; c->n_stasks = 0; c->n_failed_reads = 0;
    c->n_single_hits_host = 0; c->n_rescue_hits_host = 0;
    return MP_OK;
}

extern "C" int mp_ctx_set_index(mp_ctx *c, mp_index *ix)
{
    if (!c || !ix) return MP_EINVAL;
    if (ix->device != c->ix->device) return MP_EINVAL;          // streams, events and buffers of a context live on one GPU
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t n_reads = c->n_reads;
    int rc = forget_batch(c);                                   // seeds and hit lists belong to the index they were computed on
    c->n_reads = n_reads;                                       // the reads stay (and what mp_chunk_stash keeps of other chunks)
    c->ix = ix;
    return rc;
}

static void forget_stash(mp_ctx *c)
{
    for (auto &st : c->stash) st.used = false;
    c->stash_used = 0;
}

extern "C" int mp_set_reads(mp_ctx *c, const uint32_t *packed, uint32_t W, const uint32_t *lengths, uint32_t n_reads)
{
    if (!c || !packed || !lengths || W == 0 || (n_reads & 1)) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n_reads; i++) { if (lengths[i] > mx) mx = lengths[i]; if (lengths[i] < mn) mn = lengths[i]; if (lengths[i] > W * 16) return MP_EINVAL; }
    if (c->reads.reserve((size_t)n_reads * W + 16) || c->lens.reserve(n_reads + 16)) return MP_ENOMEM;
    HIPCHK(hipMemcpyAsync(c->reads.p, packed, (size_t)n_reads * W * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->lens.p, lengths, (size_t)n_reads * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->d_reads = c->reads.p; c->d_lens = c->lens.p;
    c->n_reads = n_reads; c->W = W; c->max_len = mx; c->min_len = n_reads ? mn : 0;
    forget_stash(c);
    return forget_batch(c);
}

extern "C" int mp_set_reads_device(mp_ctx *c, const uint32_t *d_packed, uint32_t W, const uint32_t *d_lengths, uint32_t n_reads, uint32_t max_length)
{
    if (!c || !d_packed || !d_lengths || W == 0 || (n_reads & 1) || max_length > W * 16) return MP_EINVAL;
    c->d_reads = d_packed; c->d_lens = d_lengths;
    c->n_reads = n_reads; c->W = W; c->max_len = max_length; c->min_len = 0;
    forget_stash(c);
    return forget_batch(c);
}

// ------------------------------------------------------------------------------------------------ seeding
static mp_dev_mmp to_dev(const mp_mmp_params *p)
{
    mp_dev_mmp d;
    d.sa_size_threshold = p->sa_size_threshold; d.seed_min_length = p->seed_min_length; d.uniq_threshold = p->uniq_threshold;
    d.indel_fuzz = p->indel_fuzz; d.good_seed_len = p->good_seed_len; d.reseed_len = p->reseed_len; d.reseed_abs_diff = p->reseed_abs_diff;
    d.reseed_rlt_ratio = p->reseed_rlt_ratio; d.short_seed_ratio = p->short_seed_ratio;
    return d;
}

extern "C" int mp_seed_batch(mp_ctx *c, const mp_mmp_params *params, mp_seed_stats *out_stats)
{
    if (!c || !params || c->n_reads == 0 || !c->d_reads) return MP_EINVAL;
    if (params->seed_min_length < MP_LKT_K || params->sa_size_threshold < 1 || params->sa_size_threshold > 1000) return MP_ERANGE;
    if (c->max_len > MP_SEED_BIG) return MP_ERANGE;
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t n_reads = c->n_reads, W = c->W, n_items = 2 * n_reads;
    const mp_dev_mmp P = to_dev(params);
    uint32_t slab_limit = MP_SEED_SLAB;
    if (const char *e = getenv("MP_SEED_SLAB_LIMIT")) { int v = atoi(e); if (v >= 1 && v < MP_SEED_SLAB) slab_limit = (uint32_t)v; }   // test knob
    mp_dev_index dix = c->ix->d;
    if (!c->use_prefilter) { dix.kmer17 = nullptr; dix.kmer15x3 = nullptr; dix.sa_dense = nullptr; dix.lkt16 = nullptr; dix.lkt16u = nullptr; }
    hipStream_t s = c->stream;
    const uint32_t overflow_cap = n_items;
    if (c->scan.reserve((size_t)n_items * mp_scan_stride(W) + 16) || c->slabs.reserve((size_t)n_items * MP_SEED_SLAB) || c->seeds_per_rs.reserve(n_items + 2) ||
        c->hits_per_rs.reserve(n_items + 2) || c->rs_base.reserve(n_items + 2) || c->overflow_rs.reserve(overflow_cap) || c->over_slot.reserve(n_items + 1) || c->queue.reserve(1) ||
        c->n_pos.reserve(n_reads + 1) || c->n_neg.reserve(n_reads + 1)) return MP_ENOMEM;
    hipLaunchKernelGGL(prep_scan_kernel, dim3(grid_for((uint64_t)n_items * mp_scan_stride(W), 256)), dim3(256), 0, s, c->d_reads, c->d_lens, n_reads, W, c->scan.p);
    MP_TRACE(s, "prep_scan_kernel");
    HIPCHK(hipMemsetAsync(c->ctr.p, 0, sizeof(mp_counters), s));
    HIPCHK(hipMemsetAsync(c->queue.p, 0, sizeof(unsigned int), s));
    HIPCHK(hipMemsetAsync(c->hits_per_rs.p + n_items, 0, 2 * sizeof(uint32_t), s));
    // persistent grid: as many blocks of 256 lanes as the kernel's registers and its LDS rows (one scan row per lane) let a CU hold
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->ix->device);
    const size_t seed_lds = (size_t)256 * (mp_scan_stride(W) | 1u) * sizeof(uint32_t);
    int per_cu = 0;
    auto kern = seed_kernel;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, seed_lds) != hipSuccess || per_cu < 1) per_cu = 4;
    const unsigned blocks = std::min<unsigned>(grid_for(n_items, 256), (unsigned)n_cu * (unsigned)per_cu);
    HIPCHK(hipEventRecord(c->ev[0], s));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), seed_lds, s, dix, c->scan.p, c->d_lens, n_items, W, P, c->slabs.p, c->seeds_per_rs.p,
                       c->hits_per_rs.p, c->overflow_rs.p, c->over_slot.p, overflow_cap, c->queue.p, c->ctr.p, slab_limit);
    MP_TRACE(s, "seed_kernel");
    HIPCHK(hipEventRecord(c->ev[1], s));
    mp_counters h{};
    HIPCHK(hipMemcpyAsync(&h, c->ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
    // per-item segments of the hit array (exclusive scan; entry n_items = total)
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, c->hits_per_rs.p, c->rs_base.p, (int)(n_items + 1), s);
    if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
    (void)hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tmp_bytes, c->hits_per_rs.p, c->rs_base.p, (int)(n_items + 1), s);
    uint32_t n_hits = 0;
    HIPCHK(hipMemcpyAsync(&n_hits, c->rs_base.p + n_items, sizeof n_hits, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const uint32_t n_over = (uint32_t)h.overflow;
    if (n_over > overflow_cap) return MP_ERANGE;
    if (getenv("MP_DEBUG")) {
        float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        fprintf(stderr, "[megapath_hip] seed: %u items, %u overflowed the %d-seed slab, %u hits; seed_kernel %.3f ms, %llu wave trips (%.1f per item-lane slot), %llu lane trips (%.1f per item), "
                        "lane occupancy of trips %.2f\n", n_items, n_over, MP_SEED_SLAB, n_hits, ms, h.n_wave_trips, (double)h.n_wave_trips * 64 / n_items, h.n_lane_trips,
                (double)h.n_lane_trips / n_items, (double)h.n_lane_trips / ((double)h.n_wave_trips * 64));
    }
    HIPCHK(hipEventRecord(c->ev[8], s));
    if (n_over) {
        if (c->big_slabs.reserve((size_t)n_over * MP_SEED_BIG)) return MP_ENOMEM;
        hipLaunchKernelGGL(seed_overflow_kernel, dim3(grid_for(n_over, 64)), dim3(64), 0, s, dix, c->scan.p, c->d_lens, W, P, c->overflow_rs.p, n_over,
                           c->big_slabs.p);
        MP_TRACE(s, "seed_overflow_kernel");
    }
    // With the whole suffix array resident the merge kernel makes its sort keys straight from the seeding kernel's records (a hit is
    // at most one 4-byte load): no expand / locate launch, no 16-byte hit array.  Otherwise (LF walks to a sampled row): one lane per hit.
    const bool fused = dix.sa_dense && dix.sa_dense_shift == 0;
    if (c->seedpos.reserve((size_t)n_hits + 64)) return MP_ENOMEM;
    if (fused) { if (c->hits.reserve(((size_t)n_hits + 1) / 2 + 64)) return MP_ENOMEM; }
    else if (c->seeds.reserve((size_t)n_hits + 64) || c->hit_row.reserve((size_t)n_hits + 64) || c->hits.reserve((size_t)n_hits + 64)) return MP_ENOMEM;
    HIPCHK(hipEventRecord(c->ev[9], s));
    HIPCHK(hipEventRecord(c->ev[2], s));
    if (!fused) {
        hipLaunchKernelGGL(expand_hits_kernel, dim3(grid_for(n_items, 256)), dim3(256), 0, s, c->slabs.p, c->seeds_per_rs.p, c->rs_base.p, n_items,
                           c->over_slot.p, c->big_slabs.p, c->seeds.p, c->hit_row.p, c->ctr.p, slab_limit);
        MP_TRACE(s, "expand_hits_kernel");
        if (n_hits)
            hipLaunchKernelGGL(locate_kernel, dim3(grid_for(((uint64_t)n_hits + 1) / 2, 256)), dim3(256), 0, s, dix, c->seeds.p, c->hit_row.p, n_hits, c->d_lens, P, c->hits.p,
                               c->ctr.p);      // (one lane per two hits)
        MP_TRACE(s, "locate_kernel");
    }
    HIPCHK(hipEventRecord(c->ev[3], s));
    const int cov_words = (int)((c->max_len + 63) / 64);
    uint32_t stage_cap = MP_MERGE_CAP;
    if (const char *e = getenv("MP_MERGE_STAGE_CAP")) { const int v = atoi(e); if (v >= 0 && v < MP_MERGE_CAP) stage_cap = (uint32_t)v; }   // test knob
    const int merge_rows = getenv("MP_MERGE_ROWS") ? atoi(getenv("MP_MERGE_ROWS")) : 1;      // 0: test switch (kernels_seed.hip, merge_reads_rows)
    if (fused)
        hipLaunchKernelGGL(merge_kernel<true>, dim3(grid_for(n_reads, MP_MERGE_BLOCK)), dim3(MP_MERGE_THREADS), 0, s, c->hits.p, c->rs_base.p, n_reads, P, (uint64_t)c->ix->n, cov_words,
                           c->seedpos.p, c->n_pos.p, c->n_neg.p, c->ctr.p, dix, c->slabs.p, c->seeds_per_rs.p, c->over_slot.p, c->big_slabs.p, slab_limit, c->d_lens, stage_cap, merge_rows);
    else
        hipLaunchKernelGGL(merge_kernel<false>, dim3(grid_for(n_reads, MP_MERGE_BLOCK)), dim3(MP_MERGE_THREADS), 0, s, c->hits.p, c->rs_base.p, n_reads, P, (uint64_t)c->ix->n, cov_words,
                           c->seedpos.p, c->n_pos.p, c->n_neg.p, c->ctr.p, dix, c->slabs.p, c->seeds_per_rs.p, c->over_slot.p, c->big_slabs.p, slab_limit, c->d_lens, stage_cap, merge_rows);
    MP_TRACE(s, "merge_kernel");
    HIPCHK(hipEventRecord(c->ev[4], s));
    HIPCHK(hipMemcpyAsync(&h, c->ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    mp_seed_stats st{};
    st.n_occ = h.n_occ; st.n_lkt = h.n_lkt;
    if (fused) h.n_hits = n_hits;                     // (locate_kernel's counter)
    st.n_sa = h.n_sa + (h.n_hits - h.n_located);      // seed_kernel's reads for single-row matches + one per hit locate had to look up
    st.n_hits = h.n_hits; st.n_seedpos = h.n_seedpos;
    st.n_seeds = h.n_seeds; st.n_prefilter = h.n_kmer; st.n_occ_loaded = h.n_occ_loaded;
    st.n_lkt16 = h.n_lkt16; st.n_prefilter_loads = h.n_kmer_loads; st.n_text_words = h.n_text_words; st.n_steps_skipped = h.n_steps_skipped;
    st.n_sa_seed_kernel = h.n_sa;
    st.n_lf_steps = h.n_lf;
    float ms_over = 0;
    (void)hipEventElapsedTime(&st.ms_seed, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&ms_over, c->ev[8], c->ev[9]);     // items redone with the full-size slab belong to K1
    st.ms_seed += ms_over;
    (void)hipEventElapsedTime(&st.ms_locate, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&st.ms_merge, c->ev[3], c->ev[4]);
    c->last_seed = st; c->n_hits_total = n_hits; c->seeded = true;
    if (out_stats) *out_stats = st;
    return MP_OK;
}

extern "C" int mp_ctx_set_prefilter(mp_ctx *c, int on)
{
    if (!c) return MP_EINVAL;
    c->use_prefilter = on != 0;
    return MP_OK;
}

extern "C" int mp_ctx_set_dp_band(mp_ctx *c, int on)
{
    if (!c) return MP_EINVAL;
    c->use_band = on != 0;
    return MP_OK;
}

extern "C" int mp_ctx_set_dp_thermo(mp_ctx *c, int on)
{
    if (!c) return MP_EINVAL;
    c->use_thermo = on < 0 || on > 2 ? 1 : on;
    return MP_OK;
}

// instances the certificate kernel has seen since the context was created: [0] certified (no alignment possible), [1] a score of 12 or more
// somewhere besides the seed, [2] bound not below the cutoff, [3] not eligible (shape)
extern "C" int mp_ctx_dp_thermo_stats(mp_ctx *c, uint64_t out[4])
{
    if (!c || !out) return MP_EINVAL;
    for (int k = 0; k < 4; k++) out[k] = c->th_total[k];
    return MP_OK;
}

// test hook: the certificate kernel on caller-given instances of the reads now set.  kind: -1 by the longest read (narrow up to 160 bases, else wide),
// 0 reads of up to 100 bases (margin 25), 1 narrow, 2 wide, 3 mate-rescue windows / reads of up to 160 bases, 4 mate-rescue windows / up to 320
extern "C" int mp_test_dp_thermo_kind(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int kind, uint8_t *out_cert, uint64_t out_stats[4])
{
    if (!c || !tasks || !out_cert || !c->d_reads || kind < -1 || kind > 4) return MP_EINVAL;
    if (n == 0) return MP_OK;
    for (uint32_t t = 0; t < n; t++) {
        if (tasks[t].read_id >= c->n_reads || (tasks[t].strand != 1 && tasks[t].strand != 2)) return MP_EINVAL;
        if (tasks[t].win_len > MP_DP_MAX_WIN || tasks[t].win_start + tasks[t].win_len > c->ix->n) return MP_ERANGE;
    }
    if (kind < 0) kind = c->max_len <= MP_TH_MAXL ? 1 : 2;
    if (c->max_len > ((kind == 2 || kind == 4) ? MP_TH_MAXL_WIDE : MP_TH_MAXL)) return MP_ERANGE;
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    if (c->tasks_tmp.reserve(n) || c->th_cert.reserve(n + 64) || c->th_stats.reserve(4)) return MP_ENOMEM;
    HIPCHK(hipMemcpyAsync(c->tasks_tmp.p, tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->th_stats.p, 0, 4 * sizeof(unsigned long long), s));
    const dim3 g2(grid_for(n, 64)), g1(grid_for(n, 32)), blk(64);
#define MP_TH_ARGS c->ix->d.text, c->d_reads, c->d_lens, c->W, (const mp_dp_task *)c->tasks_tmp.p, n, c->th_cert.p, (const uint32_t *)nullptr, c->th_stats.p
    switch (kind) {
        case 0: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T, 32, MP_TH_MARGIN_SHORT, false, MP_TH_MAXM_SHORT>), g2, blk, 0, s, MP_TH_ARGS); break;
        case 1: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T, 32>), g2, blk, 0, s, MP_TH_ARGS); break;
        case 2: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_WIDE, 64>), g1, blk, 0, s, MP_TH_ARGS); break;
        case 3: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_RESCUE, 32, 0, true, MP_TH_MAXM_RESCUE>), g2, blk, 0, s, MP_TH_ARGS); break;
        default: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_RESCUE_WIDE, 64, 0, true, MP_TH_MAXM_RESCUE>), g1, blk, 0, s, MP_TH_ARGS); break;
    }
#undef MP_TH_ARGS
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_cert, c->th_cert.p, n, hipMemcpyDeviceToHost, s));
    unsigned long long ths[4];
    HIPCHK(hipMemcpyAsync(ths, c->th_stats.p, sizeof ths, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (out_stats) for (int k = 0; k < 4; k++) out_stats[k] = ths[k];
    return MP_OK;
}

extern "C" int mp_test_dp_thermo(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, uint8_t *out_cert, uint64_t out_stats[4])
{
    return mp_test_dp_thermo_kind(c, tasks, n, -1, out_cert, out_stats);
}

extern "C" int mp_ctx_set_dp_census(mp_ctx *c, int on)
{
    if (!c) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    if (on) {
        if (c->census.reserve(CENSUS_SLOTS)) return MP_ENOMEM;
        HIPCHK(hipMemsetAsync(c->census.p, 0, CENSUS_SLOTS * sizeof(unsigned long long), c->stream));
    }
    c->census_on = on != 0;
    return MP_OK;
}

extern "C" int mp_ctx_get_dp_census(mp_ctx *c, mp_dp_census *out)
{
    if (!c || !out || !c->census.p) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    unsigned long long h[CENSUS_SLOTS];
    HIPCHK(hipMemcpyAsync(h, c->census.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out->instances = h[CENSUS_INSTANCES]; out->cells_full_table = h[CENSUS_FULL_EQUIV]; out->cells_band_score = h[CENSUS_BAND_SCORE];
    out->cells_band_trace = h[CENSUS_BAND_TRACE]; out->cells_full_score = h[CENSUS_FULL_SCORE]; out->cells_full_trace = h[CENSUS_FULL_TRACE];
    return MP_OK;
}

extern "C" int mp_ctx_dp_band_stats(mp_ctx *c, uint64_t *out_n_band, uint64_t *out_n_full)
{
    if (!c) return MP_EINVAL;
    if (out_n_band) *out_n_band = c->n_band;
    if (out_n_full) *out_n_full = c->n_band_full;
    return MP_OK;
}

extern "C" int mp_get_seeds(mp_ctx *c, mp_seedpos *out_seeds, uint64_t cap, uint64_t *out_offsets)
{
    if (!c || !c->seeded || !out_offsets) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t n = c->n_reads;
    std::vector<uint32_t> base(2 * (size_t)n + 1), np(n), nn(n);
    HIPCHK(hipMemcpy(base.data(), c->rs_base.p, (2 * (size_t)n + 1) * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(np.data(), c->n_pos.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nn.data(), c->n_neg.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::vector<mp_seedpos> all(c->n_hits_total + 1);
    if (c->n_hits_total) HIPCHK(hipMemcpy(all.data(), c->seedpos.p, (size_t)c->n_hits_total * sizeof(mp_seedpos), hipMemcpyDeviceToHost));
    uint64_t w = 0;
    for (uint32_t r = 0; r < n; r++) {
        out_offsets[r] = w;
        uint32_t k = np[r] + nn[r];
        if (out_seeds) {
            if (w + k > cap) return MP_ERANGE;
            for (uint32_t j = 0; j < k; j++) out_seeds[w + j] = all[base[2 * (size_t)r] + j];
        }
        w += k;
    }
    out_offsets[n] = w;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ DP
namespace {

// adds the cells one kernel launch computes to the census (no-op unless the caller switched the census on)
void census_add(mp_ctx *c, int family, const mp_dp_task *d_tasks, const uint32_t *list, uint32_t n, uint32_t band_w)
{
    if (!c->census_on || n == 0 || !c->census.p) return;
    hipLaunchKernelGGL(dp_census_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, d_tasks, list, n, c->d_lens, band_w, c->census.p + family,
                       family == CENSUS_FULL_EQUIV ? c->census.p + CENSUS_INSTANCES : (unsigned long long *)nullptr);
    MP_TRACE(c->stream, "dp_census_kernel");
}

struct DpShape { int lpi, c, nw; };

inline int pick_shape(uint32_t max_len, DpShape &s)
{
    // LPI lanes per instance pair, C read columns per lane; 5 bits per cell, 3 cells per 16-bit half of a flag word.
    // Few lanes with many columns each amortise the per-row work (lane hand-over, window base, row bookkeeping) and
    // shorten the pipeline fill of the anti-diagonal sweep; 32 lanes only where 16 would need too many registers.
    static const int shapes[][2] = {{16, 5}, {16, 10}, {16, 12}, {32, 8}, {32, 10}, {32, 12}};
    static const bool wide = getenv("MP_DP_WIDE") != nullptr;       // measurement aid: the 32-lane shape for <= 160 bases
    if (wide && max_len <= 160) { s.lpi = 32; s.c = 5; s.nw = 2; return MP_OK; }
    for (auto &sh : shapes) if ((uint32_t)(sh[0] * sh[1]) >= max_len) { s.lpi = sh[0]; s.c = sh[1]; s.nw = (sh[1] + 2) / 3; return MP_OK; }
    return MP_ERANGE;   // reads longer than 384 bases (the reference's DP limit is ~312)
}

template <int LPI, int C>
void launch_fill(hipStream_t s, const mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int cl, int cr, uint32_t *flags, uint32_t flag_rows,
                 mp_fill_result *fill, bool count)
{
    const unsigned pairs = (n + 1) / 2, groups = MP_DP_BLOCK / LPI;
    if (count)
        hipLaunchKernelGGL((dp_fill_kernel<LPI, C, 1>), dim3(grid_for(pairs, groups)), dim3(MP_DP_BLOCK), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                           tasks, n, cl, cr, flags, flag_rows, fill);
    else
        hipLaunchKernelGGL((dp_fill_kernel<LPI, C, 0>), dim3(grid_for(pairs, groups)), dim3(MP_DP_BLOCK), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                           tasks, n, cl, cr, flags, flag_rows, fill);
    MP_TRACE(s, "dp_fill_kernel");
}
// instances one "round" of resident blocks of the fill kernel covers on this device: launches sized in whole rounds
// leave no partly filled last round (blocks of one launch take the same time, so they finish round by round)
template <int LPI, int C>
void fill_round(const mp_ctx *c, bool count, uint32_t *out)
{
    int per_cu = 0, n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->ix->device);
    hipError_t e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_fill_kernel<LPI, C, 1>, MP_DP_BLOCK, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_fill_kernel<LPI, C, 0>, MP_DP_BLOCK, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    *out = (uint32_t)n_cu * (uint32_t)per_cu * (uint32_t)(MP_DP_BLOCK / LPI) * 2u;
}

template <int LPI, int C>
void launch_score(hipStream_t s, const mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int cl, int cr, mp_fill_result *fill, const uint32_t *only = nullptr, uint32_t n_only = 0,
                  const uint8_t *skip = nullptr /* instances the certificate kernel decided (mixed-length batches: the lists are made per length class) */)
{
    if (only && c->max_len <= 8 * 20) {
        // the instances the certificate kernel left undecided (dp_thermo_kernel), by list
        if (c->max_len <= 8 * 19) hipLaunchKernelGGL((dp_score_kernel<8, 19>), dim3(grid_for((n_only + 1) / 2, MP_DP_BLOCK / 8)), dim3(MP_DP_BLOCK), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, tasks, n_only, fill, only);
        else hipLaunchKernelGGL((dp_score_kernel<8, 20>), dim3(grid_for((n_only + 1) / 2, MP_DP_BLOCK / 8)), dim3(MP_DP_BLOCK), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, tasks, n_only, fill, only);
        MP_TRACE(s, "dp_score_kernel (list)");
        return;
    }
    if (c->max_len <= 320) {
        // the leaner filter kernel (kernels_dp.hip): 8 lanes x 19 or 20 columns per instance pair for reads of up to 152 / 160 bases
        // (150-base reads then leave 2 idle columns instead of 10), 16 lanes x 12 / 16 / 20 columns up to 192 / 256 / 320 bases
#define MP_SCORE_LAUNCH(L_, C_, TASKS_N, LIST) hipLaunchKernelGGL((dp_score_kernel<L_, C_>), dim3(grid_for(((TASKS_N) + 1) / 2, MP_DP_BLOCK / L_)), dim3(MP_DP_BLOCK), 0, s, \
                                                   c->ix->d.text, c->d_reads, c->d_lens, c->W, tasks, (TASKS_N), fill, (LIST))
        const uint32_t *none = nullptr;
        if (c->max_len <= 8 * 19) MP_SCORE_LAUNCH(8, 19, n, none);
        else if (c->max_len <= 8 * 20) MP_SCORE_LAUNCH(8, 20, n, none);
        else {
            // a batch with longer reads is usually a batch of mixed lengths (trimmed reads): one launch per length class, each with the
            // kernel of its width, instead of the widest kernel for all
            mp_ctx *cc = const_cast<mp_ctx *>(c);
            const uint32_t bounds[5] = {0u, 152u, 192u, 256u, 320u};
            auto by_class = [&]() -> bool {
                for (int k = 0; k < 4; k++) {
                    if (c->max_len <= bounds[k]) break;
                    if (cc->sel_flag.reserve(n + 2) || cc->sel_idx.reserve(n + 2)) return false;
                    hipLaunchKernelGGL(dp_len_class_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, tasks, c->d_lens, n, bounds[k], bounds[k + 1], cc->sel_flag.p, skip);
                    size_t tmp_bytes = 0;
                    (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, cc->sel_flag.p, cc->sel_idx.p, (int)n, s);
                    if (cc->cub_tmp.reserve(tmp_bytes + 256)) return false;
                    (void)hipcub::DeviceScan::InclusiveSum(cc->cub_tmp.p, tmp_bytes, cc->sel_flag.p, cc->sel_idx.p, (int)n, s);
                    uint32_t n_k = 0;
                    if (hipMemcpyAsync(&n_k, cc->sel_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
                    if (n_k == 0) continue;
                    if (cc->score_list.reserve(n_k + 2)) return false;
                    hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, cc->sel_flag.p, cc->sel_idx.p, n, cc->score_list.p, none);
                    const uint32_t *lst = cc->score_list.p;
                    if (k == 0) MP_SCORE_LAUNCH(8, 19, n_k, lst);
                    else if (k == 1) MP_SCORE_LAUNCH(16, 12, n_k, lst);
                    else if (k == 2) MP_SCORE_LAUNCH(16, 16, n_k, lst);
                    else MP_SCORE_LAUNCH(16, 20, n_k, lst);
                    if (hipStreamSynchronize(s) != hipSuccess) return false;        // the list buffer is reused by the next class
                }
                return true;
            };
            if (!by_class()) MP_SCORE_LAUNCH(16, 20, n, none);                       // no memory for the lists: the widest kernel over everything
        }
#undef MP_SCORE_LAUNCH
        MP_TRACE(s, "dp_score_kernel");
        return;
    }
    const unsigned pairs = (n + 1) / 2, groups = MP_DP_BLOCK / LPI;
    hipLaunchKernelGGL((dp_fill_kernel<LPI, C, 2>), dim3(grid_for(pairs, groups)), dim3(MP_DP_BLOCK), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                       tasks, n, cl, cr, (uint32_t *)nullptr, 0u, fill);
    MP_TRACE(s, "dp_fill_kernel");
}

template <int LPI, int C>
void launch_trace(hipStream_t s, const mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int cl, int cr, const uint32_t *flags, uint32_t flag_rows,
                  const mp_fill_result *fill, mp_dp_result *res, uint8_t *ops, uint32_t ops_stride)
{
    hipLaunchKernelGGL((dp_trace_kernel<LPI, C>), dim3(grid_for(n, 64)), dim3(64), 0, s, tasks, c->d_lens, n, cl, cr, flags, flag_rows, fill, res, ops, ops_stride);
    MP_TRACE(s, "dp_trace_kernel");
}

#define MP_DP_DISPATCH(FN, ...)                                   \
    switch (shape.lpi * 100 + shape.c) {                          \
        case 1605: FN<16, 5>(__VA_ARGS__); break;                 \
        case 3205: FN<32, 5>(__VA_ARGS__); break;                 \
        case 1610: FN<16, 10>(__VA_ARGS__); break;                \
        case 1612: FN<16, 12>(__VA_ARGS__); break;                \
        case 3208: FN<32, 8>(__VA_ARGS__); break;                 \
        case 3210: FN<32, 10>(__VA_ARGS__); break;                \
        default: FN<32, 12>(__VA_ARGS__); break;                  \
    }

// Runs fill + traceback for device-resident tasks, chunked so that the flag slab stays bounded.
int run_dp(mp_ctx *c, const mp_dp_task *d_tasks, uint32_t n, uint32_t max_win, int cl, int cr, mp_dp_result *d_res, uint8_t *d_ops,
           uint32_t ops_stride, float *ms_fill, float *ms_trace, bool count)
{
    if (n == 0) return MP_OK;
    DpShape shape;
    if (pick_shape(c->max_len, shape)) return MP_ERANGE;
    if (max_win > MP_DP_MAX_WIN) return MP_ERANGE;
    hipStream_t s = c->stream;
    if (g_trace_kernels) {
        std::vector<mp_dp_task> ht(n);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(ht.data(), d_tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyDeviceToHost);
        uint32_t bad = 0;
        for (uint32_t q = 0; q < n; q++)
            if (ht[q].read_id >= c->n_reads || ht[q].win_start + ht[q].win_len > c->ix->n || ht[q].win_len > MP_DP_MAX_WIN) {
                if (bad++ < 4) fprintf(stderr, "[megapath_hip] run_dp: BAD task %u of %u: read %u strand %u start %llu len %u (text %llu, tasks %p)\n", q, n, ht[q].read_id,
                                       ht[q].strand, (unsigned long long)ht[q].win_start, ht[q].win_len, (unsigned long long)c->ix->n, (const void *)d_tasks);
            }
        fprintf(stderr, "[megapath_hip] run_dp: %u tasks, %u bad\n", n, bad);
        if (bad) return MP_ERANGE;
    }
    const uint32_t flag_rows = (max_win + (uint32_t)shape.lpi + MP_DP_FLUSH - 1) / MP_DP_FLUSH * MP_DP_FLUSH;   // whole 8-step runs
    const size_t per_pair = (size_t)flag_rows * (size_t)shape.lpi * shape.nw * 4;
    // traceback slab: a quarter of the free HBM, between 2 and 24 GiB (with a 260 GB index replica little is left; with a
    // human-sized index the launches become long enough that their tails do not matter)
    if (c->slab_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 8ull << 30;
        c->slab_budget = std::min<size_t>(24ull << 30, std::max<size_t>(2ull << 30, (free_b + c->flags.cap) / 4));
    }
    // instances per launch: everything when the slab holds it (an odd n must not be cut into n - 1 and 1: the second launch would
    // cost a whole table's latency for one instance), else an even number (two instances share a lane group)
    uint32_t chunk = (uint32_t)std::min<uint64_t>(n, 2 * std::max<uint64_t>(512, c->slab_budget / per_pair));
    if (chunk < n) { chunk &= ~1u; if (chunk < 2) chunk = 2; }
    uint32_t round = 0;
    MP_DP_DISPATCH(fill_round, c, count, &round);
    if (round >= 2 && chunk > round && chunk < n) chunk = chunk / round * round;      // whole rounds of resident blocks per launch
    if (c->flags.reserve(((size_t)chunk / 2 + 1) * per_pair) || c->fill.reserve(chunk + 2)) return MP_ENOMEM;
    census_add(c, CENSUS_FULL_TRACE, d_tasks, nullptr, n, 0);
    float tf = 0, tt = 0;
    const size_t n_chunks = (n + chunk - 1) / chunk;
    while (c->ev_pool.size() < 3 * n_chunks) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    size_t ci = 0;
    for (uint32_t off = 0; off < n; off += chunk, ci++) {
        uint32_t cnt = std::min(chunk, n - off);
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci], s));
        MP_DP_DISPATCH(launch_fill, s, c, d_tasks + off, cnt, cl, cr, (uint32_t *)c->flags.p, flag_rows, c->fill.p, count);
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 1], s));
        uint8_t *ops_here = d_ops ? d_ops + (size_t)off * ops_stride : nullptr;
        MP_DP_DISPATCH(launch_trace, s, c, d_tasks + off, cnt, cl, cr, (const uint32_t *)c->flags.p, flag_rows, c->fill.p, d_res + off, ops_here, ops_stride);
        if (ops_here)
            hipLaunchKernelGGL(dp_ops_fix_kernel, dim3(grid_for(cnt, 64)), dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks + off, cnt,
                               c->fill.p, d_res + off, ops_here, ops_stride);
            MP_TRACE(s, "dp_ops_fix_kernel");
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 2], s));
    }
    HIPCHK(hipStreamSynchronize(s));
    for (size_t q = 0; q < n_chunks; q++) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, c->ev_pool[3 * q], c->ev_pool[3 * q + 1]);
        (void)hipEventElapsedTime(&b, c->ev_pool[3 * q + 1], c->ev_pool[3 * q + 2]);
        tf += a; tt += b;
    }
    if (ms_fill) *ms_fill += tf;
    if (ms_trace) *ms_trace += tt;
    return MP_OK;
}

// For batches where most instances cannot reach their cutoff (the rescue passes): one score-only pass over all of them
// (no traceback bits, no slab, one launch), then fill + traceback of the few that can pass.  Same results as run_dp.
static int exclusive_scan_inclusive_u32(mp_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n)
{
    size_t tmp_bytes = 0;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, in, out, (int)n, c->stream) != hipSuccess) return MP_ENODEV;
    if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
    if (hipcub::DeviceScan::InclusiveSum(c->cub_tmp.p, tmp_bytes, in, out, (int)n, c->stream) != hipSuccess) return MP_ENODEV;
    return MP_OK;
}

bool band_applicable(const mp_ctx *c, uint32_t max_win, int cl, int cr);
int run_dp_band(mp_ctx *c, const mp_dp_task *d_tasks, uint32_t n, uint32_t max_win, int cl, int cr, mp_dp_result *d_res, float *ms_fill, float *ms_trace, bool count);

int run_dp_filtered(mp_ctx *c, const mp_dp_task *d_tasks, uint32_t n, uint32_t max_win, int cl, int cr, mp_dp_result *d_res,
                    float *ms_fill, float *ms_trace, bool count, bool rescue_windows = false /* mate-rescue windows: no seed inside (HalfEndAlgnBatch) */)
{
    if (n == 0) return MP_OK;
    DpShape shape;
    if (pick_shape(c->max_len, shape)) return MP_ERANGE;
    if (max_win > MP_DP_MAX_WIN) return MP_ERANGE;
    hipStream_t s = c->stream;
    if (c->fill.reserve(n + 2) || c->sel_flag.reserve(n + 2) || c->sel_idx.reserve(n + 2)) return MP_ENOMEM;
    while (c->ev_pool.size() < 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    // Most instances of the rescue passes hold nothing: windows around a chance seed (read + 2 x margin), windows of a mate that is not there
    // (mate rescue, up to 751 rows).  dp_thermo_kernel certifies that for 32 instances per word (kernels_thermo.hip); only what it leaves
    // undecided goes through the score-only filter.  One variant per shape, each over the list of its length class:
    //   windows around a seed   reads of 30 .. 100 bases (margin 25) | 101 .. 160 (margin 30) | 161 .. 320 (margin 30, one set of 32 instances per wavefront)
    //   mate-rescue windows     reads of 30 .. 160 bases | 161 .. 320, windows of up to 768 rows, no seed
    static const bool thermo_env = !getenv("MP_DP_THERMO") || atoi(getenv("MP_DP_THERMO")) != 0;
    static const bool thermo_more_env = !getenv("MP_DP_THERMO_MORE") || atoi(getenv("MP_DP_THERMO_MORE")) != 0;      // 0: round 3's coverage only (test / measurement switch)
    const bool thermo_more = thermo_more_env && c->use_thermo == 1;
    struct ThClass { uint32_t lo, hi; int kind; };                                     // kind: 0 short, 1 narrow, 2 wide, 3 rescue narrow, 4 rescue wide
    ThClass cls[3];
    int n_cls = 0;
    if (!rescue_windows) {
        if (thermo_more && c->min_len <= MP_TH_MAXL_SHORT && max_win <= MP_TH_MAXM_WIDE) cls[n_cls++] = ThClass{(uint32_t)MP_MIN_ALIGNABLE_LEN - 1, MP_TH_MAXL_SHORT, 0};
        if (c->max_len > 100 && max_win <= MP_TH_MAXM_WIDE) cls[n_cls++] = ThClass{100u, MP_TH_MAXL, 1};
        if (c->max_len > MP_TH_MAXL && c->max_len <= MP_TH_MAXL_WIDE && max_win <= MP_TH_MAXM_WIDE) cls[n_cls++] = ThClass{MP_TH_MAXL, MP_TH_MAXL_WIDE, 2};
    } else if (thermo_more && max_win <= MP_TH_MAXM_RESCUE && c->max_len <= MP_TH_MAXL_WIDE) {
        // Where the mate IS there (most pairs of a clean library) the certificate finds a plane-T bit and the filter runs all the same: the variant
        // pays only for read sets with many mates that are not (broken pairs, chimeras, contamination).  Whether it does is a property of the read
        // set, so the context watches the share of instances the last trial certified: below a third it sits out the next 15 calls, then tries again.
        // (Results do not depend on it: a certificate only ever replaces a filter call that would have said "no".)
        const char *sw = getenv("MP_DP_THERMO_RESCUE");
        const bool always = sw && atoi(sw) == 2;                                           // test switch: no sitting out, no minimum size
        const bool sit_out = !always && c->th_rescue_skip > 0;
        if (sit_out) c->th_rescue_skip--;
        else if (n >= 65536 || always) {
            // (a launch of this variant takes ~0.45 ms however few instances it gets -- 782 dependent steps per wavefront --, the filter 21 ns per instance: below
            // some ten thousand instances the filter alone is faster even where every instance would be certified)
            cls[n_cls++] = ThClass{(uint32_t)MP_MIN_ALIGNABLE_LEN - 1, MP_TH_MAXL, 3};
            if (c->max_len > MP_TH_MAXL) cls[n_cls++] = ThClass{MP_TH_MAXL, MP_TH_MAXL_WIDE, 4};
        }
    }
    // every instance in ONE class and no list needed: the single-end windows of a batch of 101 .. 160-base reads (the NT stage's shape)
    const bool one_launch = !rescue_windows && n_cls == 1 && cls[0].kind == 1 && c->min_len > 100 && c->max_len <= MP_TH_MAXL && max_win <= MP_TH_MAXM;
    const bool use_list = c->max_len <= 8 * 20;                                        // the filter takes the undecided instances by list (dp_score_kernel<8, C>); otherwise per length class with `skip`
    const bool thermo = thermo_env && c->use_thermo != 0 && n_cls > 0 && n >= 64;
    HIPCHK(hipEventRecord(c->ev_pool[0], s));
    if (thermo) {
        if (c->th_cert.reserve(n + 64) || c->th_stats.reserve(4) || c->score_list.reserve(n + 2)) return MP_ENOMEM;
        HIPCHK(hipMemsetAsync(c->th_stats.p, 0, 4 * sizeof(unsigned long long), s));
        auto launch_class = [&](int kind, uint32_t n_k, const uint32_t *lst) {
            const dim3 g2(grid_for(n_k, 64)), g1(grid_for(n_k, 32)), blk(64);
#define MP_TH_ARGS c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n_k, c->th_cert.p, lst, c->th_stats.p
            switch (kind) {
                case 0: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T, 32, MP_TH_MARGIN_SHORT, false, MP_TH_MAXM_SHORT>), g2, blk, 0, s, MP_TH_ARGS); break;
                case 1: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T, 32>), g2, blk, 0, s, MP_TH_ARGS); break;
                case 2: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_WIDE, 64>), g1, blk, 0, s, MP_TH_ARGS); break;
                case 3: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_RESCUE, 32, 0, true, MP_TH_MAXM_RESCUE>), g2, blk, 0, s, MP_TH_ARGS); break;
                default: hipLaunchKernelGGL((dp_thermo_kernel<MP_TH_T_RESCUE_WIDE, 64, 0, true, MP_TH_MAXM_RESCUE>), g1, blk, 0, s, MP_TH_ARGS); break;
            }
#undef MP_TH_ARGS
        };
        if (one_launch) launch_class(1, n, (const uint32_t *)nullptr);
        else {
            // mixed lengths (trimmed reads) or a shape with its own variant: each class as a list (a wavefront then sweeps as many rows as ITS
            // longest window has); what belongs to no class is left to the filter
            HIPCHK(hipMemsetAsync(c->th_cert.p, 0, n, s));
            for (int k = 0; k < n_cls; k++) {
                hipLaunchKernelGGL(dp_len_class_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_tasks, c->d_lens, n, cls[k].lo, cls[k].hi, c->sel_flag.p, (const uint8_t *)nullptr);
                int rck = exclusive_scan_inclusive_u32(c, c->sel_flag.p, c->sel_idx.p, n);
                if (rck) return rck;
                uint32_t n_k = 0;
                HIPCHK(hipMemcpyAsync(&n_k, c->sel_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (n_k == 0) continue;
                hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->sel_flag.p, c->sel_idx.p, n, c->score_list.p, (const uint32_t *)nullptr);
                const uint32_t *lst = c->score_list.p;
                static const bool sort_lists = !getenv("MP_DP_THERMO_SORT") || atoi(getenv("MP_DP_THERMO_SORT")) != 0;     // 0: measurement switch
                if (sort_lists && cls[k].kind <= 2 && n_k >= 4096) {
                    // windows around a seed are read + 2 x margin long: in a class of mixed read lengths the list goes out sorted by window length, so
                    // that a wavefront's instances need about the same number of rows (the flag / scan buffers are free again: keys in, keys out)
                    if (c->und_list[0].reserve(n_k + 2)) return MP_ENOMEM;
                    hipLaunchKernelGGL(thermo_window_key_kernel, dim3(grid_for(n_k, 256)), dim3(256), 0, s, d_tasks, (const uint32_t *)c->score_list.p, n_k, c->sel_flag.p);
                    size_t tb = 0;
                    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const uint32_t *)c->sel_flag.p, c->sel_idx.p, (const uint32_t *)c->score_list.p, c->und_list[0].p, (int)n_k, 0, 10, s) != hipSuccess) return MP_ENODEV;
                    if (c->cub_tmp.reserve(tb + 256)) return MP_ENOMEM;
                    if (hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, (const uint32_t *)c->sel_flag.p, c->sel_idx.p, (const uint32_t *)c->score_list.p, c->und_list[0].p, (int)n_k, 0, 10, s) != hipSuccess) return MP_ENODEV;
                    lst = c->und_list[0].p;
                }
                launch_class(cls[k].kind, n_k, lst);
                HIPCHK(hipStreamSynchronize(s));                                 // the list buffers are reused by the next class
            }
        }
        MP_TRACE(s, "dp_thermo_kernel");
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(thermo_flag_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint8_t *)c->th_cert.p, n, c->sel_flag.p, c->fill.p);
        int rc0 = exclusive_scan_inclusive_u32(c, c->sel_flag.p, c->sel_idx.p, n);
        if (rc0) return rc0;
        uint32_t n_und = 0;
        unsigned long long ths[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(&n_und, c->sel_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(ths, c->th_stats.p, sizeof ths, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int k = 0; k < 4; k++) c->th_total[k] += ths[k];
        if (rescue_windows && 3 * ths[0] < (unsigned long long)n) c->th_rescue_skip = 15;      // (see above)
        if (n_und) {
            hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->sel_flag.p, c->sel_idx.p, n, c->score_list.p, (const uint32_t *)nullptr);
            census_add(c, CENSUS_FULL_SCORE, d_tasks, c->score_list.p, n_und, 0);
            if (use_list) { MP_DP_DISPATCH(launch_score, s, c, d_tasks, n, cl, cr, c->fill.p, (const uint32_t *)c->score_list.p, n_und); }
            else { MP_DP_DISPATCH(launch_score, s, c, d_tasks, n, cl, cr, c->fill.p, (const uint32_t *)nullptr, 0u, (const uint8_t *)c->th_cert.p); }
        }
    } else {
        census_add(c, CENSUS_FULL_SCORE, d_tasks, nullptr, n, 0);
        MP_DP_DISPATCH(launch_score, s, c, d_tasks, n, cl, cr, c->fill.p);
    }
    HIPCHK(hipEventRecord(c->ev_pool[1], s));
    hipLaunchKernelGGL(dp_select_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_tasks, c->fill.p, c->d_lens, n, c->sel_flag.p, d_res);
    MP_TRACE(s, "dp_select_kernel");
    {
        size_t tmp_bytes = 0;
        (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, c->sel_flag.p, c->sel_idx.p, (int)n, s);
        if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
        (void)hipcub::DeviceScan::InclusiveSum(c->cub_tmp.p, tmp_bytes, c->sel_flag.p, c->sel_idx.p, (int)n, s);
    }
    uint32_t n_sel = 0;
    HIPCHK(hipMemcpyAsync(&n_sel, c->sel_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float t_score = 0;
    (void)hipEventElapsedTime(&t_score, c->ev_pool[0], c->ev_pool[1]);
    if (ms_fill) *ms_fill += t_score;
    if (n_sel == 0) return MP_OK;
    if (c->sel_tasks.reserve(n_sel + 2) || c->sel_res.reserve(n_sel + 2)) return MP_ENOMEM;
    hipLaunchKernelGGL(dp_compact_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_tasks, c->sel_flag.p, c->sel_idx.p, n, c->sel_tasks.p);
    MP_TRACE(s, "dp_compact_kernel");
    // What passes the filter are mostly real alignments with few differences: windows of read + 60 go through the band scheme (perfect and
    // one-mismatch diagonals need no table, most others 15 or 27 diagonals), anything else (mate-rescue windows) through the full table.
    // The scheme is a dozen dependent launches with host round trips between them: measured on the default step it costs 1 ms MORE than the
    // full table for 66 k instances and wins from roughly twice that on (rescue-heavy batches: C5 3.5 -> 3.9 M pairs/s), hence the threshold.
    // MP_DP_RESCUE_BAND = 0: never; = n: from n instances on (1 = always: the soak's setting).
    static const uint32_t band_from = getenv("MP_DP_RESCUE_BAND") ? (uint32_t)atoi(getenv("MP_DP_RESCUE_BAND")) : 160000u;
    int rc = band_from && n_sel >= band_from && band_applicable(c, max_win, cl, cr) ? run_dp_band(c, c->sel_tasks.p, n_sel, max_win, cl, cr, c->sel_res.p, ms_fill, ms_trace, count)
                                                            : run_dp(c, c->sel_tasks.p, n_sel, max_win, cl, cr, c->sel_res.p, nullptr, 0, ms_fill, ms_trace, count);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->sel_flag.p, c->sel_idx.p, c->sel_res.p, n, d_res);
    MP_TRACE(s, "dp_scatter_kernel");
    return MP_OK;
}

// Deep-DP batches (windows = read + 2 x DP2_MARGIN) without op strings: the band kernel decides every
// instance whose best alignment provably stays inside the band (kernels_dp_band.hip), the rest go through the
// score-only pre-pass and the full kernels.  Same results as run_dp.
bool band_applicable(const mp_ctx *c, uint32_t max_win, int cl, int cr)
{
    static const bool off = getenv("MP_DP_BAND") && atoi(getenv("MP_DP_BAND")) == 0;
    return !off && c->use_band && c->max_len <= MP_BAND_MAXL && max_win <= MP_BAND_MAXL + MP_BAND_DMAX && cl >= MP_BAND_GB && cr >= MP_BAND_GB;
}

// second pass of the two-pass band scheme over the instances of one class (`list`, n_c of them) + their traceback
template <int W2, int G2>
int band_second_pass(mp_ctx *c, const mp_dp_task *d_tasks, const uint32_t *list, uint32_t n_c, int cl, int cr, mp_dp_result *d_res, size_t *ev_cursor)
{
    // no host synchronisation here: the event triples (start, after fill, after traceback) are taken from *ev_cursor on
    // and read by the caller once the stream has drained
    if (n_c == 0) return MP_OK;
    hipStream_t s = c->stream;
    census_add(c, CENSUS_BAND_TRACE, d_tasks, list, n_c, W2);
    const uint32_t ncols = c->max_len;
    const bool small = c->max_len <= 160;
    const size_t per_block = (size_t)ncols * ((W2 + 2) / 3) * 64 * 4;
    const uint32_t n_blocks = (uint32_t)(((uint64_t)n_c + 127) / 128);
    const uint32_t chunk_blocks = (uint32_t)std::min<uint64_t>(n_blocks, std::max<uint64_t>(1, c->slab_budget / per_block));
    if (c->flags.reserve((size_t)chunk_blocks * per_block)) return MP_ENOMEM;
    const size_t n_chunks = (n_blocks + chunk_blocks - 1) / chunk_blocks;
    while (c->ev_pool.size() < *ev_cursor + 3 * n_chunks) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    uint32_t *slab = (uint32_t *)c->flags.p;
    size_t ci = *ev_cursor / 3;
    *ev_cursor += 3 * n_chunks;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += chunk_blocks, ci++) {
        const uint32_t off = b0 * 128u;
        const uint32_t cnt = (uint32_t)std::min<uint64_t>((uint64_t)chunk_blocks * 128u, n_c - off);
        const dim3 grid((cnt + 127) / 128), block(64);
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci], s));
        if (small)
            hipLaunchKernelGGL((dp_band_kernel<W2, G2, 160, 2>), grid, block, 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, cnt, cl, cr, slab, ncols,
                               (const mp_fill_result *)c->band_fill.p, (mp_fill_result *)nullptr, list + off);
        else
            hipLaunchKernelGGL((dp_band_kernel<W2, G2, MP_BAND_MAXL, 2>), grid, block, 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, cnt, cl, cr, slab,
                               ncols, (const mp_fill_result *)c->band_fill.p, (mp_fill_result *)nullptr, list + off);
        MP_TRACE(s, "dp_band_kernel");
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 1], s));
        hipLaunchKernelGGL((dp_trace_kernel<16, 5, W2, G2, true>), dim3(grid_for(cnt, 64)), dim3(64), 0, s, d_tasks, c->d_lens, cnt, cl, cr, (const uint32_t *)slab,
                           ncols, (const mp_fill_result *)c->band_fill.p, d_res, (uint8_t *)nullptr, 0u, list + off);
        MP_TRACE(s, "dp_trace_kernel");
        HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 2], s));
    }
    return MP_OK;
}

// One tier of the two-pass band scheme over the instances `in_list` (nullptr: all n_in instances of d_tasks):
// block-match pre-filter with budget GBUD, first pass over 2 GBUD + 1 diagonals for the instances it confines (TIER 1 also
// over the whole band for the ones it cannot), classification, second pass by class with traceback.  Results of decided
// instances land in d_res; the undecided ones are listed in und_out (*n_und of them).
template <int GBUD, int MINB, bool TIER1>
int band_tier(mp_ctx *c, const mp_dp_task *d_tasks, const uint32_t *in_list, uint32_t n_in, int cl, int cr, mp_dp_result *d_res, float *ms_fill,
              float *ms_trace, DevBuf<uint32_t> &und_out, uint32_t *n_und)
{
    *n_und = 0;
    if (n_in == 0) return MP_OK;
    hipStream_t s = c->stream;
    const bool small = c->max_len <= 160;
    const mp_fill_result *none = nullptr;
    const uint32_t n = n_in;
    if (c->pre_cls.reserve(n + 2) || c->near_flag.reserve(n + 2) || c->near_idx.reserve(n + 2) || c->far_flag.reserve(n + 2) || c->far_idx.reserve(n + 2) ||
        c->vfar_flag.reserve(n + 2) || c->vfar_idx.reserve(n + 2) || c->mid_flag.reserve(n + 2) || c->mid_idx.reserve(n + 2) ||
        c->band_flag.reserve(n + 2) || c->band_idx.reserve(n + 2))
        return MP_ENOMEM;
    while (c->ev_pool.size() < 4) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
    auto scan = [&](uint32_t *flag, uint32_t *idx) -> int {
        size_t tmp_bytes = 0;
        (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, flag, idx, (int)n, s);
        if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
        (void)hipcub::DeviceScan::InclusiveSum(c->cub_tmp.p, tmp_bytes, flag, idx, (int)n, s);
        return MP_OK;
    };
    HIPCHK(hipEventRecord(c->ev_pool[0], s));
    // block-match pre-filter: finished (exact diagonal), narrow first pass, whole band, or no qualifying alignment at all
    const int one_mm = getenv("MP_DP_ONE_MISMATCH") ? atoi(getenv("MP_DP_ONE_MISMATCH")) : 2;      // 0, 1: test switch (kernels_dp_band.hip)
    if (small)
        hipLaunchKernelGGL((dp_prefilter_kernel<GBUD, MINB, 160>), dim3(grid_for(n, 64)), dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n,
                           c->band_fill.p, c->pre_cls.p, in_list, one_mm);
    else
        hipLaunchKernelGGL((dp_prefilter_kernel<GBUD, MINB, MP_BAND_MAXL>), dim3(grid_for(n, 64)), dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n,
                           c->band_fill.p, c->pre_cls.p, in_list, one_mm);
    MP_TRACE(s, "dp_prefilter_kernel");
    hipLaunchKernelGGL(dp_class_flag_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint32_t *)c->pre_cls.p, n, MP_PRE_NARROW, c->near_flag.p);
    MP_TRACE(s, "dp_class_flag_kernel");
    hipLaunchKernelGGL(dp_class_flag_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint32_t *)c->pre_cls.p, n, MP_PRE_WIDE, c->far_flag.p);
    MP_TRACE(s, "dp_class_flag_kernel");
    if (TIER1) { hipLaunchKernelGGL(dp_class_flag_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint32_t *)c->pre_cls.p, n, MP_PRE_NARROW15, c->mid_flag.p); MP_TRACE(s, "dp_class_flag_kernel"); }
    { int rc0 = scan(c->near_flag.p, c->near_idx.p); if (rc0) return rc0; rc0 = scan(c->far_flag.p, c->far_idx.p); if (rc0) return rc0;
      if (TIER1) { rc0 = scan(c->mid_flag.p, c->mid_idx.p); if (rc0) return rc0; } }
    uint32_t n_narrow = 0, n_wide = 0, n_narrow15 = 0;
    HIPCHK(hipMemcpyAsync(&n_narrow, c->near_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&n_wide, c->far_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    if (TIER1) HIPCHK(hipMemcpyAsync(&n_narrow15, c->mid_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (c->narrow_list.reserve(n_narrow + 130) || c->wide_list.reserve(n_wide + 130) || c->narrow15_list.reserve(n_narrow15 + 130)) return MP_ENOMEM;
    if (n_narrow15) {
        // a diagonal with at most two mismatches is known: 15 diagonals decide the instance (dp_prefilter_kernel)
        hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->mid_flag.p, c->mid_idx.p, n, c->narrow15_list.p, in_list);
        MP_TRACE(s, "dp_band_list_kernel");
        const dim3 g15((n_narrow15 + 127) / 128);
        census_add(c, CENSUS_BAND_SCORE, d_tasks, c->narrow15_list.p, n_narrow15, MP_BAND_WM);
        if (small)
            hipLaunchKernelGGL((dp_band_kernel<MP_BAND_WM, MP_BAND_GBM, 160, 3>), g15, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n_narrow15, cl, cr,
                               (uint32_t *)nullptr, 0u, (const mp_fill_result *)c->band_fill.p, c->band_fill.p, (const uint32_t *)c->narrow15_list.p);
        else
            hipLaunchKernelGGL((dp_band_kernel<MP_BAND_WM, MP_BAND_GBM, MP_BAND_MAXL, 3>), g15, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n_narrow15,
                               cl, cr, (uint32_t *)nullptr, 0u, (const mp_fill_result *)c->band_fill.p, c->band_fill.p, (const uint32_t *)c->narrow15_list.p);
        MP_TRACE(s, "dp_band_kernel");
    }
    if (n_narrow) {
        hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->near_flag.p, c->near_idx.p, n, c->narrow_list.p, in_list);
        MP_TRACE(s, "dp_band_list_kernel");
        const dim3 g1((n_narrow + 127) / 128);
        census_add(c, CENSUS_BAND_SCORE, d_tasks, c->narrow_list.p, n_narrow, 2 * GBUD + 1);
        if (small)
            hipLaunchKernelGGL((dp_band_kernel<2 * GBUD + 1, GBUD, 160, 3>), g1, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n_narrow, cl, cr,
                               (uint32_t *)nullptr, 0u, (const mp_fill_result *)c->band_fill.p, c->band_fill.p, (const uint32_t *)c->narrow_list.p);
        else
            hipLaunchKernelGGL((dp_band_kernel<2 * GBUD + 1, GBUD, MP_BAND_MAXL, 3>), g1, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, d_tasks, n_narrow,
                               cl, cr, (uint32_t *)nullptr, 0u, (const mp_fill_result *)c->band_fill.p, c->band_fill.p, (const uint32_t *)c->narrow_list.p);
        MP_TRACE(s, "dp_band_kernel");
    }
    // whole-band instances (repeats inside the window, reads too short for the block test): first tier only, and only when
    // they are worth a launch of their own -- otherwise they stay undecided (the pre-filter left score -1)
    if (TIER1 && n_wide >= 1024) {
        hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->far_flag.p, c->far_idx.p, n, c->wide_list.p, in_list);
        MP_TRACE(s, "dp_band_list_kernel");
        const dim3 g1((n_wide + 127) / 128);
        census_add(c, CENSUS_BAND_SCORE, d_tasks, c->wide_list.p, n_wide, MP_BAND_W);
        if (small)
            hipLaunchKernelGGL((dp_band_kernel<MP_BAND_W, MP_BAND_GB, 160, 1>), g1, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                               d_tasks, n_wide, cl, cr, (uint32_t *)nullptr, 0u, none, c->band_fill.p, (const uint32_t *)c->wide_list.p);
        else
            hipLaunchKernelGGL((dp_band_kernel<MP_BAND_W, MP_BAND_GB, MP_BAND_MAXL, 1>), g1, dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                               d_tasks, n_wide, cl, cr, (uint32_t *)nullptr, 0u, none, c->band_fill.p, (const uint32_t *)c->wide_list.p);
        MP_TRACE(s, "dp_band_kernel");
    }
    static const bool verbose2 = getenv("MP_DP_BAND_STATS") != nullptr;
    if (verbose2) fprintf(stderr, "[band] tier %d pre-filter: %u narrow (15 diagonals), %u narrow, %u whole band, of %u\n", TIER1 ? 1 : 2, n_narrow15, n_narrow, n_wide, n);
    HIPCHK(hipEventRecord(c->ev_pool[1], s));
    // classes by distance from the best cell's diagonal; perfect matches, plain diagonal alignments (mismatches only; MP_DP_DIAG=0
    // sends them through the second pass like the rest: test switch) and instances below their cutoff are finished here
    const bool diag_shortcut = !(getenv("MP_DP_DIAG") && atoi(getenv("MP_DP_DIAG")) == 0);
    hipLaunchKernelGGL(dp_band_classify_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, d_tasks, c->d_lens, (const mp_fill_result *)c->band_fill.p, n,
                       c->near_flag.p, c->far_flag.p, c->band_flag.p, d_res, in_list, TIER1 ? (uint32_t *)nullptr : c->vfar_flag.p, c->mid_flag.p,
                       diag_shortcut ? c->ix->d.text : (const uint32_t *)nullptr, (const uint32_t *)c->d_reads, c->W);
    MP_TRACE(s, "dp_band_classify_kernel");
    int rc = scan(c->near_flag.p, c->near_idx.p); if (rc) return rc;
    rc = scan(c->far_flag.p, c->far_idx.p); if (rc) return rc;
    rc = scan(c->band_flag.p, c->band_idx.p); if (rc) return rc;
    rc = scan(c->mid_flag.p, c->mid_idx.p); if (rc) return rc;
    if (!TIER1) { rc = scan(c->vfar_flag.p, c->vfar_idx.p); if (rc) return rc; }
    uint32_t n_near = 0, n_far = 0, n_vfar = 0, n_mid = 0, n_sel = 0;
    HIPCHK(hipMemcpyAsync(&n_near, c->near_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&n_far, c->far_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&n_sel, c->band_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&n_mid, c->mid_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    if (!TIER1) HIPCHK(hipMemcpyAsync(&n_vfar, c->vfar_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    { float a = 0; (void)hipEventElapsedTime(&a, c->ev_pool[0], c->ev_pool[1]); if (ms_fill) *ms_fill += a; }
    if (c->near_list.reserve(n_near + 130) || c->far_list.reserve(n_far + 130) || c->vfar_list.reserve(n_vfar + 130) || c->mid_list.reserve(n_mid + 130) ||
        und_out.reserve(n_sel + 130)) return MP_ENOMEM;
    if (n_near) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->near_flag.p, c->near_idx.p, n, c->near_list.p, in_list);
    if (n_far) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->far_flag.p, c->far_idx.p, n, c->far_list.p, in_list);
    if (n_vfar) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->vfar_flag.p, c->vfar_idx.p, n, c->vfar_list.p, in_list);
    if (n_mid) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->mid_flag.p, c->mid_idx.p, n, c->mid_list.p, in_list);
    if (n_sel) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->band_flag.p, c->band_idx.p, n, und_out.p, in_list);
    // the three second passes share the slab and run back to back in stream order; their timings are read after one sync.
    // The slab is sized for the largest of them first: growing it between unsynchronised launches would free memory in use.
    {
        auto slab_bytes = [&](uint32_t n_c, int w) -> size_t {
            const size_t per_block = (size_t)c->max_len * ((w + 2) / 3) * 64 * 4;
            const uint64_t blocks = ((uint64_t)n_c + 127) / 128;
            return (size_t)std::min<uint64_t>(blocks, std::max<uint64_t>(1, c->slab_budget / per_block)) * per_block;
        };
        const size_t need = std::max(std::max(slab_bytes(n_near, MP_BAND_W1), slab_bytes(n_mid, MP_BAND_WM)),
                                     std::max(slab_bytes(n_far, MP_BAND_W2), slab_bytes(n_vfar, MP_BAND_W3)));
        if (c->flags.reserve(need)) return MP_ENOMEM;
    }
    size_t ev_cursor = 6;                              // a multiple of 3 past the events used above
    rc = band_second_pass<MP_BAND_W1, MP_BAND_GB1>(c, d_tasks, c->near_list.p, n_near, cl, cr, d_res, &ev_cursor); if (rc) return rc;
    rc = band_second_pass<MP_BAND_WM, MP_BAND_GBM>(c, d_tasks, c->mid_list.p, n_mid, cl, cr, d_res, &ev_cursor); if (rc) return rc;
    rc = band_second_pass<MP_BAND_W2, MP_BAND_GB>(c, d_tasks, c->far_list.p, n_far, cl, cr, d_res, &ev_cursor); if (rc) return rc;
    rc = band_second_pass<MP_BAND_W3, MP_BAND_GB3>(c, d_tasks, c->vfar_list.p, n_vfar, cl, cr, d_res, &ev_cursor); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s));
    for (size_t q = 2; q < ev_cursor / 3; q++) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, c->ev_pool[3 * q], c->ev_pool[3 * q + 1]);
        (void)hipEventElapsedTime(&b, c->ev_pool[3 * q + 1], c->ev_pool[3 * q + 2]);
        if (ms_fill) *ms_fill += a;
        if (ms_trace) *ms_trace += b;
    }
    *n_und = n_sel;
    return MP_OK;
}

int run_dp_band(mp_ctx *c, const mp_dp_task *d_tasks, uint32_t n, uint32_t max_win, int cl, int cr, mp_dp_result *d_res,
                float *ms_fill, float *ms_trace, bool count)
{
    if (n == 0) return MP_OK;
    hipStream_t s = c->stream;
    const uint32_t ncols = c->max_len;
    // Two passes when the soft-clip limits allow it (rows <= 0 of a centred band reach column 2 GB + 1): scores first, then
    // the traceback bits only around the best cell's diagonal.  MP_DP_BAND=1 keeps the single pass over the whole band.
    static const bool one_pass_env = getenv("MP_DP_BAND") && atoi(getenv("MP_DP_BAND")) == 1;
    const bool two_pass = !one_pass_env && cl >= 2 * MP_BAND_GB + 1 && cr >= 2 * MP_BAND_GB + 1;
    if (c->slab_budget == 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 8ull << 30;
        c->slab_budget = std::min<size_t>(24ull << 30, std::max<size_t>(2ull << 30, (free_b + c->flags.cap) / 4));
    }
    const bool small = c->max_len <= 160;                       // the 160-base instances stage fewer window rows in LDS
    const uint32_t n_blocks = (uint32_t)(((uint64_t)n + 127) / 128);
    if (c->band_fill.reserve(n + 2) || c->band_flag.reserve(n + 2) || c->band_idx.reserve(n + 2)) return MP_ENOMEM;
    const mp_fill_result *none = nullptr;
    uint32_t n_sel = 0;
    const uint32_t *und = nullptr;                               // instances left to the full-table kernels
    if (two_pass) {
        int rc = band_tier<MP_BAND_GB, MP_PRE_MINB, true>(c, d_tasks, nullptr, n, cl, cr, d_res, ms_fill, ms_trace, c->und_list[0], &n_sel);
        if (rc) return rc;
        und = c->und_list[0].p;
        // second tier: alignments that lose up to 25 (scores down to L - 25) over 51 diagonals, for what the first left over
        if (n_sel >= 1024 && cl >= 2 * MP_BAND_GB3 + 1 && cr >= 2 * MP_BAND_GB3 + 1) {
            uint32_t n2 = 0;
            rc = band_tier<MP_BAND_GB3, 7, false>(c, d_tasks, c->und_list[0].p, n_sel, cl, cr, d_res, ms_fill, ms_trace, c->und_list[1], &n2);
            if (rc) return rc;
            n_sel = n2; und = c->und_list[1].p;
        }
    } else {
        const size_t per_block = (size_t)ncols * MP_BAND_NW * 64 * 4;              // 64 instance pairs
        static uint32_t round_cache[2] = {0, 0};
        uint32_t &round_blocks = round_cache[small ? 0 : 1];
        if (round_blocks == 0) {
            int per_cu = 0, n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->ix->device);
            hipError_t e = small ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_band_kernel<MP_BAND_W, MP_BAND_GB, 160, 0>, 64, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dp_band_kernel<MP_BAND_W, MP_BAND_GB, MP_BAND_MAXL, 0>, 64, 0);
            if (e != hipSuccess || per_cu < 1) per_cu = 1;
            round_blocks = (uint32_t)n_cu * (uint32_t)per_cu;
        }
        uint32_t chunk_blocks = (uint32_t)std::min<uint64_t>(n_blocks, std::max<uint64_t>(1, c->slab_budget / per_block));
        if (chunk_blocks > round_blocks && chunk_blocks < n_blocks) chunk_blocks = chunk_blocks / round_blocks * round_blocks;
        if (c->flags.reserve((size_t)chunk_blocks * per_block)) return MP_ENOMEM;
        const size_t n_chunks = (n_blocks + chunk_blocks - 1) / chunk_blocks;
        while (c->ev_pool.size() < 3 * n_chunks) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->ev_pool.push_back(e); }
        uint32_t *slab = (uint32_t *)c->flags.p;
        size_t ci = 0;
        census_add(c, CENSUS_BAND_TRACE, d_tasks, nullptr, n, MP_BAND_W);
        for (uint32_t b0 = 0; b0 < n_blocks; b0 += chunk_blocks, ci++) {
            const uint32_t off = b0 * 128u;
            const uint32_t cnt = (uint32_t)std::min<uint64_t>((uint64_t)chunk_blocks * 128u, n - off);
            const dim3 grid((cnt + 127) / 128), block(64);
            HIPCHK(hipEventRecord(c->ev_pool[3 * ci], s));
            if (small)
                hipLaunchKernelGGL((dp_band_kernel<MP_BAND_W, MP_BAND_GB, 160, 0>), grid, block, 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                                   d_tasks + off, cnt, cl, cr, slab, ncols, none, c->band_fill.p + off);
            else
                hipLaunchKernelGGL((dp_band_kernel<MP_BAND_W, MP_BAND_GB, MP_BAND_MAXL, 0>), grid, block, 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W,
                                   d_tasks + off, cnt, cl, cr, slab, ncols, none, c->band_fill.p + off);
            MP_TRACE(s, "dp_band_kernel");
            HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 1], s));
            hipLaunchKernelGGL((dp_trace_kernel<16, 5, MP_BAND_W, MP_BAND_GB, false>), dim3(grid_for(cnt, 64)), dim3(64), 0, s, d_tasks + off, c->d_lens, cnt, cl, cr,
                               (const uint32_t *)slab, ncols, c->band_fill.p + off, d_res + off, (uint8_t *)nullptr, 0u);
            MP_TRACE(s, "dp_trace_kernel");
            HIPCHK(hipEventRecord(c->ev_pool[3 * ci + 2], s));
        }
        hipLaunchKernelGGL(dp_band_select_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->band_fill.p, n, c->band_flag.p);
        MP_TRACE(s, "dp_band_select_kernel");
        {
            size_t tmp_bytes = 0;
            (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, c->band_flag.p, c->band_idx.p, (int)n, s);
            if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
            (void)hipcub::DeviceScan::InclusiveSum(c->cub_tmp.p, tmp_bytes, c->band_flag.p, c->band_idx.p, (int)n, s);
        }
        HIPCHK(hipMemcpyAsync(&n_sel, c->band_idx.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (size_t q = 0; q < n_chunks; q++) {
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, c->ev_pool[3 * q], c->ev_pool[3 * q + 1]);
            (void)hipEventElapsedTime(&b, c->ev_pool[3 * q + 1], c->ev_pool[3 * q + 2]);
            if (ms_fill) *ms_fill += a;
            if (ms_trace) *ms_trace += b;
        }
        if (c->und_list[0].reserve(n_sel + 130)) return MP_ENOMEM;
        if (n_sel) hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, c->band_flag.p, c->band_idx.p, n, c->und_list[0].p,
                                      (const uint32_t *)nullptr);
        und = c->und_list[0].p;
    }
    c->n_band += n; c->n_band_full += n_sel;
    static const bool verbose = getenv("MP_DP_BAND_STATS") != nullptr;
    if (verbose) fprintf(stderr, "[band] %u instances, %u left to the full kernels\n", n, n_sel);
    if (n_sel == 0) return MP_OK;
    if (c->band_tasks.reserve(n_sel + 2) || c->band_res.reserve(n_sel + 2)) return MP_ENOMEM;
    hipLaunchKernelGGL(dp_gather_tasks_kernel, dim3(grid_for(n_sel, 256)), dim3(256), 0, s, d_tasks, und, n_sel, c->band_tasks.p);
    MP_TRACE(s, "dp_gather_tasks_kernel");
    // what the band leaves over are mostly real alignments with several differences: no score-only pre-pass for them
    int rc = run_dp(c, c->band_tasks.p, n_sel, max_win, cl, cr, c->band_res.p, nullptr, 0, ms_fill, ms_trace, count);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_scatter_list_kernel, dim3(grid_for(n_sel, 256)), dim3(256), 0, s, (const mp_dp_result *)c->band_res.p, und, n_sel, d_res);
    MP_TRACE(s, "dp_scatter_list_kernel");
    return MP_OK;
}

}  // namespace

extern "C" int mp_sgdp_batch(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int clip_left, int clip_right, mp_dp_result *out_results,
                             uint8_t *out_ops, uint32_t ops_stride, float *out_ms_fill, float *out_ms_trace)
{
    if (!c || !tasks || !out_results || !c->d_reads) return MP_EINVAL;
    if (n == 0) return MP_OK;
    HIPCHK(hipSetDevice(c->ix->device));
    uint32_t max_win = 0;
    for (uint32_t t = 0; t < n; t++) {
        if (tasks[t].read_id >= c->n_reads || (tasks[t].strand != 1 && tasks[t].strand != 2)) return MP_EINVAL;
        if (tasks[t].win_len > MP_DP_MAX_WIN || tasks[t].win_start + tasks[t].win_len > c->ix->n) return MP_ERANGE;
        max_win = std::max(max_win, tasks[t].win_len);
    }
    if (c->tasks_tmp.reserve(n) || c->res_tmp.reserve(n)) return MP_ENOMEM;
    if (out_ops && c->ops.reserve((size_t)n * ops_stride + 64)) return MP_ENOMEM;
    HIPCHK(hipMemcpyAsync(c->tasks_tmp.p, tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyHostToDevice, c->stream));
    if (out_ops) HIPCHK(hipMemsetAsync(c->ops.p, 0, (size_t)n * ops_stride, c->stream));
    float f = 0, t = 0;
    int rc = run_dp(c, c->tasks_tmp.p, n, max_win, clip_left, clip_right, c->res_tmp.p, out_ops ? c->ops.p : nullptr, ops_stride, &f, &t, true);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out_results, c->res_tmp.p, (size_t)n * sizeof(mp_dp_result), hipMemcpyDeviceToHost));
    if (out_ops) HIPCHK(hipMemcpy(out_ops, c->ops.p, (size_t)n * ops_stride, hipMemcpyDeviceToHost));
    if (out_ms_fill) *out_ms_fill = f;
    if (out_ms_trace) *out_ms_trace = t;
    return MP_OK;
}

// Test hook: the same instances through the BAND scheme of the deep-DP round (block-match pre-filter with its no-table short cuts, band passes, second
// tier, full table for what is left: run_dp_band) -- scores, start offsets and tie counts must equal mp_sgdp_batch's.  MP_ERANGE where the scheme does
// not apply (reads longer than 320 bases, windows wider than read + 60, clips below the guard band).
extern "C" int mp_test_dp_band(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int clip_left, int clip_right, mp_dp_result *out_results)
{
    if (!c || !tasks || !out_results || !c->d_reads) return MP_EINVAL;
    if (n == 0) return MP_OK;
    HIPCHK(hipSetDevice(c->ix->device));
    uint32_t max_win = 0;
    for (uint32_t t = 0; t < n; t++) {
        if (tasks[t].read_id >= c->n_reads || (tasks[t].strand != 1 && tasks[t].strand != 2)) return MP_EINVAL;
        if (tasks[t].win_len > MP_DP_MAX_WIN || tasks[t].win_start + tasks[t].win_len > c->ix->n) return MP_ERANGE;
        max_win = std::max(max_win, tasks[t].win_len);
    }
    if (!band_applicable(c, max_win, clip_left, clip_right)) return MP_ERANGE;
    if (c->tasks_tmp.reserve(n) || c->res_tmp.reserve(n)) return MP_ENOMEM;
    HIPCHK(hipMemcpyAsync(c->tasks_tmp.p, tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyHostToDevice, c->stream));
    float f = 0, t = 0;
    int rc = run_dp_band(c, c->tasks_tmp.p, n, max_win, clip_left, clip_right, c->res_tmp.p, &f, &t, true);
    if (rc) return rc;
    HIPCHK(hipMemcpy(out_results, c->res_tmp.p, (size_t)n * sizeof(mp_dp_result), hipMemcpyDeviceToHost));
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ deep-DP round
static int start_hits_copy(mp_ctx *c);
static int exclusive_scan_u32(mp_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n);

extern "C" int mp_deep_dp_round(mp_ctx *c, const mp_align_params *ap, uint64_t *out_n_hits, mp_round_stats *out_stats)
{
    if (c && c->hits_copy_started) { (void)hipEventSynchronize(c->ev_hits_copied); c->hits_copy_started = false; }   // the previous round's hits are still travelling
    if (!c || !ap || !c->d_reads || c->n_reads == 0) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    c->deep_done = false; c->rescued = false;
    mp_round_stats st{};
    const hipEvent_t t0 = c->ev_round[0], t1 = c->ev_round[1];
    HIPCHK(hipEventRecord(t0, s));
    int rc = MP_OK;
    if (!c->seeded) { rc = mp_seed_batch(c, &ap->mmp, nullptr); if (rc) return rc; }
    st.seed = c->last_seed;
    const uint32_t n_pairs = c->n_reads / 2;
    // ---- pairing (retry once with the exact size on overflow)
    size_t cand_cap = std::max<size_t>(c->cands.cap, (size_t)n_pairs * 4 + 1024);
    mp_counters h{};
    for (int attempt = 0; attempt < 3; attempt++) {
        if (c->cands.reserve(cand_cap)) return MP_ENOMEM;
        HIPCHK(hipMemsetAsync(&c->ctr.p->n_cand, 0, 2 * sizeof(unsigned long long), s));   // n_cand + overflow
        HIPCHK(hipEventRecord(c->ev[0], s));
        hipLaunchKernelGGL(pair_kernel, dim3(grid_for(n_pairs, 128)), dim3(128), 0, s, c->seedpos.p, c->rs_base.p, c->n_pos.p, c->n_neg.p, c->d_lens, n_pairs,
                           ap->insert_low, ap->insert_high, c->cands.p, (uint32_t)c->cands.cap, c->ctr.p);
        MP_TRACE(s, "pair_kernel");
        HIPCHK(hipEventRecord(c->ev[1], s));
        HIPCHK(hipMemcpyAsync(&h, c->ctr.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h.overflow == 0) break;
        cand_cap = (size_t)h.n_cand + 1024;
        if (attempt == 2) return MP_ERANGE;
    }
    (void)hipEventElapsedTime(&st.ms_pair, c->ev[0], c->ev[1]);
    const uint32_t nc = (uint32_t)h.n_cand;
    st.n_candidates = nc; st.n_dp_left = nc;
    c->n_pair_hits_host = 0;
    // pairs with a hit of this round (the rescue round works on the others): set by collect_pair_hits_kernel as it stores a hit
    if (c->pair_done.reserve((size_t)c->n_reads / 2 + 1)) return MP_ENOMEM;
    HIPCHK(hipMemsetAsync(c->pair_done.p, 0, ((size_t)c->n_reads / 2 + 1) * 4, s));
    if (nc) {
        if (c->tasks_l.reserve(nc) || c->tasks_r.reserve(nc) || c->res_l.reserve(nc) || c->res_r.reserve(nc) || c->pair_hits.reserve(nc)) return MP_ENOMEM;
        const uint32_t margin2 = 60;
        const uint32_t max_win = c->max_len + margin2;
        const bool use_band = band_applicable(c, max_win, ap->clip_left, ap->clip_right);
        hipLaunchKernelGGL(build_left_tasks_kernel, dim3(grid_for(nc, 256)), dim3(256), 0, s, c->cands.p, nc, c->d_lens, c->ix->n, c->tasks_l.p);
        MP_TRACE(s, "build_left_tasks_kernel");
        census_add(c, CENSUS_FULL_EQUIV, c->tasks_l.p, nullptr, nc, 0);
        rc = use_band ? run_dp_band(c, c->tasks_l.p, nc, max_win, ap->clip_left, ap->clip_right, c->res_l.p, &st.ms_dp_fill, &st.ms_dp_trace, ap->want_tie_counts != 0)
                      : run_dp(c, c->tasks_l.p, nc, max_win, ap->clip_left, ap->clip_right, c->res_l.p, nullptr, 0, &st.ms_dp_fill, &st.ms_dp_trace, ap->want_tie_counts != 0);
        if (rc) return rc;
        hipLaunchKernelGGL(build_right_tasks_kernel, dim3(grid_for(nc, 256)), dim3(256), 0, s, c->cands.p, nc, c->d_lens, c->ix->n, ap->insert_high,
                           c->tasks_l.p, c->res_l.p, c->tasks_r.p);
        MP_TRACE(s, "build_right_tasks_kernel");
        census_add(c, CENSUS_FULL_EQUIV, c->tasks_r.p, nullptr, nc, 0);
        rc = use_band ? run_dp_band(c, c->tasks_r.p, nc, max_win, ap->clip_left, ap->clip_right, c->res_r.p, &st.ms_dp_fill, &st.ms_dp_trace, ap->want_tie_counts != 0)
                      : run_dp(c, c->tasks_r.p, nc, max_win, ap->clip_left, ap->clip_right, c->res_r.p, nullptr, 0, &st.ms_dp_fill, &st.ms_dp_trace, ap->want_tie_counts != 0);
        if (rc) return rc;
        // the hits leave grouped by pair, pairs ascending: count per pair, scan, store at the group's offset
        const uint32_t np1 = n_pairs + 1;
        if (c->pair_cnt.reserve(np1) || c->pair_off.reserve(np1) || c->hit_flag.reserve(nc + 2) || c->hit_idx.reserve(nc + 2) || c->cand_first.reserve(np1)) return MP_ENOMEM;
        HIPCHK(hipMemsetAsync(c->pair_cnt.p, 0, (size_t)np1 * 4, s));
        hipLaunchKernelGGL(cand_first_kernel, dim3(grid_for(nc, 256)), dim3(256), 0, s, c->cands.p, nc, c->cand_first.p);
        hipLaunchKernelGGL(collect_pair_hits_kernel<true>, dim3(grid_for(nc, 256)), dim3(256), 0, s, c->cands.p, nc, c->d_lens, c->tasks_l.p, c->res_l.p, c->tasks_r.p,
                           c->res_r.p, c->pair_hits.p, c->pair_cnt.p, (const uint32_t *)c->pair_off.p, (uint32_t)c->pair_hits.cap, c->pair_done.p,
                           c->hit_flag.p, (const uint32_t *)c->hit_idx.p, (const uint32_t *)c->cand_first.p);
        rc = exclusive_scan_u32(c, c->pair_cnt.p, c->pair_off.p, np1);
        if (rc) return rc;
        rc = exclusive_scan_u32(c, c->hit_flag.p, c->hit_idx.p, nc);      // a hit's place inside its pair's group: candidate order, not arrival order
        if (rc) return rc;
        hipLaunchKernelGGL(collect_pair_hits_kernel<false>, dim3(grid_for(nc, 256)), dim3(256), 0, s, c->cands.p, nc, c->d_lens, c->tasks_l.p, c->res_l.p, c->tasks_r.p,
                           c->res_r.p, c->pair_hits.p, c->pair_cnt.p, (const uint32_t *)c->pair_off.p, (uint32_t)c->pair_hits.cap, c->pair_done.p,
                           c->hit_flag.p, (const uint32_t *)c->hit_idx.p, (const uint32_t *)c->cand_first.p);
        MP_TRACE(s, "collect_pair_hits_kernel");
        uint32_t nh = 0;
        HIPCHK(hipMemcpyAsync(&nh, c->pair_off.p + n_pairs, sizeof nh, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        c->n_pair_hits_host = nh;
    }
    st.n_pair_hits = c->n_pair_hits_host;
    HIPCHK(hipEventRecord(t1, s));
    HIPCHK(hipEventSynchronize(t1));
    (void)hipEventElapsedTime(&st.ms_total, t0, t1);
    c->deep_done = true;
    if (out_n_hits) *out_n_hits = c->n_pair_hits_host;
    if (out_stats) *out_stats = st;
    // the hits start their way to the host now (mp_get_pair_hits_view waits for them): a rescue round called in between
    // runs while they travel
    return start_hits_copy(c);
}

extern "C" int mp_get_pair_hits(mp_ctx *c, mp_pair_hit *out_hits, uint64_t cap)
{
    if (!c || !out_hits) return MP_EINVAL;
    if (cap < c->n_pair_hits_host) return MP_ERANGE;
    HIPCHK(hipSetDevice(c->ix->device));
    if (c->n_pair_hits_host) HIPCHK(hipMemcpy(out_hits, c->pair_hits.p, (size_t)c->n_pair_hits_host * sizeof(mp_pair_hit), hipMemcpyDeviceToHost));
    return MP_OK;
}

// pinned staging buffer for n hits
static int reserve_pinned_hits(mp_ctx *c, size_t n)
{
    if (n <= c->h_pair_hits_cap) return MP_OK;
    if (c->h_pair_hits) (void)hipHostFree(c->h_pair_hits);
    c->h_pair_hits = nullptr; c->h_pair_hits_cap = 0;
    const size_t want = n + n / 4 + 1024;
    if (hipHostMalloc((void **)&c->h_pair_hits, want * sizeof(mp_pair_hit), hipHostMallocDefault) != hipSuccess) return MP_ENOMEM;
    c->h_pair_hits_cap = want;
    return MP_OK;
}

// Starts the device-to-host copy of the round's hits on the copy stream, behind everything queued on the compute stream so
// far: whatever the caller runs next (the rescue round) overlaps with the transfer.
static int start_hits_copy(mp_ctx *c)
{
    c->hits_copy_started = false;
    const size_t n = c->n_pair_hits_host;
    if (n == 0 || !c->copy_hits) return MP_OK;
    int rc = reserve_pinned_hits(c, n);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_hits_ready, c->stream));
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_hits_ready, 0));
    HIPCHK(hipMemcpyAsync(c->h_pair_hits, c->pair_hits.p, n * sizeof(mp_pair_hit), hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(hipEventRecord(c->ev_hits_copied, c->copy_stream));
    c->hits_copy_started = true;
    return MP_OK;
}

extern "C" int mp_get_pair_hits_view(mp_ctx *c, const mp_pair_hit **out_hits, uint64_t *out_n)
{
    if (!c || !out_hits || !out_n) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    const size_t n = c->n_pair_hits_host;
    if (n) {
        if (c->hits_copy_started) HIPCHK(hipEventSynchronize(c->ev_hits_copied));      // started by mp_deep_dp_round
        else {
            int rc = reserve_pinned_hits(c, n);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(c->h_pair_hits, c->pair_hits.p, n * sizeof(mp_pair_hit), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
    }
    *out_hits = c->h_pair_hits; *out_n = n;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ rescue passes
static int exclusive_scan_u32(mp_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n)
{
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, in, out, (int)n, c->stream);
    if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
    (void)hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tmp_bytes, in, out, (int)n, c->stream);
    return MP_OK;
}

// pinned staging buffers of the rescue hit lists
template <typename T>
static int reserve_pinned(T *&p, size_t &cap, size_t n)
{
    if (n <= cap) return MP_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = n + n / 4 + 1024;
    if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) return MP_ENOMEM;
    cap = want;
    return MP_OK;
}

extern "C" int mp_rescue_round(mp_ctx *c, const mp_align_params *ap, int skip_default_dp, mp_rescue_stats *out_stats)
{
    // the pairs to rescue are the ones without a hit of the deep-DP round of THIS batch
    if (!c || !ap || !c->seeded || !c->deep_done || c->n_reads == 0) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    if (c->rescue_copy_started) { HIPCHK(hipEventSynchronize(c->ev_rescue_copied)); c->rescue_copy_started = false; }
    hipStream_t s = c->stream;
    const uint32_t n_reads = c->n_reads;
    mp_rescue_stats st{};
    const hipEvent_t t0 = c->ev_round[0], t1 = c->ev_round[1], t2 = c->ev_round[2];
    HIPCHK(hipEventRecord(t0, s));
    if (c->failed_flag.reserve(n_reads + 2) || c->failed_off.reserve(n_reads + 2) || !c->pair_done.p) return MP_ENOMEM;      // (pair_done: left by the deep-DP round of this batch)
    hipLaunchKernelGGL(failed_flag_kernel, dim3(grid_for(n_reads + 1, 256)), dim3(256), 0, s, c->pair_done.p, n_reads, c->failed_flag.p);
    MP_TRACE(s, "failed_flag_kernel");
    int rc = exclusive_scan_u32(c, c->failed_flag.p, c->failed_off.p, n_reads + 1);
    if (rc) return rc;
    uint32_t nf = 0;
    HIPCHK(hipMemcpyAsync(&nf, c->failed_off.p + n_reads, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n_failed_reads = nf; c->n_stasks = 0; c->rescued = true;
    c->n_single_hits_host = 0; c->n_rescue_hits_host = 0;
    st.n_failed_pairs = nf / 2;
    uint32_t nst = 0, n_rt = 0;
    if (nf) {
        if (c->failed_reads.reserve(nf + 1) || c->n_scand.reserve(nf + 2) || c->s_off.reserve(nf + 2) || c->scand.reserve(c->n_hits_total + 64)) return MP_ENOMEM;
        hipLaunchKernelGGL(list_failed_kernel, dim3(grid_for(n_reads, 256)), dim3(256), 0, s, c->failed_flag.p, c->failed_off.p, n_reads, c->failed_reads.p);
        MP_TRACE(s, "list_failed_kernel");
        hipLaunchKernelGGL(single_cand_kernel, dim3(grid_for(nf + 1, 64)), dim3(64), 0, s, c->failed_reads.p, nf, c->seedpos.p, c->rs_base.p, c->n_pos.p,
                           c->n_neg.p, c->scand.p, c->n_scand.p);
        MP_TRACE(s, "single_cand_kernel");
        rc = exclusive_scan_u32(c, c->n_scand.p, c->s_off.p, nf + 1);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(&nst, c->s_off.p + nf, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        c->n_stasks = nst;
        st.n_single_tasks = nst;
        if (c->stasks.reserve(nst + 2) || c->sres.reserve(nst + 2) || c->rtasks.reserve(4 * (size_t)nf + 2) || c->rres.reserve(4 * (size_t)nf + 2) ||
            c->rsrc.reserve(4 * (size_t)nf + 2)) return MP_ENOMEM;
        hipLaunchKernelGGL(single_task_kernel, dim3(grid_for(nf, 128)), dim3(128), 0, s, c->failed_reads.p, nf, c->scand.p, c->rs_base.p, c->n_scand.p,
                           c->s_off.p, c->d_lens, c->ix->n, c->stasks.p);
        MP_TRACE(s, "single_task_kernel");
        // ---- the -top rule across chunks (mp_ctx_set_score_floor, kernels_floor.hip): pairs whose rescue work here cannot put anything into a header lose
        // their instances -- a cheap bound per instance, a decision per pair, then the task list once more without them
        static const bool floor_env = !getenv("MP_TOP_FLOOR") || atoi(getenv("MP_TOP_FLOOR")) != 0;
        if (floor_env && c->floor_top > 0 && c->red_chunks > 0 && c->best_known.p && nst >= 64 && c->max_len <= MP_FLOOR_MAXL) {
            // MP_TOP_FLOOR_MATE=0: without the bound on the rescued mate (floor_mate_kernel): the rule as first built, a test / measurement switch
            static const bool mate_env = !getenv("MP_TOP_FLOOR_MATE") || atoi(getenv("MP_TOP_FLOOR_MATE")) != 0;
            const bool mate_rule = mate_env && !skip_default_dp && ap->insert_high > 0 && ap->insert_high <= 4096;
            if (c->floor_ub.reserve(nst + 2) || c->floor_counts.reserve(4) || (mate_rule && c->floor_hope.reserve(nf / 2 + 2))) return MP_ENOMEM;
            HIPCHK(hipMemsetAsync(c->floor_counts.p, 0, 4 * sizeof(unsigned long long), s));
            hipLaunchKernelGGL(floor_bound_kernel, dim3(grid_for(nst, 64)), dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, (const mp_dp_task *)c->stasks.p, nst,
                               (const uint32_t *)c->best_known.p, c->floor_top, c->floor_ub.p);
            hipLaunchKernelGGL(floor_decide_kernel, dim3(grid_for(nf / 2 + 1, 256)), dim3(256), 0, s, (const uint32_t *)c->failed_reads.p, nf, (const uint32_t *)c->s_off.p,
                               (const uint16_t *)c->floor_ub.p, c->d_lens, (const uint32_t *)c->best_known.p, c->floor_top, c->n_scand.p, c->floor_counts.p,
                               mate_rule ? c->floor_hope.p : (uint8_t *)nullptr);
            if (mate_rule && nf >= 2)
                hipLaunchKernelGGL(floor_mate_kernel, dim3(nf / 2), dim3(64), 0, s, c->ix->d.text, (uint64_t)c->ix->n, c->d_reads, c->d_lens, c->W, (const mp_dp_task *)c->stasks.p,
                                   (const uint32_t *)c->failed_reads.p, nf, (const uint32_t *)c->s_off.p, (const uint16_t *)c->floor_ub.p, (const uint32_t *)c->best_known.p,
                                   c->floor_top, (const uint8_t *)c->floor_hope.p, ap->insert_high, c->n_scand.p, c->floor_counts.p, (uint16_t *)nullptr);
            MP_TRACE(s, "floor kernels");
            HIPCHK(hipGetLastError());
            unsigned long long fc[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(fc, c->floor_counts.p, sizeof fc, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            st.n_floor_pairs = fc[0]; st.n_floor_tasks = fc[1]; st.n_floor_mate_tried = fc[2]; st.n_floor_mate_pairs = fc[3];
            if (fc[0]) {
                rc = exclusive_scan_u32(c, c->n_scand.p, c->s_off.p, nf + 1);
                if (rc) return rc;
                HIPCHK(hipMemcpyAsync(&nst, c->s_off.p + nf, 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                c->n_stasks = nst;
                st.n_single_tasks = nst;
                hipLaunchKernelGGL(single_task_kernel, dim3(grid_for(nf, 128)), dim3(128), 0, s, c->failed_reads.p, nf, c->scand.p, c->rs_base.p, c->n_scand.p,
                                   c->s_off.p, c->d_lens, c->ix->n, c->stasks.p);
                MP_TRACE(s, "single_task_kernel (after the floor)");
            }
        }
        HIPCHK(hipEventRecord(t1, s));
        census_add(c, CENSUS_FULL_EQUIV, c->stasks.p, nullptr, nst, 0);
        const uint32_t max_win = c->max_len + 60;
        rc = run_dp_filtered(c, c->stasks.p, nst, max_win, ap->clip_left, ap->clip_right, c->sres.p, &st.ms_dp_fill, &st.ms_dp_trace, ap->want_tie_counts != 0);
        if (rc) return rc;
        if (!skip_default_dp && nst) {
            hipLaunchKernelGGL(mate_rescue_task_kernel, dim3(grid_for(nf, 128)), dim3(128), 0, s, c->failed_reads.p, nf, c->s_off.p, c->stasks.p, c->sres.p,
                               c->d_lens, c->ix->n, ap->insert_low, ap->insert_high, c->rtasks.p, c->rsrc.p);
            MP_TRACE(s, "mate_rescue_task_kernel");
            const uint32_t rwin = (uint32_t)(ap->insert_high - ap->insert_low) + c->max_len + 1;   // HalfEnd maxDNALength (DV-DPfunctions.cpp:1340)
            if (rwin > MP_DP_MAX_WIN) return MP_ERANGE;
            census_add(c, CENSUS_FULL_EQUIV, c->rtasks.p, nullptr, 4 * nf, 0);
            // four slots per read, most of them empty (a read without a passing single-end hit has none to rescue its mate from): the DP
            // runs over the filled slots only, gathered in slot order; the empty ones keep an all-zero result
            {
                const uint32_t n_slots = 4 * nf;
                if (c->sel_flag.reserve(n_slots + 2) || c->sel_idx.reserve(n_slots + 2)) return MP_ENOMEM;
                HIPCHK(hipMemsetAsync(c->rres.p, 0, (size_t)n_slots * sizeof(mp_dp_result), s));
                hipLaunchKernelGGL(dp_nonempty_flag_kernel, dim3(grid_for(n_slots, 256)), dim3(256), 0, s, (const mp_dp_task *)c->rtasks.p, n_slots, c->sel_flag.p);
                MP_TRACE(s, "dp_nonempty_flag_kernel");
                size_t tmp_bytes = 0;
                (void)hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, c->sel_flag.p, c->sel_idx.p, (int)n_slots, s);
                if (c->cub_tmp.reserve(tmp_bytes + 256)) return MP_ENOMEM;
                (void)hipcub::DeviceScan::InclusiveSum(c->cub_tmp.p, tmp_bytes, c->sel_flag.p, c->sel_idx.p, (int)n_slots, s);
                uint32_t n_filled = 0;
                HIPCHK(hipMemcpyAsync(&n_filled, c->sel_idx.p + (n_slots - 1), 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (n_filled) {
                    if (c->rlist.reserve(n_filled + 2) || c->rtasks_c.reserve(n_filled + 2) || c->rres_c.reserve(n_filled + 2)) return MP_ENOMEM;
                    hipLaunchKernelGGL(dp_band_list_kernel, dim3(grid_for(n_slots, 256)), dim3(256), 0, s, c->sel_flag.p, c->sel_idx.p, n_slots, c->rlist.p, (const uint32_t *)nullptr);
                    hipLaunchKernelGGL(dp_gather_tasks_kernel, dim3(grid_for(n_filled, 256)), dim3(256), 0, s, (const mp_dp_task *)c->rtasks.p, (const uint32_t *)c->rlist.p, n_filled, c->rtasks_c.p);
                    MP_TRACE(s, "dp_gather_tasks_kernel");
                    rc = run_dp_filtered(c, c->rtasks_c.p, n_filled, rwin, ap->clip_left, ap->clip_right, c->rres_c.p, &st.ms_dp_fill, &st.ms_dp_trace,
                                         ap->want_tie_counts != 0, true);
                    if (rc) return rc;
                    hipLaunchKernelGGL(dp_scatter_list_kernel, dim3(grid_for(n_filled, 256)), dim3(256), 0, s, (const mp_dp_result *)c->rres_c.p, (const uint32_t *)c->rlist.p, n_filled, c->rres.p);
                    MP_TRACE(s, "dp_scatter_list_kernel");
                }
            }
            n_rt = 4 * nf;
            st.n_rescue_tasks = n_rt;
        }
        // ---- the passing instances, compacted in task order
        uint32_t n_sh = 0, n_rh = 0;
        if (nst) {
            if (c->sh_flag.reserve(nst + 2) || c->sh_slot.reserve(nst + 2)) return MP_ENOMEM;
            hipLaunchKernelGGL(single_hit_flag_kernel, dim3(grid_for(nst + 1, 256)), dim3(256), 0, s, c->stasks.p, c->sres.p, c->d_lens, nst, c->sh_flag.p);
            MP_TRACE(s, "single_hit_flag_kernel");
            rc = exclusive_scan_u32(c, c->sh_flag.p, c->sh_slot.p, nst + 1);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(&n_sh, c->sh_slot.p + nst, 4, hipMemcpyDeviceToHost, s));
        }
        if (n_rt) {
            if (c->rh_flag.reserve(n_rt + 2) || c->rh_slot.reserve(n_rt + 2)) return MP_ENOMEM;
            hipLaunchKernelGGL(rescue_hit_flag_kernel, dim3(grid_for(n_rt + 1, 256)), dim3(256), 0, s, c->rtasks.p, c->rres.p, c->rsrc.p, c->d_lens, n_rt, c->rh_flag.p);
            MP_TRACE(s, "rescue_hit_flag_kernel");
            rc = exclusive_scan_u32(c, c->rh_flag.p, c->rh_slot.p, n_rt + 1);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(&n_rh, c->rh_slot.p + n_rt, 4, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        if (c->d_single_hits.reserve(n_sh + 1) || c->d_rescue_hits.reserve(n_rh + 1)) return MP_ENOMEM;
        if (c->copy_hits && (reserve_pinned(c->h_single_hits, c->h_single_cap, n_sh) || reserve_pinned(c->h_rescue_hits, c->h_rescue_cap, n_rh))) return MP_ENOMEM;
        if (n_sh) hipLaunchKernelGGL(single_hit_emit_kernel, dim3(grid_for(nst, 256)), dim3(256), 0, s, c->stasks.p, c->sres.p, c->sh_flag.p, c->sh_slot.p, nst, c->d_single_hits.p);
        if (n_rh) hipLaunchKernelGGL(rescue_hit_emit_kernel, dim3(grid_for(n_rt, 256)), dim3(256), 0, s, c->stasks.p, c->sres.p, c->rtasks.p, c->rres.p, c->rsrc.p,
                                     c->rh_flag.p, c->rh_slot.p, n_rt, c->d_rescue_hits.p);
        c->n_single_hits_host = n_sh; c->n_rescue_hits_host = n_rh;
        st.n_single_hits = n_sh; st.n_rescue_hits = n_rh;
        if ((n_sh || n_rh) && c->copy_hits) {
            HIPCHK(hipEventRecord(c->ev_rescue_ready, s));
            HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_rescue_ready, 0));
            if (n_sh) HIPCHK(hipMemcpyAsync(c->h_single_hits, c->d_single_hits.p, (size_t)n_sh * sizeof(mp_single_hit), hipMemcpyDeviceToHost, c->copy_stream));
            if (n_rh) HIPCHK(hipMemcpyAsync(c->h_rescue_hits, c->d_rescue_hits.p, (size_t)n_rh * sizeof(mp_rescue_hit), hipMemcpyDeviceToHost, c->copy_stream));
            HIPCHK(hipEventRecord(c->ev_rescue_copied, c->copy_stream));
            c->rescue_copy_started = true;
        }
    }
    HIPCHK(hipEventRecord(t2, s));
    HIPCHK(hipEventSynchronize(t2));
    (void)hipEventElapsedTime(&st.ms_total, t0, t2);
    if (nf) (void)hipEventElapsedTime(&st.ms_tasks, t0, t1);
    if (out_stats) *out_stats = st;
    return MP_OK;
}

static int wait_rescue_copy(mp_ctx *c)
{
    if (c->rescue_copy_started) { HIPCHK(hipEventSynchronize(c->ev_rescue_copied)); c->rescue_copy_started = false; }
    return MP_OK;
}

extern "C" int mp_get_single_hits_view(mp_ctx *c, const mp_single_hit **out_hits, uint64_t *out_n)
{
    if (!c || !c->rescued || !out_hits || !out_n || !c->copy_hits) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = wait_rescue_copy(c);
    if (rc) return rc;
    *out_hits = c->h_single_hits; *out_n = c->n_single_hits_host;
    return MP_OK;
}

extern "C" int mp_get_rescue_hits_view(mp_ctx *c, const mp_rescue_hit **out_hits, uint64_t *out_n)
{
    if (!c || !c->rescued || !out_hits || !out_n || !c->copy_hits) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = wait_rescue_copy(c);
    if (rc) return rc;
    *out_hits = c->h_rescue_hits; *out_n = c->n_rescue_hits_host;
    return MP_OK;
}

extern "C" int mp_get_single_hits(mp_ctx *c, mp_single_hit *out, uint64_t cap, uint64_t *out_n)
{
    const mp_single_hit *v = nullptr;
    uint64_t n = 0;
    if (!out_n) return MP_EINVAL;
    int rc = mp_get_single_hits_view(c, &v, &n);
    if (rc) return rc;
    if (out) {
        if (n > cap) return MP_ERANGE;
        if (n) memcpy(out, v, (size_t)n * sizeof(mp_single_hit));
    }
    *out_n = n;
    return MP_OK;
}

extern "C" int mp_get_rescue_hits(mp_ctx *c, mp_rescue_hit *out, uint64_t cap, uint64_t *out_n)
{
    const mp_rescue_hit *v = nullptr;
    uint64_t n = 0;
    if (!out_n) return MP_EINVAL;
    int rc = mp_get_rescue_hits_view(c, &v, &n);
    if (rc) return rc;
    if (out) {
        if (n > cap) return MP_ERANGE;
        if (n) memcpy(out, v, (size_t)n * sizeof(mp_rescue_hit));
    }
    *out_n = n;
    return MP_OK;
}

// Test aid: the order single_cand_kernel sorts n_arrays lists of seed lengths into (array t = seed_len[off[t] .. off[t+1]));
// out_perm[off[t] + k] = index inside array t of the seed that ends up at rank k.
extern "C" int mp_test_sort_longer_seed(int device, const uint32_t *seed_len, const uint32_t *off, uint32_t n_arrays, uint32_t *out_perm)
{
    if (!seed_len || !off || !out_perm) return MP_EINVAL;
    if (n_arrays == 0) return MP_OK;
    int rc = mp_device_check(device, nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    const uint32_t total = off[n_arrays];
    for (uint32_t t = 0; t < n_arrays; t++) if (off[t + 1] - off[t] >= 4096) return MP_ERANGE;      // a read has < 2 * MP_SEED_BIG candidates
    std::vector<mp_single_cand> h(total + 1);
    for (uint32_t t = 0; t < n_arrays; t++)
        for (uint32_t i = off[t]; i < off[t + 1]; i++) { h[i].pos = i - off[t]; h[i].strand = 1; h[i].seed_len = seed_len[i]; }
    mp_single_cand *d = nullptr; uint32_t *d_off = nullptr;
    if (hipMalloc((void **)&d, (size_t)(total + 1) * sizeof(mp_single_cand)) != hipSuccess || hipMalloc((void **)&d_off, (size_t)(n_arrays + 1) * 4) != hipSuccess) {
        if (d) (void)hipFree(d);
        return MP_ENOMEM;
    }
    hipError_t e = hipMemcpy(d, h.data(), (size_t)total * sizeof(mp_single_cand), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, off, (size_t)(n_arrays + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sort_longer_seed_test_kernel, dim3(grid_for(n_arrays, 64)), dim3(64), 0, 0, d, d_off, n_arrays);
        e = hipMemcpy(h.data(), d, (size_t)total * sizeof(mp_single_cand), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d); (void)hipFree(d_off);
    if (e != hipSuccess) return MP_ENODEV;
    for (uint32_t i = 0; i < total; i++) out_perm[i] = (uint32_t)h[i].pos;
    return MP_OK;
}


// free / total bytes of a GPU's memory right now (the executable sizes its batches by it when the -top rule needs room for its stash)
extern "C" int mp_device_memory(int device, uint64_t *out_free, uint64_t *out_total)
{
    if (!out_free || !out_total) return MP_EINVAL;
    int rc = mp_device_check(device, nullptr, 0);
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    *out_free = f; *out_total = t;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ the -top rule across resident chunks
// 0: the rescue rounds do all their work.  1 .. 100 (the executable's -top value): mp_rescue_round leaves out the rescue work of pairs that cannot put an
// entry of top% of their best KNOWN score into a header (kernels_floor.hip) -- to be switched off for the last chunk of a database (its classes are the
// sections of the output) and whenever the hit lists themselves are wanted.  Needs a reduction in progress (mp_reduce_begin): that is where the best scores live.
extern "C" int mp_ctx_set_score_floor(mp_ctx *c, int top_percent)
{
    if (!c || top_percent < 0 || top_percent > 100) return MP_EINVAL;
    c->floor_top = top_percent;
    return MP_OK;
}

// test hook: the bound of kernels_floor.hip for caller-given instances of the reads now set (every instance, whatever is known about its read);
// out_ub[t] >= the exact DP score of instance t is what the rule's soundness rests on.  0xFFFF: shape not covered.
extern "C" int mp_test_floor_bound(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, uint16_t *out_ub)
{
    if (!c || !tasks || !out_ub || !c->d_reads) return MP_EINVAL;
    if (n == 0) return MP_OK;
    for (uint32_t t = 0; t < n; t++) {
        if (tasks[t].read_id >= c->n_reads || (tasks[t].strand != 1 && tasks[t].strand != 2)) return MP_EINVAL;
        if (tasks[t].win_len > MP_DP_MAX_WIN || tasks[t].win_start + tasks[t].win_len > c->ix->n) return MP_ERANGE;
    }
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    DevBuf<uint32_t> known;
    if (c->tasks_tmp.reserve(n) || c->floor_ub.reserve(n + 2) || known.reserve((size_t)c->n_reads + 1)) { known.release(); return MP_ENOMEM; }
    hipError_t e = hipMemcpyAsync(c->tasks_tmp.p, tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(known.p, 0x01, ((size_t)c->n_reads + 1) * 4, s);                 // 0x01010101: "a very good score is known for every read"
    if (e == hipSuccess) {
        hipLaunchKernelGGL(floor_bound_kernel, dim3(grid_for(n, 64)), dim3(64), 0, s, c->ix->d.text, c->d_reads, c->d_lens, c->W, (const mp_dp_task *)c->tasks_tmp.p, n,
                           (const uint32_t *)known.p, 100, c->floor_ub.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_ub, c->floor_ub.p, (size_t)n * 2, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    known.release();
    return e == hipSuccess ? MP_OK : MP_ENODEV;
}

// test hook: floor_mate_kernel's bound on the rescued mate for caller-given single-end instances -- tasks grouped by read, reads in ascending order, the
// instances of read 2k (anchor; its mate is read 2k + 1) and of read 2k + 1 (mate: read 2k).  out_ub[t] >= the exact DP score of the mate in ANY window
// mate_rescue_task_kernel can make from a hit inside instance t's window is what the rule's second half rests on.
extern "C" int mp_test_floor_mate_bound(mp_ctx *c, const mp_dp_task *tasks, uint32_t n, int insert_high, uint16_t *out_ub)
{
    if (!c || !tasks || !out_ub || !c->d_reads || insert_high <= 0 || insert_high > 4096 || (c->n_reads & 1)) return MP_EINVAL;
    if (n == 0) return MP_OK;
    if (c->max_len > MP_FLOOR_MAXL) return MP_ERANGE;
    std::vector<uint32_t> off((size_t)c->n_reads + 2, 0), fr(c->n_reads);
    for (uint32_t t = 0; t < n; t++) {
        if (tasks[t].read_id >= c->n_reads || (tasks[t].strand != 1 && tasks[t].strand != 2) || (t && tasks[t].read_id < tasks[t - 1].read_id)) return MP_EINVAL;
        if (tasks[t].win_len > MP_DP_MAX_WIN || tasks[t].win_start + tasks[t].win_len > c->ix->n) return MP_ERANGE;
        off[tasks[t].read_id + 1]++;
    }
    for (uint32_t r = 0; r < c->n_reads; r++) { off[r + 1] += off[r]; fr[r] = r; }
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    DevBuf<uint32_t> known, d_off, d_fr;
    DevBuf<uint16_t> d_ub, d_out;
    auto done = [&](int rc) { known.release(); d_off.release(); d_fr.release(); d_ub.release(); d_out.release(); return rc; };
    if (c->tasks_tmp.reserve(n) || known.reserve((size_t)c->n_reads + 1) || d_off.reserve(off.size()) || d_fr.reserve(fr.size() + 1) || d_ub.reserve(n + 2) || d_out.reserve(n + 2))
        return done(MP_ENOMEM);
    hipError_t e = hipMemcpyAsync(c->tasks_tmp.p, tasks, (size_t)n * sizeof(mp_dp_task), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_fr.p, fr.data(), fr.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(known.p, 0x01, ((size_t)c->n_reads + 1) * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_ub.p, 0, (size_t)n * 2, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_out.p, 0xFF, (size_t)n * 2, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(floor_mate_kernel, dim3(c->n_reads / 2), dim3(64), 0, s, c->ix->d.text, (uint64_t)c->ix->n, c->d_reads, c->d_lens, c->W, (const mp_dp_task *)c->tasks_tmp.p,
                           (const uint32_t *)d_fr.p, c->n_reads, (const uint32_t *)d_off.p, (const uint16_t *)d_ub.p, (const uint32_t *)known.p, 100, (const uint8_t *)nullptr,
                           insert_high, (uint32_t *)nullptr, (unsigned long long *)nullptr, d_out.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_ub, d_out.p, (size_t)n * 2, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return done(e == hipSuccess ? MP_OK : MP_ENODEV);
}

// Between the two passes over the resident chunks: after mp_deep_dp_round (and mp_reduce_chunk_part(.., 1)) of the bound chunk, keeps what its rescue round
// needs -- the seed positions of every read and which pairs the round aligned -- in the context's stash (one arena, exact sizes); mp_chunk_unstash(slot),
// after the context has been bound to that chunk again, puts it back, and mp_rescue_round carries on as if nothing had happened in between.
// A new batch (mp_set_reads) empties the stash.
extern "C" int mp_chunk_stash(mp_ctx *c, uint32_t slot)
{
    if (!c || !c->seeded || !c->deep_done || slot >= MP_RED_MAX_CHUNKS) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    if (c->stash.size() <= slot) c->stash.resize(slot + 1);
    const size_t n_items = 2 * (size_t)c->n_reads, n_pairs = c->n_reads / 2;
    // seed positions as 8-byte words (kernels_floor.hip, stash_pack_kernel) when the batch's pair numbers fit 21 bits; MP_STASH_PLAIN=1: as they stand (test switch)
    static const bool plain_env = getenv("MP_STASH_PLAIN") && atoi(getenv("MP_STASH_PLAIN")) != 0;
    const bool packed = !plain_env && c->n_reads <= (1u << 22);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_rs = (n_items + 1) * 4, b_np = (size_t)c->n_reads * 4, b_pd = (n_pairs + 1) * 4;
    const size_t b_seed = (size_t)c->n_hits_total * (packed ? sizeof(unsigned long long) : sizeof(mp_seedpos));
    const size_t need = up(b_seed) + up(b_rs) + 2 * up(b_np) + up(b_pd);
    if (c->stash_used + need > c->stash_cap) {
        // grow: what this chunk needed times the chunks still to come (the reduction knows how many chunks the batch goes through), a tenth of slack
        const size_t n_more = c->red_chunks > slot ? c->red_chunks - slot : 1;
        size_t want = c->stash_used + need * n_more + need / 10 * n_more + (1u << 20);
        uint8_t *q = nullptr;
        if (hipMalloc((void **)&q, want) != hipSuccess) return MP_ENOMEM;
        if (c->stash_used) { if (hipMemcpyAsync(q, c->stash_arena, c->stash_used, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(q); return MP_ENODEV; } }
        if (c->stash_arena) (void)hipFree(c->stash_arena);
        c->stash_arena = q; c->stash_cap = want;
    }
    uint8_t *p = c->stash_arena + c->stash_used;
    if (b_seed && packed) {
        hipLaunchKernelGGL(stash_pack_kernel, dim3(grid_for(c->n_hits_total, 256)), dim3(256), 0, s, (const mp_seedpos *)c->seedpos.p, (uint64_t)c->n_hits_total, (unsigned long long *)p);
        MP_TRACE(s, "stash_pack_kernel");
    } else if (b_seed) HIPCHK(hipMemcpyAsync(p, c->seedpos.p, b_seed, hipMemcpyDeviceToDevice, s));
    mp_ctx::Stash &st = c->stash[slot];
    st.off = c->stash_used; st.bytes = need; st.n_hits = c->n_hits_total; st.n_reads = c->n_reads; st.ix = c->ix; st.seed = c->last_seed; st.used = true; st.packed = packed;
    p += up(b_seed);
    HIPCHK(hipMemcpyAsync(p, c->rs_base.p, b_rs, hipMemcpyDeviceToDevice, s)); p += up(b_rs);
    HIPCHK(hipMemcpyAsync(p, c->n_pos.p, b_np, hipMemcpyDeviceToDevice, s)); p += up(b_np);
    HIPCHK(hipMemcpyAsync(p, c->n_neg.p, b_np, hipMemcpyDeviceToDevice, s)); p += up(b_np);
    HIPCHK(hipMemcpyAsync(p, c->pair_done.p, b_pd, hipMemcpyDeviceToDevice, s));
    c->stash_used += need;
    HIPCHK(hipStreamSynchronize(s));
    return MP_OK;
}

extern "C" int mp_chunk_unstash(mp_ctx *c, uint32_t slot)
{
    if (!c || slot >= c->stash.size() || !c->stash[slot].used) return MP_EINVAL;
    const mp_ctx::Stash &st = c->stash[slot];
    if (st.ix != c->ix || st.n_reads != c->n_reads) return MP_EINVAL;              // bind the chunk first (mp_ctx_set_index); same batch
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    const size_t n_items = 2 * (size_t)c->n_reads, n_pairs = c->n_reads / 2;
    const size_t b_seed = (size_t)st.n_hits * (st.packed ? sizeof(unsigned long long) : sizeof(mp_seedpos)), b_rs = (n_items + 1) * 4, b_np = (size_t)c->n_reads * 4, b_pd = (n_pairs + 1) * 4;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if (c->seedpos.reserve((size_t)st.n_hits + 64) || c->rs_base.reserve(n_items + 2) || c->n_pos.reserve(c->n_reads + 1) || c->n_neg.reserve(c->n_reads + 1) ||
        c->pair_done.reserve(n_pairs + 1)) return MP_ENOMEM;
    const uint8_t *p = c->stash_arena + st.off;
    if (b_seed && st.packed) {
        hipLaunchKernelGGL(stash_unpack_kernel, dim3(grid_for(st.n_hits, 256)), dim3(256), 0, s, (const unsigned long long *)p, (uint64_t)st.n_hits, (uint64_t)c->ix->d.n, (mp_seedpos *)c->seedpos.p);
        MP_TRACE(s, "stash_unpack_kernel");
    } else if (b_seed) HIPCHK(hipMemcpyAsync(c->seedpos.p, p, b_seed, hipMemcpyDeviceToDevice, s));
    p += up(b_seed);
    HIPCHK(hipMemcpyAsync(c->rs_base.p, p, b_rs, hipMemcpyDeviceToDevice, s)); p += up(b_rs);
    HIPCHK(hipMemcpyAsync(c->n_pos.p, p, b_np, hipMemcpyDeviceToDevice, s)); p += up(b_np);
    HIPCHK(hipMemcpyAsync(c->n_neg.p, p, b_np, hipMemcpyDeviceToDevice, s)); p += up(b_np);
    HIPCHK(hipMemcpyAsync(c->pair_done.p, p, b_pd, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n_hits_total = st.n_hits; c->last_seed = st.seed;
    c->seeded = true; c->deep_done = true; c->rescued = false;
    c->n_pair_hits_host = 0;                                                       // (that round's pair hits were reduced when they were made: mp_reduce_chunk_part(.., 1))
    c->n_single_hits_host = 0; c->n_rescue_hits_host = 0;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ per-reference reduction over chunks
static std::mutex g_tr_mu;
static int ensure_translate(mp_index *ix)
{
    std::lock_guard<std::mutex> lk(g_tr_mu);
    if (ix->d_tr_grid) return MP_OK;
    if (ix->grid.empty() || ix->translate.empty()) return MP_EINVAL;
    const size_t ng = ix->grid.size(), nr = ix->translate.size();
    std::vector<uint64_t> st(nr);
    std::vector<uint32_t> ch(nr);
    for (size_t i = 0; i < nr; i++) { st[i] = ix->translate[i].start; ch[i] = ix->translate[i].chr; }
    void *g = nullptr, *a = nullptr, *b = nullptr;
    if (hipMalloc(&g, ng * 4) != hipSuccess || hipMalloc(&a, nr * 8) != hipSuccess || hipMalloc(&b, nr * 4) != hipSuccess ||
        hipMemcpy(g, ix->grid.data(), ng * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(a, st.data(), nr * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b, ch.data(), nr * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(g); (void)hipFree(a); (void)hipFree(b); return MP_ENOMEM; }
    ix->d_tr_start = a; ix->d_tr_chr = b; ix->d_tr_grid = g;
    return MP_OK;
}

// grows a raw device array, keeping its first `keep` elements
static int grow_keys(unsigned long long *&p, size_t &cap, size_t need, size_t keep, hipStream_t s)
{
    if (need <= cap) return MP_OK;
    const size_t want = need + need / 2 + 4096;
    unsigned long long *q = nullptr;
    if (hipMalloc((void **)&q, want * 8) != hipSuccess) return MP_ENOMEM;
    if (p && keep) { if (hipMemcpyAsync(q, p, keep * 8, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(q); return MP_ENODEV; } }
    if (p) (void)hipFree(p);
    p = q; cap = want;
    return MP_OK;
}

// 0: the rounds keep their hit lists on the device (a caller that only wants the reduced list saves the transfers and the pinned
// buffers; the *_view calls then fail); 1 (default): they travel to the host as soon as a round is done.
extern "C" int mp_ctx_set_hit_copies(mp_ctx *c, int on)
{
    if (!c) return MP_EINVAL;
    c->copy_hits = on != 0;
    return MP_OK;
}

extern "C" int mp_reduce_begin(mp_ctx *c, uint32_t n_chunks)
{
    if (!c || n_chunks == 0 || n_chunks > MP_RED_MAX_CHUNKS) return MP_EINVAL;
    if (c->n_reads > (1u << 22)) return MP_ERANGE;                      // 22 bits of read id in a key
    c->red_n = 0; c->red_chunks = n_chunks;
    HIPCHK(hipSetDevice(c->ix->device));
    if (c->best_known.reserve((size_t)c->n_reads + 1)) return MP_ENOMEM;
    HIPCHK(hipMemsetAsync(c->best_known.p, 0, ((size_t)c->n_reads + 1) * 4, c->stream));
    return MP_OK;
}

// After mp_deep_dp_round and mp_rescue_round of the batch against the bound index chunk: adds the chunk's per-mate (sequence, score)
// entries to the batch's accumulator.  chunk_index counts from 0 in database order.
extern "C" int mp_reduce_chunk(mp_ctx *c, int megapath_mode, uint32_t chunk_index, mp_reduce_stats *out_stats, uint8_t *out_pair_class)
{
    return mp_reduce_chunk_part(c, megapath_mode, chunk_index, 3, out_stats, out_pair_class);
}

// The same in two parts, for the two passes of the -top rule (mp_chunk_stash): parts & 1 = the pair hits of the deep-DP round just done, parts & 2 = the
// rescue round's hits, the counts and the classes.  Every call raises the best score known per read by what it adds.
extern "C" int mp_reduce_chunk_part(mp_ctx *c, int megapath_mode, uint32_t chunk_index, int parts, mp_reduce_stats *out_stats, uint8_t *out_pair_class)
{
    if (!c || !c->deep_done || c->red_chunks == 0 || chunk_index >= c->red_chunks || (parts & 3) == 0 || !c->best_known.p) return MP_EINVAL;
    mp_index *ix = c->ix;
    if (ix->seq_names.size() > MP_RED_MAX_SEQ) return MP_ERANGE;        // 22 bits of sequence id in a key
    HIPCHK(hipSetDevice(ix->device));
    int rc = ensure_translate(ix);
    if (rc) return rc;
    hipStream_t s = c->stream;
    // the hit lists of the round are on their way to the host on the copy stream: they stay valid on the device meanwhile
    const uint32_t n_pairs = c->n_reads / 2;
    const size_t n_ph = (parts & 1) ? c->n_pair_hits_host : 0, n_rh = (parts & 2) && c->rescued ? c->n_rescue_hits_host : 0, n_sh = (parts & 2) && c->rescued ? c->n_single_hits_host : 0;
    const size_t add = 2 * n_ph + 2 * n_rh + n_sh;
    rc = grow_keys(c->red_keys, c->red_cap, c->red_n + add, c->red_n, s);
    if (rc) return rc;
    if (c->resc_done.reserve(n_pairs + 1) || c->red_counts.reserve(2)) return MP_ENOMEM;
    HIPCHK(hipMemsetAsync(c->resc_done.p, 0, ((size_t)n_pairs + 1) * 4, s));
    HIPCHK(hipMemsetAsync(c->red_counts.p, 0, 2 * sizeof(unsigned long long), s));
    const mp_dev_translate tr{(const uint32_t *)ix->d_tr_grid, (const uint64_t *)ix->d_tr_start, (const uint32_t *)ix->d_tr_chr, (uint32_t)ix->grid.size(), (uint32_t)ix->translate.size()};
    const uint32_t inv = c->red_chunks - 1 - chunk_index;
    unsigned long long *k = c->red_keys + c->red_n;
    if (n_ph) hipLaunchKernelGGL(reduce_pairs_kernel<false>, dim3(grid_for(n_ph, 256)), dim3(256), 0, s, (const void *)c->pair_hits.p, (uint32_t)n_ph, c->d_lens, tr, megapath_mode, inv, k, c->resc_done.p);
    if (n_rh) hipLaunchKernelGGL(reduce_pairs_kernel<true>, dim3(grid_for(n_rh, 256)), dim3(256), 0, s, (const void *)c->d_rescue_hits.p, (uint32_t)n_rh, c->d_lens, tr, megapath_mode, inv, k + 2 * n_ph, c->resc_done.p);
    if (n_sh) hipLaunchKernelGGL(reduce_singles_kernel, dim3(grid_for(n_sh, 256)), dim3(256), 0, s, (const mp_single_hit *)c->d_single_hits.p, (uint32_t)n_sh, c->d_lens, tr, megapath_mode, inv,
                                 (const uint32_t *)c->pair_done.p, (const uint32_t *)c->resc_done.p, k + 2 * n_ph + 2 * n_rh);
    if (add) hipLaunchKernelGGL(floor_best_kernel, dim3(grid_for(add, 256)), dim3(256), 0, s, (const unsigned long long *)k, (uint64_t)add, c->best_known.p, c->n_reads);
    if (out_pair_class && c->red_cls.reserve(n_pairs + 4)) return MP_ENOMEM;
    if (parts & 2) {
        hipLaunchKernelGGL(reduce_count_kernel, dim3(grid_for(n_pairs, 256)), dim3(256), 0, s, (const uint32_t *)c->pair_done.p, (const uint32_t *)c->resc_done.p, n_pairs, c->red_counts.p,
                           out_pair_class ? c->red_cls.p : (uint8_t *)nullptr);
        if (out_pair_class && n_pairs) HIPCHK(hipMemcpyAsync(out_pair_class, c->red_cls.p, n_pairs, hipMemcpyDeviceToHost, s));
    }
    MP_TRACE(s, "reduce kernels");
    HIPCHK(hipGetLastError());
    c->red_n += add;
    unsigned long long cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, c->red_counts.p, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (out_stats) { out_stats->n_deep_pairs = cnt[0]; out_stats->n_rescued_pairs = cnt[1]; out_stats->n_unpaired_pairs = n_pairs - cnt[0] - cnt[1]; out_stats->n_entries = add; }
    return MP_OK;
}

// After the last chunk: the batch's entries, one 64-bit key each (read << 42 | chunks after the entry's << 32 | sequence id << 10 | 1023 - score),
// sorted, one per (read, chunk, sequence) -- the best score --, in a pinned buffer owned by the context (valid until the next mp_reduce_begin).
extern "C" int mp_reduce_finish(mp_ctx *c, const uint64_t **out_keys, uint64_t *out_n)
{
    if (!c || !out_keys || !out_n || c->red_chunks == 0) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    hipStream_t s = c->stream;
    *out_keys = nullptr; *out_n = 0;
    const size_t n = c->red_n;
    if (n == 0) return MP_OK;
    if (n >= (1ull << 31)) return MP_ERANGE;
    int rc = grow_keys(c->red_sorted, c->red_sorted_cap, n, 0, s);
    if (rc) return rc;
    size_t tb = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, tb, c->red_keys, c->red_sorted, (int)n, 0, 64, s) != hipSuccess) return MP_ENODEV;
    if (c->cub_tmp.reserve(tb + 256)) return MP_ENOMEM;
    if (hipcub::DeviceRadixSort::SortKeys(c->cub_tmp.p, tb, c->red_keys, c->red_sorted, (int)n, 0, 64, s) != hipSuccess) return MP_ENODEV;
    if (c->red_flag.reserve(n + 2) || c->red_slot.reserve(n + 2)) return MP_ENOMEM;
    hipLaunchKernelGGL(reduce_first_flag_kernel, dim3(grid_for(n + 1, 256)), dim3(256), 0, s, (const unsigned long long *)c->red_sorted, (uint64_t)n, c->red_flag.p);
    rc = exclusive_scan_u32(c, c->red_flag.p, c->red_slot.p, (uint32_t)n + 1);
    if (rc) return rc;
    uint32_t n_out = 0;
    HIPCHK(hipMemcpyAsync(&n_out, c->red_slot.p + n, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (n_out) {
        rc = grow_keys(c->red_out, c->red_out_cap, n_out, 0, s);
        if (rc) return rc;
        rc = reserve_pinned(c->h_red, c->h_red_cap, n_out);
        if (rc) return rc;
        hipLaunchKernelGGL(reduce_compact_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, (const unsigned long long *)c->red_sorted, (const uint32_t *)c->red_flag.p,
                           (const uint32_t *)c->red_slot.p, (uint64_t)n, c->red_out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->h_red, c->red_out, (size_t)n_out * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    *out_keys = (const uint64_t *)c->h_red; *out_n = n_out;
    return MP_OK;
}

// ------------------------------------------------------------------------------------------------ coverage / spike filter
struct mp_coverage {
    const mp_index *ix = nullptr;
    uint32_t n_seq = 0;
    uint64_t total = 0;                       // slots: every sequence's actual length + 1
    std::vector<uint64_t> h_off, h_len;
    uint64_t *d_off = nullptr, *d_len = nullptr;
    int32_t *d_depth = nullptr;               // difference array, then depth
    uint32_t *d_mask = nullptr;               // spike mask, then its inclusive prefix sum
    double *d_max = nullptr;
    std::vector<double> mean, var, max_depth;
    bool finished = false, mask_scanned = false;
    uint64_t n_added = 0;
};

// inclusive prefix sum over more entries than the library's int count allows: parts of 2^30 with the carry added
static int scan_u32_inplace(uint32_t *d, uint64_t n)
{
    const uint64_t part = 1ull << 30;
    size_t tb = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb, d, d, (int)std::min<uint64_t>(n, part), 0);
    void *tmp = nullptr;
    if (hipMalloc(&tmp, tb + 256) != hipSuccess) return MP_ENOMEM;
    uint32_t carry = 0;
    for (uint64_t lo = 0; lo < n; lo += part) {
        const uint64_t cnt = std::min<uint64_t>(part, n - lo);
        if (hipcub::DeviceScan::InclusiveSum(tmp, tb, d + lo, d + lo, (int)cnt, 0) != hipSuccess) { (void)hipFree(tmp); return MP_ENODEV; }
        if (carry) hipLaunchKernelGGL(add_offset_kernel, dim3(grid_for(cnt, 256)), dim3(256), 0, 0, d + lo, cnt, carry);
        if (hipMemcpy(&carry, d + lo + cnt - 1, 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(tmp); return MP_ENODEV; }
    }
    (void)hipFree(tmp);
    return hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess ? MP_OK : MP_ENODEV;
}

extern "C" int mp_coverage_create(const mp_index *ix, mp_coverage **out)
{
    if (!ix || !out || ix->seq_names.empty()) return MP_EINVAL;
    HIPCHK(hipSetDevice(ix->device));
    mp_coverage *c = new (std::nothrow) mp_coverage();
    if (!c) return MP_ENOMEM;
    c->ix = ix; c->n_seq = (uint32_t)ix->seq_names.size();
    c->h_off.resize(c->n_seq); c->h_len.resize(c->n_seq);
    uint64_t at = 0;
    for (uint32_t s = 0; s < c->n_seq; s++) {
        // the coordinates of a BAM record: positions in the sequence as the FASTA had it, removed N-runs included (.tra tail)
        c->h_len[s] = s < ix->seq_actual_len.size() && ix->seq_actual_len[s] ? ix->seq_actual_len[s] : ix->seq_len[s];
        c->h_off[s] = at; at += c->h_len[s] + 1;
    }
    c->total = at;
    if (hipMalloc((void **)&c->d_off, c->n_seq * 8) != hipSuccess || hipMalloc((void **)&c->d_len, c->n_seq * 8) != hipSuccess ||
        hipMalloc((void **)&c->d_depth, c->total * 4) != hipSuccess || hipMalloc((void **)&c->d_max, c->n_seq * 8) != hipSuccess) { mp_coverage_destroy(c); return MP_ENOMEM; }
    HIPCHK(hipMemcpy(c->d_off, c->h_off.data(), c->n_seq * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_len, c->h_len.data(), c->n_seq * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->d_depth, 0, c->total * 4));
    *out = c;
    return MP_OK;
}

extern "C" int mp_coverage_destroy(mp_coverage *c)
{
    if (!c) return MP_OK;
    (void)hipSetDevice(c->ix->device);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_len) (void)hipFree(c->d_len);
    if (c->d_depth) (void)hipFree(c->d_depth);
    if (c->d_mask) (void)hipFree(c->d_mask);
    if (c->d_max) (void)hipFree(c->d_max);
    delete c;
    return MP_OK;
}

// uploads n intervals and runs `fn(d_seq, d_start, d_len)` on them
template <typename F>
static int with_intervals(const mp_coverage *c, const uint32_t *seq_id_1based, const uint64_t *start0, const uint32_t *len, uint64_t n, F &&fn)
{
    std::vector<uint32_t> s0(n);
    for (uint64_t i = 0; i < n; i++) { if (seq_id_1based[i] == 0 || seq_id_1based[i] > c->n_seq) return MP_EINVAL; s0[i] = seq_id_1based[i] - 1; }
    uint32_t *d_seq = nullptr, *d_l = nullptr; uint64_t *d_st = nullptr;
    int rc = MP_OK;
    if (hipMalloc((void **)&d_seq, n * 4) != hipSuccess || hipMalloc((void **)&d_l, n * 4) != hipSuccess || hipMalloc((void **)&d_st, n * 8) != hipSuccess) rc = MP_ENOMEM;
    if (!rc && (hipMemcpy(d_seq, s0.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_l, len, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_st, start0, n * 8, hipMemcpyHostToDevice) != hipSuccess)) rc = MP_ENODEV;
    if (!rc) rc = fn(d_seq, d_st, d_l);
    if (d_seq) (void)hipFree(d_seq);
    if (d_l) (void)hipFree(d_l);
    if (d_st) (void)hipFree(d_st);
    return rc;
}

extern "C" int mp_coverage_add(mp_coverage *c, const uint32_t *seq_id_1based, const uint64_t *start0, const uint32_t *len, uint64_t n)
{
    if (!c || c->finished || (n && (!seq_id_1based || !start0 || !len))) return MP_EINVAL;
    if (n == 0) return MP_OK;
    HIPCHK(hipSetDevice(c->ix->device));
    c->n_added += n;
    return with_intervals(c, seq_id_1based, start0, len, n, [&](const uint32_t *d_seq, const uint64_t *d_st, const uint32_t *d_l) {
        hipLaunchKernelGGL(cov_add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, d_seq, d_st, d_l, n, c->d_off, c->d_len, c->d_depth);
        return hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess ? MP_OK : MP_ENODEV;
    });
}

extern "C" int mp_coverage_finish(mp_coverage *c, double stdev_k)
{
    if (!c || c->finished) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    int rc = scan_u32_inplace((uint32_t *)c->d_depth, c->total);                     // difference array -> depth
    if (rc) return rc;
    unsigned long long *d_sums = nullptr;
    HIPCHK(hipMalloc((void **)&d_sums, (size_t)c->n_seq * 16));
    HIPCHK(hipMemset(d_sums, 0, (size_t)c->n_seq * 16));
    const uint32_t slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4096, (c->total / c->n_seq) / 65536 + 1));
    hipLaunchKernelGGL(cov_moments_kernel, dim3(c->n_seq * slices), dim3(256), 0, 0, c->d_depth, c->d_off, c->d_len, c->n_seq, slices, d_sums);
    std::vector<unsigned long long> sums((size_t)c->n_seq * 2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(sums.data(), d_sums, sums.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_sums);
    if (e != hipSuccess) return MP_ENODEV;
    c->mean.resize(c->n_seq); c->var.resize(c->n_seq); c->max_depth.resize(c->n_seq);
    for (uint32_t s = 0; s < c->n_seq; s++) {
        // mean and population variance over ALL bases of the sequence (genomecov -bga lists the zero-depth stretches too,
        // cc/genomeCovFilter.cpp:58-76 weights every interval by its length); exact integer sums, one rounding at the end
        const long double N = (long double)c->h_len[s];
        const long double m = N > 0 ? (long double)sums[2 * (size_t)s] / N : 0.0L;
        long double v = N > 0 ? (long double)sums[2 * (size_t)s + 1] / N - m * m : 0.0L;
        if (v < 0) v = 0;
        c->mean[s] = (double)m; c->var[s] = (double)v;
        c->max_depth[s] = (double)(m + (long double)stdev_k * sqrtl(v));
    }
    HIPCHK(hipMemcpy(c->d_max, c->max_depth.data(), (size_t)c->n_seq * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&c->d_mask, c->total * 4));
    hipLaunchKernelGGL(cov_mask_kernel, dim3(grid_for(c->total, 256)), dim3(256), 0, 0, c->d_depth, c->d_off, c->d_len, c->d_max, c->n_seq, c->total, c->d_mask);
    HIPCHK(hipGetLastError()); HIPCHK(hipDeviceSynchronize());
    c->finished = true;
    return MP_OK;
}

extern "C" int mp_coverage_seq_stats(const mp_coverage *c, uint32_t seq_id_1based, double *out_mean, double *out_variance, double *out_max_depth)
{
    if (!c || !c->finished || seq_id_1based == 0 || seq_id_1based > c->n_seq) return MP_EINVAL;
    if (out_mean) *out_mean = c->mean[seq_id_1based - 1];
    if (out_variance) *out_variance = c->var[seq_id_1based - 1];
    if (out_max_depth) *out_max_depth = c->max_depth[seq_id_1based - 1];
    return MP_OK;
}

extern "C" int mp_coverage_spike_intervals(mp_coverage *c, uint32_t *out_seq_id, uint64_t *out_start0, uint64_t *out_end0, uint64_t cap, uint64_t *out_n)
{
    if (!c || !c->finished || !out_n) return MP_EINVAL;
    HIPCHK(hipSetDevice(c->ix->device));
    const uint32_t edge_cap = 1u << 22;
    unsigned long long *d_edges = nullptr; unsigned int *d_n = nullptr, n_e = 0;
    if (hipMalloc((void **)&d_edges, (size_t)edge_cap * 8) != hipSuccess || hipMalloc((void **)&d_n, 4) != hipSuccess) { if (d_edges) (void)hipFree(d_edges); return MP_ENOMEM; }
    HIPCHK(hipMemset(d_n, 0, 4));
    // the mask may already be prefix-summed (after a query): edges are where the running count changes -- recompute from the depth
    uint32_t *d_m = nullptr;
    if (hipMalloc((void **)&d_m, c->total * 4) != hipSuccess) { (void)hipFree(d_edges); (void)hipFree(d_n); return MP_ENOMEM; }
    hipLaunchKernelGGL(cov_mask_kernel, dim3(grid_for(c->total, 256)), dim3(256), 0, 0, c->d_depth, c->d_off, c->d_len, c->d_max, c->n_seq, c->total, d_m);
    hipLaunchKernelGGL(cov_edges_kernel, dim3(grid_for(c->total, 256)), dim3(256), 0, 0, d_m, c->total, d_edges, edge_cap, d_n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&n_e, d_n, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> edges(std::min<uint32_t>(n_e, edge_cap));
    if (e == hipSuccess && !edges.empty()) e = hipMemcpy(edges.data(), d_edges, edges.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_edges); (void)hipFree(d_n); (void)hipFree(d_m);
    if (e != hipSuccess) return MP_ENODEV;
    if (n_e > edge_cap) return MP_ERANGE;
    std::sort(edges.begin(), edges.end());
    uint64_t w = 0;
    for (size_t i = 0; i + 1 < edges.size() + 1 && i < edges.size(); i += 2) {
        if (!(edges[i] & 1ull) || i + 1 >= edges.size()) return MP_ERANGE;         // starts and ends alternate (a sequence's last slot is never a spike)
        const uint64_t a = edges[i] >> 1, b = edges[i + 1] >> 1;
        const uint32_t s = (uint32_t)(std::upper_bound(c->h_off.begin(), c->h_off.end(), a) - c->h_off.begin()) - 1;
        if (out_seq_id && out_start0 && out_end0) {
            if (w >= cap) return MP_ERANGE;
            out_seq_id[w] = s + 1; out_start0[w] = a - c->h_off[s]; out_end0[w] = b - c->h_off[s];
        }
        w++;
    }
    *out_n = w;
    return MP_OK;
}

extern "C" int mp_coverage_spike_bases(mp_coverage *c, const uint32_t *seq_id_1based, const uint64_t *start0, const uint32_t *len, uint64_t n, uint32_t *out_covered)
{
    if (!c || !c->finished || (n && (!seq_id_1based || !start0 || !len || !out_covered))) return MP_EINVAL;
    if (n == 0) return MP_OK;
    HIPCHK(hipSetDevice(c->ix->device));
    if (!c->mask_scanned) {                   // first query: the mask becomes its own inclusive prefix sum
        int rc = scan_u32_inplace(c->d_mask, c->total);
        if (rc) return rc;
        c->mask_scanned = true;
    }
    uint32_t *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, n * 4));
    int rc = with_intervals(c, seq_id_1based, start0, len, n, [&](const uint32_t *d_seq, const uint64_t *d_st, const uint32_t *d_l) {
        hipLaunchKernelGGL(cov_query_kernel, dim3(grid_for(n, 256)), dim3(256), 0, 0, d_seq, d_st, d_l, n, c->d_off, c->d_len, c->d_mask, d_out);
        if (hipGetLastError() != hipSuccess) return MP_ENODEV;
        return hipMemcpy(out_covered, d_out, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? MP_OK : MP_ENODEV;
    });
    (void)hipFree(d_out);
    return rc;
}
