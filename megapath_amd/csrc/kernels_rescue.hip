// megapath_amd/csrc/kernels_rescue.hip -- product code: task generation for the rescue passes of pairs the deep-DP
// round left unaligned (reference soap4/alignment.cpp:137-250).  The DP itself is the same dp_fill / dp_trace kernels.
//
//   mark_done_kernel          pairs that produced at least one deep-DP hit
//   list_failed_kernel        compact list of reads belonging to failed pairs
//   single_cand_kernel        transferSeed + singleMerge: pooled seeds -> <= 200 DP candidates per read
//                             (DV-DPForSingleReads.cpp:121-215, DV-DPfunctions.cpp:295-342)
//   single_task_kernel        SingleEndAlgnBatch::pack windows (DV-DPfunctions.cpp:401-445)
//   mate_rescue_task_kernel   last four single-end hits per read in (score desc, window start asc) order
//                             (DV-SemiDP.cpp:37-44, DV-DPfunctions.cpp:1048-1079) -> HalfEndAlgnBatch::pack windows (:1151-1231)
#include <hip/hip_runtime.h>
#include "mp_internal.h"

__global__ void mark_done_kernel(const mp_pair_hit *__restrict__ hits, uint32_t n_hits, uint32_t *__restrict__ pair_done)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_hits) pair_done[hits[t].pair_id] = 1u;
}

// failed_flag[r] = 1 for both reads of a pair without deep-DP hit; an exclusive scan of it gives the slot of each read
__global__ void failed_flag_kernel(const uint32_t *__restrict__ pair_done, uint32_t n_reads, uint32_t *__restrict__ failed_flag)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) failed_flag[r] = (r < n_reads && !pair_done[r >> 1]) ? 1u : 0u;
}

__global__ void list_failed_kernel(const uint32_t *__restrict__ failed_flag, const uint32_t *__restrict__ failed_off, uint32_t n_reads,
                                   uint32_t *__restrict__ failed_reads)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads && failed_flag[r]) failed_reads[failed_off[r]] = r;
}

struct mp_single_cand { uint64_t pos; uint32_t strand; uint32_t seed_len; };

#define MP_SINGLE_MAX 200      // DV-DPForSingleReads.cpp:200

// One lane per failed read.  The read's kept seeds (the seed pool) sit at seedpos[rs_base[2r] ..) sorted by
// (strand, pos), which is the order RadixTraitsCandidateInfo produces (DV-DPForSingleReads.cpp:108-118).
// Candidates are written to cand[rs_base[2r] ..) (same capacity) and counted in n_cand[slot].
__global__ void __launch_bounds__(128)
single_cand_kernel(const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const mp_seedpos *__restrict__ seedpos,
                   const uint32_t *__restrict__ rs_base, const uint32_t *__restrict__ n_pos, const uint32_t *__restrict__ n_neg,
                   mp_single_cand *__restrict__ cand, uint32_t *__restrict__ n_cand)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot > n_failed) return;
    if (slot == n_failed) { n_cand[slot] = 0; return; }
    const uint32_t r = failed_reads[slot];
    const mp_seedpos *sp = seedpos + rs_base[2 * r];
    const int n = (int)(n_pos[r] + n_neg[r]);
    mp_single_cand *out = cand + rs_base[2 * r];
    int nm = 0;
    // singleMerge: drop seeds < 17, merge followers within DPS_DIVIDE_GAP (5) on the same strand keeping the longer one
    for (int p = 0; p < n; p++) {
        const uint32_t slen = sp[p].paired_seed_length & 0x7FFFFFFFu;
        if (slen < 17) continue;
        mp_single_cand cur; cur.pos = sp[p].pos; cur.strand = (sp[p].strand_read_id >> 31) + 1; cur.seed_len = slen;
        while (p + 1 < n) {
            const uint32_t nstrand = (sp[p + 1].strand_read_id >> 31) + 1, nlen = sp[p + 1].paired_seed_length & 0x7FFFFFFFu;
            if (sp[p + 1].pos < cur.pos + 5 && cur.strand == nstrand) {
                if (nlen > cur.seed_len) { cur.pos = sp[p + 1].pos; cur.seed_len = nlen; }
            } else break;
            ++p;
        }
        out[nm++] = cur;
    }
    // longest first (stable insertion sort: what std::sort does below 16 elements), drop < 0.6 x best, cap at 200
    for (int i = 1; i < nm; i++) {
        mp_single_cand t = out[i];
        int j = i - 1;
        while (j >= 0 && out[j].seed_len < t.seed_len) { out[j + 1] = out[j]; j--; }
        out[j + 1] = t;
    }
    while (nm > 0 && (double)out[nm - 1].seed_len < (double)out[0].seed_len * 0.6) nm--;
    if (nm > MP_SINGLE_MAX) nm = MP_SINGLE_MAX;
    n_cand[slot] = (uint32_t)nm;
}

__global__ void __launch_bounds__(128)
single_task_kernel(const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const mp_single_cand *__restrict__ cand,
                   const uint32_t *__restrict__ rs_base, const uint32_t *__restrict__ n_cand, const uint32_t *__restrict__ task_off,
                   const uint32_t *__restrict__ lens, uint64_t text_len, mp_dp_task *__restrict__ tasks)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_failed) return;
    const uint32_t r = failed_reads[slot];
    const mp_single_cand *c = cand + rs_base[2 * r];
    const uint32_t n = n_cand[slot], base = task_off[slot];
    const int len = (int)lens[r];
    for (uint32_t j = 0; j < n; j++) {
        mp_dp_task k; k.read_id = r; k.strand = c[j].strand; k.reserved = 0;
        mp_dp_window(c[j].pos, len, text_len, &k.win_start, &k.win_len);
        tasks[base + j] = k;
    }
}

// One lane per failed read: choose the last four passing single-end hits in (score desc, window start asc) order and
// build the mate's search window for each.  Four task slots per failed read; unused slots get win_len = 0.
__global__ void __launch_bounds__(128)
mate_rescue_task_kernel(const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const uint32_t *__restrict__ task_off,
                        const mp_dp_task *__restrict__ stasks, const mp_dp_result *__restrict__ sres, const uint32_t *__restrict__ lens,
                        uint64_t text_len, int insert_low, int insert_high, mp_dp_task *__restrict__ rtasks, uint32_t *__restrict__ rsrc)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_failed) return;
    const uint32_t r = failed_reads[slot], un = r ^ 1u;
    const uint32_t b0 = task_off[slot], b1 = task_off[slot + 1];
    const int cutoff = mp_dp_cutoff((int)lens[r]);
    // keep the 4 hits that sort last: smallest score first, then larger window start
    int pick[4] = {-1, -1, -1, -1};
    int np = 0;
    auto later = [&](uint32_t a, uint32_t b) {        // does hit a sort after hit b?
        if (sres[a].score != sres[b].score) return sres[a].score < sres[b].score;
        return stasks[a].win_start > stasks[b].win_start;
    };
    for (uint32_t t = b0; t < b1; t++) {
        if (sres[t].score < cutoff || sres[t].score <= 0) continue;
        // insert into pick[] kept in ascending sorted order of "sorts later" rank (pick[0] = the very last one)
        int q = np < 4 ? np : 4;
        while (q > 0 && later(t, (uint32_t)pick[q - 1])) { if (q < 4) pick[q] = pick[q - 1]; q--; }
        if (q < 4) { pick[q] = (int)t; if (np < 4) np++; }
    }
    const uint64_t alen = lens[r], ulen = lens[un];
    for (int q = 0; q < 4; q++) {
        mp_dp_task k; k.read_id = un; k.strand = 1; k.win_start = 0; k.win_len = 0; k.reserved = 0;
        uint32_t src = 0xFFFFFFFFu;
        if (q < np) {
            const uint32_t t = (uint32_t)pick[q];
            const uint64_t algn = stasks[t].win_start + sres[t].hit_loc;
            if (stasks[t].strand == 1) {                       // aligned mate is the left leg (DV-DPfunctions.cpp:1182-1205)
                const uint64_t r_end = algn + (uint64_t)(int64_t)insert_high;
                uint64_t r_start = algn + (uint64_t)(int64_t)insert_low - ulen;
                if (r_start < algn) r_start = algn;
                if (r_start < text_len && r_end <= text_len) { k.strand = 2; k.win_start = r_start; k.win_len = (uint32_t)(r_end - r_start); src = t; }
            } else {                                           // aligned mate is the right leg (:1207-1230)
                const uint64_t l_start = algn + alen - (uint64_t)(int64_t)insert_high;
                uint64_t l_end = algn + alen - (uint64_t)(int64_t)insert_low + ulen;
                if (l_end >= algn + alen) l_end = algn + alen - 1;
                if (l_start < text_len && l_end <= text_len) { k.strand = 1; k.win_start = l_start; k.win_len = (uint32_t)(l_end - l_start); src = t; }
            }
        }
        rtasks[4 * slot + q] = k;
        rsrc[4 * slot + q] = src;
    }
}
