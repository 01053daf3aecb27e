// megapath_amd/csrc/kernels_rescue.hip -- product code: task generation for the rescue passes of pairs the deep-DP
// round left unaligned (reference soap4/alignment.cpp:137-250).  The DP itself is the same dp_fill / dp_trace kernels.
//
//   list_failed_kernel        compact list of reads belonging to failed pairs
//   single_cand_kernel        transferSeed + singleMerge: pooled seeds -> <= 200 DP candidates per read
//                             (DV-DPForSingleReads.cpp:121-215, DV-DPfunctions.cpp:295-342)
//   single_task_kernel        SingleEndAlgnBatch::pack windows (DV-DPfunctions.cpp:401-445)
//   mate_rescue_task_kernel   last four single-end hits per read in (score desc, window start asc) order
//                             (DV-SemiDP.cpp:37-44, DV-DPfunctions.cpp:1048-1079) -> HalfEndAlgnBatch::pack windows (:1151-1231)
#include <hip/hip_runtime.h>
#include "mp_internal.h"

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

// std::sort with the comparator "longer seed first" as the reference's libstdc++ (g++ 11) carries it out, on one lane: the
// permutation of equal-length seeds must be the reference's because only the first 200 candidates of a read are kept
// (DV-DPForSingleReads.cpp:200).  Introsort: median of (first+1, middle, last-1) to `first`, Hoare partition of the rest,
// right part first (explicit stack instead of recursion), ranges of <= 16 left alone; heap sort of a range when
// 2*floor(log2 n) levels are used up; then one insertion pass, guarded for the first 16 elements only.
// Records move as one 16-byte register quad (x, y = pos; z = strand; w = seed_len): no struct temporaries, no scratch memory.
typedef uint4 mp_cand_q;
// The array being sorted: records `stride` quads apart -- 1 for a read's slice of the candidate array in global memory, the block's lane count for a
// column of the LDS staging area of single_cand_kernel (element e of lane l at [e][l]: 16-byte accesses without bank conflicts).  Same algorithm, same
// comparisons, same result either way; only the latency of every dependent access differs (a read's ~30 candidates are moved some hundred times).
struct mp_cand_arr {
    mp_cand_q *p; int stride;
    __device__ __forceinline__ mp_cand_arr at(int first) const { return mp_cand_arr{p + (size_t)first * stride, stride}; }
};
__device__ __forceinline__ mp_cand_q mp_lsl_ld(const mp_cand_arr &a, int i) { return a.p[(size_t)i * a.stride]; }
__device__ __forceinline__ void mp_lsl_st(const mp_cand_arr &a, int i, mp_cand_q v) { a.p[(size_t)i * a.stride] = v; }
__device__ __forceinline__ uint32_t mp_lsl_len(const mp_cand_arr &a, int i) { return a.p[(size_t)i * a.stride].w; }
__device__ __forceinline__ void mp_lsl_swap(const mp_cand_arr &a, int i, int j) { const mp_cand_q t = mp_lsl_ld(a, i); mp_lsl_st(a, i, mp_lsl_ld(a, j)); mp_lsl_st(a, j, t); }
static_assert(sizeof(mp_single_cand) == 16, "candidates move as 16-byte quads");

// __adjust_heap + __push_heap of libstdc++ (bits/stl_heap.h) with "longer seed first" as the order
__device__ __forceinline__ void mp_lsl_sift(const mp_cand_arr &a, int hole, int len, mp_cand_q v)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (mp_lsl_len(a, child) > mp_lsl_len(a, child - 1)) child--;
        mp_lsl_st(a, hole, mp_lsl_ld(a, child)); hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); mp_lsl_st(a, hole, mp_lsl_ld(a, child - 1)); hole = child - 1; }
    int parent = (hole - 1) / 2;
    while (hole > top && mp_lsl_len(a, parent) > v.w) { mp_lsl_st(a, hole, mp_lsl_ld(a, parent)); hole = parent; parent = (hole - 1) / 2; }
    mp_lsl_st(a, hole, v);
}

// The stack of pending ranges lives in LDS (`stk`, one column per lane, MP_LSL_STACK entries `stride` words apart): private
// arrays would be scratch memory, which no kernel of this library uses.  An entry packs first | last << 12 | depth << 24
// (a read has < 4096 candidates: at most MP_SEED_BIG seeds per strand); at most one entry per partition level is pending.
#define MP_LSL_STACK 26
__device__ __forceinline__ void mp_sort_longer_seed(const mp_cand_arr &a, int n, uint32_t *stk, int stride)
{
    if (n <= 1) return;
    if (n > 16 && n < 4096) {
        int sp = 0;
        stk[0] = (uint32_t)n << 12 | (uint32_t)(2 * (31 - __clz(n))) << 24; sp = 1;
        while (sp > 0) {
            --sp;
            const uint32_t e = stk[sp * stride];
            const int first = (int)(e & 4095u);
            int last = (int)((e >> 12) & 4095u), depth = (int)(e >> 24);
            // the reference recurses into [cut, last) before it goes on with [first, cut): the two ranges are disjoint, so doing
            // the left ones first and stacking the right ones gives the same array
            while (last - first > 16) {
                if (depth == 0) {
                    const mp_cand_arr h = a.at(first);
                    const int len = last - first;
                    for (int parent = (len - 2) / 2;; parent--) { mp_lsl_sift(h, parent, len, mp_lsl_ld(h, parent)); if (parent == 0) break; }
                    for (int e2 = len; e2 > 1;) { --e2; const mp_cand_q v = mp_lsl_ld(h, e2); mp_lsl_st(h, e2, mp_lsl_ld(h, 0)); mp_lsl_sift(h, 0, e2, v); }
                    break;
                }
                --depth;
                const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
                const uint32_t lx = mp_lsl_len(a, x), ly = mp_lsl_len(a, y), lz = mp_lsl_len(a, z);
                // __move_median_to_first with comp(p, q) = len(p) > len(q)
                int med;
                if (lx > ly) med = ly > lz ? y : (lx > lz ? z : x);
                else med = lx > lz ? x : (ly > lz ? z : y);
                mp_lsl_swap(a, first, med);
                const uint32_t pivot = mp_lsl_len(a, first);
                int lo = first + 1, hi = last;
                for (;;) {
                    while (mp_lsl_len(a, lo) > pivot) ++lo;
                    --hi;
                    while (pivot > mp_lsl_len(a, hi)) --hi;
                    if (!(lo < hi)) break;
                    mp_lsl_swap(a, lo, hi);
                    ++lo;
                }
                if (sp < MP_LSL_STACK) { stk[sp * stride] = (uint32_t)lo | (uint32_t)last << 12 | (uint32_t)depth << 24; sp++; }
                last = lo;
            }
        }
    }
    // __final_insertion_sort: guarded for the first 16 elements, unguarded after them
    const int guarded = n > 16 ? 16 : n;
    for (int i = 1; i < n; i++) {
        const mp_cand_q v = mp_lsl_ld(a, i);
        int j = i;
        if (i < guarded && v.w > mp_lsl_len(a, 0)) { for (; j > 0; j--) mp_lsl_st(a, j, mp_lsl_ld(a, j - 1)); }
        else while (v.w > mp_lsl_len(a, j - 1)) { mp_lsl_st(a, j, mp_lsl_ld(a, j - 1)); j--; }
        mp_lsl_st(a, j, v);
    }
}

// Test aid (mp_test_sort_longer_seed): one lane per array; array t holds off[t+1] - off[t] seeds whose `pos` carries the input index
__global__ void sort_longer_seed_test_kernel(mp_single_cand *__restrict__ cand, const uint32_t *__restrict__ off, uint32_t n_arrays)
{
    __shared__ uint32_t s_stk[MP_LSL_STACK][64];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_arrays) mp_sort_longer_seed(mp_cand_arr{reinterpret_cast<mp_cand_q *>(cand + off[t]), 1}, (int)(off[t + 1] - off[t]), &s_stk[0][threadIdx.x], 64);
}

// One lane per failed read.  The read's kept seeds (the seed pool) sit at seedpos[rs_base[2r] ..) sorted by
// (strand, pos), which is the order RadixTraitsCandidateInfo produces (DV-DPForSingleReads.cpp:108-118).
// Candidates are written to cand[rs_base[2r] ..) (same capacity) and counted in n_cand[slot].
#define MP_SCAND_LDS 40        // candidates of a read staged in LDS (a read of the NT stage has ~30 seeds; more: the read's slice of the global array as before)
__global__ void __launch_bounds__(64)
single_cand_kernel(const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const mp_seedpos *__restrict__ seedpos,
                   const uint32_t *__restrict__ rs_base, const uint32_t *__restrict__ n_pos, const uint32_t *__restrict__ n_neg,
                   mp_single_cand *__restrict__ cand, uint32_t *__restrict__ n_cand)
{
    __shared__ uint32_t s_stk[MP_LSL_STACK][64];
    __shared__ mp_cand_q s_c[MP_SCAND_LDS][64];              // [element][lane]: 40 KB
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot > n_failed) return;
    if (slot == n_failed) { n_cand[slot] = 0; return; }
    const uint32_t r = failed_reads[slot];
    const mp_seedpos *sp = seedpos + rs_base[2 * r];
    const int n = (int)(n_pos[r] + n_neg[r]);
    mp_single_cand *out = cand + rs_base[2 * r];
    const bool in_lds = n <= MP_SCAND_LDS;
    const mp_cand_arr arr = in_lds ? mp_cand_arr{&s_c[0][threadIdx.x], 64} : mp_cand_arr{reinterpret_cast<mp_cand_q *>(out), 1};
    // the read's seed positions: into the LDS column first, eight 16-byte loads in flight at a time (the merge below reads them one after the other, each
    // decision depending on the last: straight from global memory that is one memory latency per seed); a candidate is written at or before the seed it
    // was read from, so the merge works in place
    static_assert(sizeof(mp_seedpos) == 16, "seed positions move as 16-byte quads");
    const mp_cand_arr raw = in_lds ? arr : mp_cand_arr{const_cast<mp_cand_q *>(reinterpret_cast<const mp_cand_q *>(sp)), 1};
    if (in_lds)
        for (int p0 = 0; p0 < n; p0 += 8) {
            mp_cand_q q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) if (p0 + j < n) q[j] = reinterpret_cast<const mp_cand_q *>(sp)[p0 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) if (p0 + j < n) mp_lsl_st(arr, p0 + j, q[j]);
        }
    int nm = 0;
    // singleMerge: drop seeds < 17, merge followers within DPS_DIVIDE_GAP (5) on the same strand keeping the longer one
    // (a seed position as a quad: x, y = pos; z = strand_read_id; w = paired_seed_length)
    for (int p = 0; p < n; p++) {
        const mp_cand_q a = mp_lsl_ld(raw, p);
        const uint32_t slen = a.w & 0x7FFFFFFFu;
        if (slen < 17) continue;
        mp_single_cand cur; cur.pos = ((uint64_t)a.y << 32) | a.x; cur.strand = (a.z >> 31) + 1; cur.seed_len = slen;
        while (p + 1 < n) {
            const mp_cand_q b = mp_lsl_ld(raw, p + 1);
            const uint32_t nstrand = (b.z >> 31) + 1, nlen = b.w & 0x7FFFFFFFu;
            const uint64_t npos = ((uint64_t)b.y << 32) | b.x;
            if (npos < cur.pos + 5 && cur.strand == nstrand) {
                if (nlen > cur.seed_len) { cur.pos = npos; cur.seed_len = nlen; }
            } else break;
            ++p;
        }
        mp_lsl_st(arr, nm++, mp_cand_q{(uint32_t)cur.pos, (uint32_t)(cur.pos >> 32), cur.strand, cur.seed_len});
    }
    // longest first, in the very order std::sort(.., LongerSeedLength()) leaves equal lengths in (DV-DPfunctions.cpp:334):
    // the cut at 200 below makes that order output-visible.  Then drop < 0.6 x best, cap at 200.
    mp_sort_longer_seed(arr, nm, &s_stk[0][threadIdx.x], 64);
    while (nm > 0 && (double)mp_lsl_len(arr, nm - 1) < (double)mp_lsl_len(arr, 0) * 0.6) nm--;
    if (nm > MP_SINGLE_MAX) nm = MP_SINGLE_MAX;
    if (in_lds) for (int i = 0; i < nm; i++) *reinterpret_cast<mp_cand_q *>(out + i) = mp_lsl_ld(arr, i);
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
        if (q < np && ulen >= MP_MIN_ALIGNABLE_LEN) {           // a mate below the cutoff floor cannot be rescued: no instance for it
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

// ------------------------------------------------------------------------------------------------ hit lists of the rescue passes
// The host wants only the instances that passed (a few per failed pair out of ~60 tasks): flag, exclusive scan, emit -- in task
// order, so the lists are run-to-run identical -- into a device buffer that travels to pinned host memory in one DMA.
__global__ void single_hit_flag_kernel(const mp_dp_task *__restrict__ tasks, const mp_dp_result *__restrict__ res, const uint32_t *__restrict__ lens,
                                       uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n) return;
    flag[t] = (t < n && res[t].score > 0 && res[t].score >= mp_dp_cutoff((int)lens[tasks[t].read_id])) ? 1u : 0u;
}

__global__ void single_hit_emit_kernel(const mp_dp_task *__restrict__ tasks, const mp_dp_result *__restrict__ res, const uint32_t *__restrict__ flag,
                                       const uint32_t *__restrict__ slot, uint32_t n, mp_single_hit *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || !flag[t]) return;
    mp_single_hit h;
    h.read_id = tasks[t].read_id; h.strand = tasks[t].strand; h.algnmt = tasks[t].win_start + res[t].hit_loc; h.score = res[t].score;
    h.count = res[t].max_score_count; h.win_start = tasks[t].win_start; h.win_len = tasks[t].win_len; h.reserved = 0;
    out[slot[t]] = h;
}

__global__ void rescue_hit_flag_kernel(const mp_dp_task *__restrict__ rtasks, const mp_dp_result *__restrict__ rres, const uint32_t *__restrict__ rsrc,
                                       const uint32_t *__restrict__ lens, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n) return;
    flag[q] = (q < n && rsrc[q] != 0xFFFFFFFFu && rres[q].score > 0 && rres[q].score >= mp_dp_cutoff((int)lens[rtasks[q].read_id])) ? 1u : 0u;
}

__global__ void rescue_hit_emit_kernel(const mp_dp_task *__restrict__ stasks, const mp_dp_result *__restrict__ sres, const mp_dp_task *__restrict__ rtasks,
                                       const mp_dp_result *__restrict__ rres, const uint32_t *__restrict__ rsrc, const uint32_t *__restrict__ flag,
                                       const uint32_t *__restrict__ slot, uint32_t n, mp_rescue_hit *__restrict__ out)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || !flag[q]) return;
    const uint32_t src = rsrc[q];
    mp_rescue_hit h;
    h.aligned_read = stasks[src].read_id; h.pair_id = h.aligned_read >> 1;
    h.aligned_pos = stasks[src].win_start + sres[src].hit_loc; h.aligned_score = sres[src].score; h.aligned_strand = stasks[src].strand;
    h.dp_pos = rtasks[q].win_start + rres[q].hit_loc; h.dp_score = rres[q].score; h.dp_strand = rtasks[q].strand;
    h.aligned_win_start = stasks[src].win_start; h.aligned_win_len = stasks[src].win_len; h.aligned_count = sres[src].max_score_count;
    h.dp_win_start = rtasks[q].win_start; h.dp_win_len = rtasks[q].win_len; h.dp_count = rres[q].max_score_count;
    out[slot[q]] = h;
}
