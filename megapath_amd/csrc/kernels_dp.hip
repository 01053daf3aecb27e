// megapath_amd/csrc/kernels_dp.hip -- product code: semi-global affine-gap DP + traceback on gfx950.
//
// Replaces callDP -> SemiGlobalAlignment -> GenerateDPTable -> GPUBacktrack (reference soap4/CPU_DP.cpp:122-978) and
// the window/read gathers packRead / repackDNA (DV-DPfunctions.cpp:3009-3073).  Integer score DP: no MFMA.
//
// dp_fill_kernel<LPI, C>: LPI lanes (a half or whole wavefront) sweep one instance along anti-diagonals.  Lane k owns
//   read columns k*C+1 .. k*C+C; at step t it computes row t-k of those columns, receiving H/F of the column to its left
//   and the window base of that row from lane k-1 with one wavefront shuffle.  Plain Gotoh recurrences on ints
//   (SURVEY App. C.4b -- proven identical to the reference's u8 delta scheme); 5 traceback bits per cell go to a
//   scratch slab written 4*LPI contiguous bytes per step (coalesced).  The window is staged once into LDS.
// dp_trace_kernel: one lane per instance replays GPUBacktrack's state machine (CPU_DP.cpp:622-786) from those bits.
#include <hip/hip_runtime.h>
#include "mp_internal.h"

#define MP_DP_MISMATCH (-2)
#define MP_DP_OPEN (-3)
#define MP_DP_EXT (-1)
#define MP_DP_NEG (-100000)
#define MP_DP_MAX_WIN 1024          // bases of window staged in LDS per instance
#define MP_DP_BLOCK 256

// traceback bits per cell
#define TB_Z 1u     // soft-clipped: column 0, row 0 inside the left clip, or a cell raised to 0 inside the left clip
#define TB_D 2u     // H == H[i-1][j-1] + s  (diagonal consistent)
#define TB_E 4u     // !Z && E == H           (deletion flag, CPU_DP.cpp:183 value 1)
#define TB_EO 8u    // E opens from H[i-1][j] (ties: opening wins)
#define TB_FO 16u   // F opens from H[i][j-1]

__host__ __device__ inline int mp_dp_cutoff(int len)
{
    double v = 0.2 * (double)len;     // DP_SCORE_THRESHOLD_RATIO / LOWER_BOUND, definitions.h:166-167
    if (v < 30.0) v = 30.0;
    return (int)v;
}

__device__ __forceinline__ uint32_t mp_text_base(const uint32_t *text, uint64_t p)
{
    return (text[p >> 4] >> (2 * (p & 15))) & 3u;
}

template <int LPI, int C, typename FlagT>
__global__ void __launch_bounds__(MP_DP_BLOCK)
dp_fill_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t W,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, int clip_left, int clip_right,
               FlagT *__restrict__ flags, uint32_t flag_rows /* rows of LPI words reserved per instance */,
               mp_fill_result *__restrict__ out)
{
    constexpr int GROUPS = MP_DP_BLOCK / LPI;
    __shared__ uint32_t s_win[GROUPS][MP_DP_MAX_WIN / 16 + 1];
    const int g = threadIdx.x / LPI;              // instance slot inside the block
    const int k = threadIdx.x % LPI;              // lane inside the instance
    const uint32_t inst = blockIdx.x * GROUPS + g;
    const bool live = inst < n_tasks;
    mp_dp_task task; task.read_id = 0; task.strand = 1; task.win_start = 0; task.win_len = 0; task.reserved = 0;
    if (live) task = tasks[inst];
    const int L = live ? (int)lens[task.read_id] : 0;
    int m = (int)task.win_len;
    if (m > MP_DP_MAX_WIN) m = MP_DP_MAX_WIN;
    if (L > LPI * C) m = 0;                        // host guarantees the template covers the longest read
    const int clipL = task.strand == 1 ? clip_left : clip_right;     // soft clips follow the strand (:2908-2911)
    const int clipR = task.strand == 1 ? clip_right : clip_left;
    const int cutoff = mp_dp_cutoff(L);

    // ---- stage the window into LDS: word q holds bases 16q..16q+15 of the window, LSB-first
    for (int q = k; q * 16 < m; q += LPI) {
        uint64_t p = task.win_start + 16ull * (uint64_t)q;
        uint32_t sh = 2 * (uint32_t)(p & 15);
        uint64_t two = (uint64_t)text[p >> 4];
        if (sh) two |= (uint64_t)text[(p >> 4) + 1] << 32;
        s_win[g][q] = (uint32_t)(two >> sh);
    }
    // ---- this lane's read bases (reverse-complemented for strand 2) and row-0 boundary
    int rb[C], Hup[C], Eup[C];
    const uint32_t *rd = reads + (uint64_t)task.read_id * W;
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
        int j = k * C + cc + 1;
        int b = 4;                                 // never matches
        if (j <= L) {
            int p = task.strand == 1 ? j - 1 : L - j;
            b = (int)((rd[p >> 4] >> (2 * (p & 15))) & 3u);
            if (task.strand != 1) b = 3 - b;
        }
        rb[cc] = b;
        Hup[cc] = j <= clipL ? 0 : MP_DP_OPEN + (j - clipL - 1) * MP_DP_EXT;   // CPU_DP.cpp:394-432
        Eup[cc] = MP_DP_NEG;
    }
    int diag_carry;                                // H[i-1][j0-1]
    {
        int j0m1 = k * C;
        diag_carry = j0m1 <= clipL ? 0 : MP_DP_OPEN + (j0m1 - clipL - 1) * MP_DP_EXT;
    }
    __syncthreads();

    int min_col = L - clipR; if (min_col < 0) min_col = 0;
    int best = 0, best_row = 0, best_col = 0, cnt = 0;
    int outH = 0, outF = MP_DP_NEG, outB = 0;      // what this lane hands to lane k+1
    FlagT *frow = flags + (uint64_t)inst * flag_rows * LPI;

    // the wavefront may hold two instances with different window lengths: run to the longest
    int steps = m > 0 ? m + LPI - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(steps, o); steps = v > steps ? v : steps; }

    for (int t = 1; t <= steps; t++) {
        int inH = __shfl_up(outH, 1, LPI);
        int inF = __shfl_up(outF, 1, LPI);
        int inB = __shfl_up(outB, 1, LPI);
        const int i = t - k;                       // row handled now
        const bool row_ok = i >= 1 && i <= m;
        int base;
        if (k == 0) {
            inH = 0; inF = MP_DP_NEG;              // column 0: H = 0 on every row, no insertion state
            base = row_ok ? (int)((s_win[g][(i - 1) >> 4] >> (2 * ((i - 1) & 15))) & 3u) : 0;
        } else base = inB;
        if (row_ok) {
            int Hl = inH, Fl = inF, diag = diag_carry;
            FlagT fw = 0;
#pragma unroll
            for (int cc = 0; cc < C; cc++) {
                const int j = k * C + cc + 1;
                const int eo = Hup[cc] + MP_DP_OPEN, ee = Eup[cc] + MP_DP_EXT;
                const int e = eo >= ee ? eo : ee;
                const int fo = Hl + MP_DP_OPEN, fe = Fl + MP_DP_EXT;
                const int f = fo >= fe ? fo : fe;
                const int dg = diag + (base == rb[cc] ? 1 : MP_DP_MISMATCH);
                int h = dg > e ? dg : e;
                h = h > f ? h : f;
                uint32_t bits = 0;
                if (j <= clipL && h < 0) { h = 0; bits = TB_Z; }             // CPU_DP.cpp:505-510
                else if (e == h) bits = TB_E;
                if (h == dg) bits |= TB_D;
                if (eo >= ee) bits |= TB_EO;
                if (fo >= fe) bits |= TB_FO;
                fw |= (FlagT)bits << (5 * cc);
                diag = Hup[cc];
                Hup[cc] = h; Eup[cc] = e; Hl = h; Fl = f;
                // best cell: rows <= refLen, columns in [L - clipR, L], strict '>' in row-major order (CPU_DP.cpp:538-591)
                if (j >= min_col && j <= L && h >= cutoff) {
                    if (h > best) { best = h; best_row = i; best_col = j; cnt = 1; }
                    else if (h == best) cnt++;
                }
            }
            diag_carry = inH;
            outH = Hl; outF = Fl;
            frow[(uint64_t)(t - 1) * LPI + k] = fw;
        }
        outB = base;
    }

    // ---- reduce over the LPI lanes: max score, then smallest row, then smallest column; ties counted
    unsigned long long key = best > 0 ? ((unsigned long long)(uint32_t)best << 32) | ((uint32_t)(0xFFFF - best_row) << 16) | (uint32_t)(0xFFFF - best_col) : 0ull;
    unsigned long long kmax = key;
    for (int o = LPI / 2; o > 0; o >>= 1) { unsigned long long v = __shfl_xor(kmax, o, LPI); kmax = v > kmax ? v : kmax; }
    const int gbest = (int)(kmax >> 32);
    int c2 = (best == gbest && gbest > 0) ? cnt : 0;
    for (int o = LPI / 2; o > 0; o >>= 1) c2 += __shfl_xor(c2, o, LPI);
    if (live && k == 0) {
        mp_fill_result r;
        r.score = gbest;
        r.row = (uint16_t)(0xFFFF - ((kmax >> 16) & 0xFFFF));
        r.col = (uint16_t)(0xFFFF - (kmax & 0xFFFF));
        r.count = c2 > 255 ? 255u : (uint32_t)c2;
        if (gbest == 0) { r.row = 0; r.col = 0; r.count = 0; }
        out[inst] = r;
    }
}

// ------------------------------------------------------------------------------------------------ traceback
template <int LPI, int C, typename FlagT>
__global__ void __launch_bounds__(64)
dp_trace_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ lens, uint32_t n_tasks, int clip_left, int clip_right,
                const FlagT *__restrict__ flags, uint32_t flag_rows, const mp_fill_result *__restrict__ fill,
                mp_dp_result *__restrict__ out, uint8_t *__restrict__ ops, uint32_t ops_stride)
{
    const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= n_tasks) return;
    const mp_dp_task task = tasks[inst];
    const mp_fill_result fr = fill[inst];
    const int L = (int)lens[task.read_id];
    const int clipL = task.strand == 1 ? clip_left : clip_right;
    const int cutoff = mp_dp_cutoff(L);
    mp_dp_result res; res.score = 0; res.hit_loc = 0; res.max_score_count = 0; res.n_ops = 0;
    if (fr.score < cutoff || fr.score <= 0 || task.win_len == 0) { out[inst] = res; return; }
    res.max_score_count = fr.count;
    const FlagT *frow = flags + (uint64_t)inst * flag_rows * LPI;
    uint8_t *op = ops ? ops + (uint64_t)inst * ops_stride : nullptr;
    uint32_t n = 0;
    auto emit = [&](uint8_t ch, int rep) { for (int q = 0; q < rep; q++) { if (op && n < ops_stride) op[n] = ch; n++; } };
    auto cell = [&](int r, int c) -> uint32_t {      // traceback bits of (row r, column c)
        if (c == 0) return TB_Z;
        if (r == 0) return c <= clipL ? TB_Z : 0u;
        int lane = (c - 1) / C, cc = (c - 1) % C;
        FlagT w = frow[(uint64_t)(r + lane - 1) * LPI + lane];
        return (uint32_t)(w >> (5 * cc)) & 31u;
    };
    bool valid = true;
    int hit = 0;
    if (L < cutoff) valid = false;                                    // CPU_DP.cpp:830-831
    else {
        int c = fr.col, r = fr.row;
        emit('S', L - c);
        enum { NORMAL, I_EXT, D_EXT } state = NORMAL;
        bool ended = false, sm_exit = false;
        while (c > 0 && r > 0) {
            const uint32_t b = cell(r, c);
            if (state == NORMAL) {
                if ((b & TB_D) && c != 1 && (cell(r - 1, c - 1) & TB_Z)) { sm_exit = true; break; }   // DP_BT_SM_EXIT
                else if (b & TB_D) { emit('X', 1); r--; c--; }
                else if (b & TB_E) { emit('D', 1); r--; if (!(b & TB_EO)) state = D_EXT; }
                else { emit('I', 1); c--; if (!(b & TB_FO)) state = I_EXT; }
            } else if (state == D_EXT) {
                if ((b & TB_EO) && (cell(r - 1, c) & TB_Z)) { valid = false; ended = true; break; }    // DP_BT_SD_EXIT
                emit('D', 1); r--;
                if (b & TB_EO) state = NORMAL;
            } else {
                if ((b & TB_FO) && (cell(r, c - 1) & TB_Z)) {                                          // DP_BT_SI_EXIT
                    emit('I', 1); emit('S', c - 1); hit = r; ended = true; break;
                }
                emit('I', 1); c--;
                if (b & TB_FO) state = NORMAL;
            }
        }
        if (sm_exit) {
            // one diagonal op ('X' is resolved to M/m by dp_ops_fix_kernel), then clip the remaining c-1 read bases
            emit('X', 1); emit('S', c - 1); hit = r - 1;
        } else if (!ended) {
            if (r == 0) {
                int sc = clipL > c ? c : clipL;                       // CPU_DP.cpp:743-757
                emit('I', c - sc); emit('S', sc);
            }
            hit = r;
        }
    }
    if (valid) { res.score = fr.score; res.hit_loc = (uint32_t)hit; res.n_ops = n; }
    out[inst] = res;
}

// 'X' marks a diagonal step; this pass turns it into 'M' / 'm' by comparing the bases (kept out of the traceback
// loop so that the loop touches nothing but the flag slab).  Walks the ops in traceback order (end -> start).
__global__ void dp_ops_fix_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens,
                                  uint32_t W, const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, const mp_fill_result *__restrict__ fill,
                                  const mp_dp_result *__restrict__ res, uint8_t *__restrict__ ops, uint32_t ops_stride)
{
    const uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= n_tasks) return;
    const mp_dp_result rr = res[inst];
    if (rr.score <= 0 || rr.n_ops == 0 || rr.n_ops > ops_stride) return;
    const mp_dp_task task = tasks[inst];
    const int L = (int)lens[task.read_id];
    const uint32_t *rd = reads + (uint64_t)task.read_id * W;
    uint8_t *op = ops + (uint64_t)inst * ops_stride;
    int r = fill[inst].row, c = L;                 // ops start with the right soft clip, so the read cursor starts at L
    for (uint32_t q = 0; q < rr.n_ops; q++) {
        uint8_t o = op[q];
        if (o == 'S' || o == 'I') c--;
        else if (o == 'D') r--;
        else if (o == 'X') {
            int p = task.strand == 1 ? c - 1 : L - c;
            int b = (int)((rd[p >> 4] >> (2 * (p & 15))) & 3u);
            if (task.strand != 1) b = 3 - b;
            int t = (int)mp_text_base(text, task.win_start + (uint64_t)(r - 1));
            op[q] = b == t ? 'M' : 'm';
            r--; c--;
        }
    }
}

// ------------------------------------------------------------------------------------------------ task builders (K5)
__device__ __forceinline__ void mp_dp_window(uint64_t seed_pos, int read_len, uint64_t text_len, uint64_t *start, uint32_t *wlen)
{
    int margin = read_len > 100 ? 30 : 25;                 // DP2_MARGIN, DV-DPfunctions.cpp:1760
    uint64_t s = seed_pos - (uint64_t)margin;
    if (s >= text_len) s = 0;                               // :2871-2874
    uint64_t len = (uint64_t)read_len + 2ull * (uint64_t)margin;
    if (s + len > text_len) len = text_len - s;
    *start = s; *wlen = (uint32_t)len;
}

// packLeft (DV-DPfunctions.cpp:2857-2924)
__global__ void build_left_tasks_kernel(const mp_candidate *__restrict__ cands, uint32_t n, const uint32_t *__restrict__ lens,
                                        uint64_t text_len, mp_dp_task *__restrict__ tasks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_candidate c = cands[t];
    mp_dp_task k; k.read_id = c.read_id_left; k.strand = 1; k.reserved = 0;
    mp_dp_window(c.pos0, (int)lens[c.read_id_left], text_len, &k.win_start, &k.win_len);
    tasks[t] = k;
}

// packRight (DV-DPfunctions.cpp:2926-3005): only for candidates whose left leg reached its cutoff
__global__ void build_right_tasks_kernel(const mp_candidate *__restrict__ cands, uint32_t n, const uint32_t *__restrict__ lens,
                                         uint64_t text_len, int insert_high, const mp_dp_task *__restrict__ left_tasks,
                                         const mp_dp_result *__restrict__ left_res, mp_dp_task *__restrict__ tasks)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_candidate c = cands[t];
    const uint32_t right = c.read_id_left ^ 1u;
    mp_dp_task k; k.read_id = right; k.strand = 2; k.reserved = 0; k.win_start = 0; k.win_len = 0;
    const mp_dp_result lr = left_res[t];
    if (lr.score >= mp_dp_cutoff((int)lens[c.read_id_left]) && lr.score > 0) {
        mp_dp_window(c.pos1, (int)lens[right], text_len, &k.win_start, &k.win_len);
        uint64_t hit_left = left_tasks[t].win_start + lr.hit_loc;
        uint64_t bounded = hit_left + (uint64_t)(int64_t)insert_high - k.win_start;     // restrict maximum insert size
        if (bounded < (uint64_t)k.win_len) k.win_len = (uint32_t)bounded;
    }
    tasks[t] = k;
}

// DP2CPUAlgnThread result assembly (DV-DPfunctions.cpp:3432-3530), positions and scores only
__global__ void collect_pair_hits_kernel(const mp_candidate *__restrict__ cands, uint32_t n, const uint32_t *__restrict__ lens,
                                         const mp_dp_task *__restrict__ lt, const mp_dp_result *__restrict__ lr,
                                         const mp_dp_task *__restrict__ rt, const mp_dp_result *__restrict__ rr,
                                         mp_pair_hit *__restrict__ hits, unsigned long long *n_hits, uint32_t cap)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_candidate c = cands[t];
    const uint32_t left = c.read_id_left, right = left ^ 1u;
    const mp_dp_result a = lr[t], b = rr[t];
    if (a.score <= 0 || b.score <= 0) return;
    if (a.score < mp_dp_cutoff((int)lens[left]) || b.score < mp_dp_cutoff((int)lens[right])) return;
    uint32_t slot = (uint32_t)atomicAdd(n_hits, 1ull);
    if (slot >= cap) return;
    const int read_side = (int)(left & 1u);          // 0: the read (even id) is the left leg
    mp_pair_hit h;
    h.pair_id = left >> 1;
    h.left_is_mate = (uint32_t)read_side;
    const uint64_t pos_l = lt[t].win_start + a.hit_loc, pos_r = rt[t].win_start + b.hit_loc;
    if (read_side == 0) {
        h.algnmt_1 = pos_l; h.algnmt_2 = pos_r; h.score_1 = a.score; h.score_2 = b.score;
        h.start_1 = (uint32_t)lt[t].win_start; h.start_2 = (uint32_t)rt[t].win_start;
        h.dplen_1 = (uint16_t)lt[t].win_len; h.dplen_2 = (uint16_t)rt[t].win_len;
        h.count_1 = (uint16_t)a.max_score_count; h.count_2 = (uint16_t)b.max_score_count;
    } else {
        h.algnmt_1 = pos_r; h.algnmt_2 = pos_l; h.score_1 = b.score; h.score_2 = a.score;
        h.start_1 = (uint32_t)rt[t].win_start; h.start_2 = (uint32_t)lt[t].win_start;
        h.dplen_1 = (uint16_t)rt[t].win_len; h.dplen_2 = (uint16_t)lt[t].win_len;
        h.count_1 = (uint16_t)b.max_score_count; h.count_2 = (uint16_t)a.max_score_count;
    }
    hits[slot] = h;
}
