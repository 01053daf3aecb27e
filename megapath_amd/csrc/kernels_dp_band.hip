// megapath_amd/csrc/kernels_dp_band.hip -- banded variant of the semi-global DP fill (included by api.hip after kernels_dp.hip).
//
// Same recurrences, same traceback bits and the same answer as dp_fill_kernel (reference soap4/CPU_DP.cpp:122-978), for
// the deep-DP windows only (window = read + 2 x DP2_MARGIN, DV-DPfunctions.cpp:1760): there the alignment the reference
// reports almost always stays near the window's main diagonals, and whether it does can be PROVED from the banded result
// itself, so the band is an exact fast path, not a heuristic:
//
//   Let delta = row - column.  A path that touches a cell with delta < -GB collects at most L - GB - 1 matches (the
//   columns left of the cell can pair with at most `row` window bases), one with delta > (m - L) + GB at most L - GB - 1
//   as well (the rows below the cell are too few).  The band holds every cell with -GB <= delta <= W - 1 - GB, cells
//   outside it are treated as minus infinity, so every banded value is a lower bound of the true one and exact whenever
//   the best path to that cell stays inside.  If the banded best score S satisfies S >= L - GB then no path leaving the
//   band can tie or beat it: the true best cell (first maximum by row, then column), the number of cells sharing its score,
//   every value on its optimal path and every tie the traceback bits record are identical to the full table's.
//   Instances that do not reach that score (or whose shape does not fit the band) are marked undecided and re-run by the
//   full-table kernels.
//
// Mapping: ONE LANE per instance pair (the two instances in the halves of packed 16-bit registers, as in dp_fill_kernel),
// sweeping the read columns; the band's W cells of a column live in registers (H and F, in place), the vertical gap
// state E is a running value.  No cross-lane traffic at all, 15 (scores) to 32 (scores + traceback bits) VALU
// instructions per cell pair.  Window base codes are staged in LDS, one byte per row and lane holding both instances'
// 4-bit codes ([row][lane]: a wavefront reads 64 consecutive bytes), so that two waves per SIMD stay resident.
//
// How run_dp_band (api.hip) uses the kernels, every step exact (DESIGN.md section 4 has the arguments):
//   dp_prefilter_kernel     exact matches of read blocks in the window confine every qualifying alignment to a few
//                           diagonals (or show that there is none); perfect matches are finished without a table;
//   dp_band_kernel MODE 3/1 first pass: best cell and ties, over 27 diagonals around the matching ones (MODE 3) or the
//                           whole band (MODE 1: repeats inside the window, reads too short for the block test);
//   dp_band_classify_kernel by the score: how far the alignment can stray from the best cell's diagonal;
//   dp_band_kernel MODE 2   second pass: traceback bits over 9 / 27 / 51 diagonals around the best cell, then
//                           dp_trace_kernel;
//   second tier             the same once more, with a budget of 25 instead of 13, for what is still undecided;
//   MODE 0                  single pass over the whole band with bits, when a soft-clip limit is too small for the
//                           centred bands (their rows <= 0 must lie under the left clip).
#pragma once

#define MP_BAND_MAXL 320            // longest read the band kernel is instantiated for (a second instance for <= 160 needs less LDS)
#define MP_BAND_DMAX 60             // 2 x DP2_MARGIN: widest window surplus of a deep-DP instance
#define MP_BAND_GB 13               // guard diagonals either side; qualifies scores >= L - 13

// cells of the band in one column, words of traceback bits per column (3 cells per 16-bit half)
#define MP_BAND_W (MP_BAND_DMAX + 2 * MP_BAND_GB + 1)
#define MP_BAND_NW ((MP_BAND_W + 2) / 3)
// second pass of the two-pass scheme: diagonals of the best cell +- MP_BAND_GB
#define MP_BAND_W2 (2 * MP_BAND_GB + 1)
#define MP_BAND_NW2 ((MP_BAND_W2 + 2) / 3)
// second tier: instances the first tier left undecided, alignments that lose up to 25
#define MP_BAND_GB3 25
#define MP_BAND_W3 (2 * MP_BAND_GB3 + 1)

// One band cell of two instances.  Hd enters as H of (row-1, col-1) [same d, previous column] and leaves as H of this
// cell; Hn / Fn are the previous column's H / F one row down (d + 1): the horizontal neighbour; Hc / Ec the cell above
// in this column.  Frame arithmetic as in MP_DP_CELL_CORE (values are x + BIAS - column).
#define MP_BAND_CORE \
        "v_sub_u32 %[eo], %[Hc], %[k3]\n\t" \
        "v_sub_u32 %[Ec], %[Ec], %[k1]\n\t" \
        "v_pk_max_u16 %[Ec], %[eo], %[Ec]\n\t"                    /* E */ \
        "v_sub_u32 %[fo], %[Hn], %[k4]\n\t" \
        "v_sub_u32 %[Fd], %[Fn], %[k2]\n\t" \
        "v_pk_max_u16 %[Fd], %[fo], %[Fd]\n\t"                    /* F */ \
        "v_xor_b32 %[t1], %[base], %[rb]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k3]\n\t" \
        "v_sub_u32 %[dg], %[Hd], %[t1]\n\t" \
        "v_pk_max_u16 %[t1], %[dg], %[Ec]\n\t" \
        "v_pk_max_u16 %[t1], %[t1], %[Fd]\n\t" \
        "v_pk_max_u16 %[Hd], %[t1], %[floorv]\n\t"                /* H */
#define MP_BAND_BITS \
        "v_xor_b32 %[bits], %[Hd], %[t1]\n\t" \
        "v_pk_min_u16 %[bits], %[bits], %[k1]\n\t" \
        "v_pk_mad_u16 %[fw], %[fw], %[k2], %[bits]\n\t"           /* Z: the word is built by shifting the bits in */ \
        "v_xor_b32 %[t1], %[Hd], %[dg]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_u16 %[fw], %[fw], %[k2], %[t1]\n\t"             /* nD */ \
        "v_xor_b32 %[t1], %[Hd], %[Ec]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_u16 %[fw], %[fw], %[k2], %[t1]\n\t"             /* nE */ \
        "v_xor_b32 %[t1], %[Ec], %[eo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_u16 %[fw], %[fw], %[k2], %[t1]\n\t"            /* nEO */ \
        "v_xor_b32 %[t1], %[Fd], %[fo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_u16 %[fw], %[fw], %[k2], %[t1]\n\t"            /* nFO */
// the column's maximum; its row is looked for only when it matters
#define MP_BAND_COLMAX "v_pk_max_u16 %[colmax], %[colmax], %[Hd]"
#define MP_BAND_NOP ""

MP_PKOP2(pk_maxu, "v_pk_max_u16")
__device__ __forceinline__ v2s pk_subsat(v2s a, v2s b) { v2s r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Word w (bases 16w .. 16w+15) of a read in alignment orientation, as dp_prefilter_kernel stages it.
__device__ __forceinline__ uint32_t mp_oriented_word(const uint32_t *__restrict__ rd, int nw, int L, uint32_t strand, int w)
{
    if (w >= nw) return 0u;
    uint32_t v;
    if (strand == 1) v = rd[w];
    else {
        // reverse strand: reverse the 2-bit groups of every word, take the words in reverse order, complement, and shift out the
        // 16 nw - L pad bases that the reversal brought to the front
        auto z = [&](int q) -> uint32_t {
            if (q >= nw) return 0u;
            const uint32_t x = __brev(rd[nw - 1 - q]);
            return ~(((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u));
        };
        v = (uint32_t)((((uint64_t)z(w + 1) << 32) | z(w)) >> (2u * (uint32_t)(16 * nw - L)));
    }
    if (w == nw - 1 && (L & 15)) v &= (1u << (2 * (L & 15))) - 1u;
    return v;
}

// MODE 0: one pass -- band of W = MP_BAND_W diagonals fixed to the window, traceback bits and best cell.
// MODE 1: the same band, scores only (best cell, ties): first pass of the two-pass scheme.
// MODE 3: as MODE 1, over the W = 2 GB + 1 diagonals around a centre found by dp_prefilter_kernel (instances for which it
//         proved that every alignment of qualifying score lies within them).
// MODE 2: second pass of the two-pass scheme, traceback bits only, over the W = 2 GB + 1 diagonals around the diagonal of
//         the best cell found by the first pass (`center`).  An instance the first pass qualified scored >= L - 13: its
//         gaps cost at most 13 in total, i.e. at most 11 gap bases, so every optimal path into the best cell -- and every
//         tied alternative the bits have to record -- stays within 11 diagonals of it.  Rows <= 0 can now appear up to
//         column 2 GB + 1, which must lie under the left clip (run_dp_band checks the limit).
template <int W, int GB, int MAXL, int MODE>
__global__ void __launch_bounds__(64)
dp_band_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, int clip_left, int clip_right,
               uint32_t *__restrict__ flags, uint32_t ncols /* columns of NW x 64 words reserved per block */,
               const mp_fill_result *__restrict__ center, mp_fill_result *__restrict__ out, const uint32_t *__restrict__ list = nullptr)
{
    constexpr int NW = (W + 2) / 3;
    static_assert(W % 3 == 0, "three cells per 16-bit half of a flag word");
    constexpr int R = MAXL + W + 1;        // LDS rows: row i of the window sits at index i + GB
    constexpr int B = MP_DP_BIAS;
    __shared__ uint8_t s_code[R][64];                 // code of instance 0 | code of instance 1 << 4
    __shared__ uint32_t s_rd[MAXL / 16][2][64];       // both reads in alignment orientation, 16 bases per word: a column's base is a shift
    const int lane = threadIdx.x;
    const uint32_t pair = blockIdx.x * 64 + lane;
    // per-instance state, packed to keep the register budget for the band: L | m << 9 | forward << 18 | fit << 19 |
    // empty << 20 | bad << 21
    uint32_t info[2], rid[2];
    int rbase[2];                                   // row of band cell d in column j: j + d + rbase
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t slot = 2 * pair + q;                 // position in the launch; with a list, slot s stands for instance list[s]
        mp_dp_task task; task.read_id = 0; task.strand = 1; task.win_start = 0; task.win_len = 0; task.reserved = 0;
        const bool live = slot < n_tasks;
        const uint32_t inst = live && list ? list[slot] : slot;
        if (live) task = tasks[inst];
        int L = live ? (int)lens[task.read_id] : 0;
        int m = (int)task.win_len;
        rid[q] = task.read_id;
        const int clipL = task.strand == 1 ? clip_left : clip_right;
        // the band covers this instance: read short enough, window surplus inside the band, rows <= 0 only under the left clip
        bool fit;
        rbase[q] = -GB;
        if (MODE == 2) {
            const mp_fill_result ctr = live ? center[inst] : mp_fill_result{0, 0, 0, 0};
            fit = live && ctr.score > 0 && L <= MAXL && m >= 1 && clipL >= 2 * GB + 1;
            rbase[q] = (int)ctr.row - (int)ctr.col - GB;
        } else if (MODE == 3) {
            // first pass over 2 GB + 1 diagonals around the centre dp_prefilter_kernel found (row - col of `center`)
            const mp_fill_result ctr = live ? center[inst] : mp_fill_result{0, 0, 0, 0};
            fit = live && L >= 30 && L <= MAXL && m >= 1 && clipL >= 2 * GB + 1;
            rbase[q] = (int)ctr.row - (int)ctr.col - GB;
        } else
            fit = live && L >= 30 && L <= MAXL && m >= 1 && m - L <= W - 1 - 2 * GB && m + GB < R && clipL >= GB;
        const bool empty = live && task.win_len == 0;  // nothing to align: decided, score 0 (as the full kernels report it)
        if (!fit) { L = 0; m = 0; }
        info[q] = (uint32_t)L | ((uint32_t)m << 9) | (task.strand == 1 ? 1u << 18 : 0u) | (fit ? 1u << 19 : 0u) | (empty ? 1u << 20 : 0u);
        if (q == 0)
            for (int r = 0; r < R; r++) s_code[r][lane] = (uint8_t)(12 | (15 << 4));      // pads: match nothing
        {
            const uint32_t *rd = reads + (uint64_t)task.read_id * Wr;
            const int nw = (L + 15) / 16;
            for (int w = 0; w < MAXL / 16; w++) s_rd[w][q][lane] = w < nw ? mp_oriented_word(rd, nw, L, task.strand, w) : 0u;
        }
        for (int w = 0; w * 16 < m; w++) {
            uint64_t p = task.win_start + 16ull * (uint64_t)w;
            uint32_t sh = 2 * (uint32_t)(p & 15);
            uint64_t two = (uint64_t)text[p >> 4];
            if (sh) two |= (uint64_t)text[(p >> 4) + 1] << 32;
            uint32_t bases = (uint32_t)(two >> sh);
            for (int x = 0; x < 16 && w * 16 + x < m; x++) {
                const int at = w * 16 + x + 1 - rbase[q];               // window row i sits at index i - rbase
                const uint32_t code = (uint32_t)MP_DP_CODE(bases & 3u);
                bases >>= 2;
                if (at < 0 || at >= R) continue;
                uint8_t &cell = s_code[at][lane];
                cell = q == 0 ? (uint8_t)((cell & 0xF0u) | code) : (uint8_t)((cell & 0x0Fu) | (code << 4));
            }
        }
    }
    int Lw = (int)max(info[0] & 511u, info[1] & 511u);
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(Lw, o); Lw = v > Lw ? v : Lw; }
    __syncthreads();

    dp_consts K;
    K.k1 = pk1(1); K.k2 = pk1(2); K.k3 = pk1(3); K.k4 = pk1(4); K.k16 = pk1(16);
    v2s H[W], F[W];
#pragma unroll
    for (int d = 0; d < W; d++) { H[d] = pk1((short)B); F[d] = pk1(MP_DP_SENT); }     // column 0: H = 0, no insertion state
    const v2s sent = pk1(MP_DP_SENT);
    // best so far: key = excess score << 16 | 0xFFFF - row (larger score, then smaller row; columns ascend), and
    // column | ties << 9
    uint32_t key[2] = {0, 0}, colties[2] = {0, 0};
    uint32_t rw[2] = {0, 0};                                  // 16 read bases of each instance
    uint32_t *fcol = flags + (uint64_t)blockIdx.x * ncols * (NW * 64) + lane;

    for (int j = 1; j <= Lw; j++) {
        short b2[2], fl2[2], th2[2], el2[2];
        if (((j - 1) & 15) == 0) { rw[0] = s_rd[(j - 1) >> 4][0][lane]; rw[1] = s_rd[(j - 1) >> 4][1][lane]; }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            // read base of column j in alignment orientation (packRead, DV-DPfunctions.cpp:3009-3040)
            const int L = (int)(info[q] & 511u);
            const bool fwd = (info[q] >> 18) & 1u;
            const int clipL = fwd ? clip_left : clip_right, clipR = fwd ? clip_right : clip_left;
            int b = 9 + q;
            if (j <= L) b = MP_DP_CODE((rw[q] >> (2 * ((j - 1) & 15))) & 3u);     // (rw: this column's word of the oriented read, below)
            b2[q] = (short)b;
            fl2[q] = (short)(j <= clipL ? B - j : 0);
            th2[q] = (short)(L - GB - 1 + B - j);             // score L - GB - 1 in this column's frame
            int mc = L - clipR; if (mc < 0) mc = 0;
            el2[q] = (short)((j >= mc && j <= L) ? -1 : 0);
        }
        const v2s rb = pk((short)(b2[0] << 12), b2[1]), floorv = pk(fl2[0], fl2[1]), thr = pk(th2[0], th2[1]);
        v2s Ec = sent, colmax = pk1(0);
        v2s fw = pk1(0);
        const uint8_t *c0 = &s_code[j][lane];
        constexpr int PF = 4;                                                   // LDS bytes are fetched this many cells ahead
        uint32_t cbq[W + PF];
#pragma unroll
        for (int d = 0; d < PF; d++) cbq[d] = c0[d * 64];
#pragma unroll
        for (int d = 0; d < W; d++) {
            if (d + PF < W) cbq[d + PF] = c0[(d + PF) * 64];
            const uint32_t cb = cbq[d];                                         // row j + d - GB of both windows
            // one shift separates the two codes: instance 0's lands in bits 12-15 of the low half, instance 1's in bits 0-3 of
            // the high half (the read codes of the column are laid out the same way, so XOR / min_u16(.,3) still work)
            const v2s base = as_v2s((int)(cb << 12));
            v2s eo, fo, dg, t1, bits;
            const v2s Hc = d == 0 ? sent : H[d - 1];
            const v2s Hn = d + 1 < W ? H[d + 1] : sent;
            const v2s Fn = d + 1 < W ? F[d + 1] : sent;
            if (MODE == 1 || MODE == 3)
                asm(MP_BAND_CORE MP_BAND_COLMAX
                    : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [Hd] "+&v"(H[d]), [Ec] "+&v"(Ec), [Fd] "=&v"(F[d]), [colmax] "+&v"(colmax)
                    : [Hc] "v"(Hc), [Hn] "v"(Hn), [Fn] "v"(Fn), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv),
                      [k1] "v"(K.k1), [k2] "v"(K.k2), [k3] "v"(K.k3), [k4] "v"(K.k4));
            else if (MODE == 2)
                asm(MP_BAND_CORE MP_BAND_BITS MP_BAND_NOP
                    : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [Hd] "+&v"(H[d]), [Ec] "+&v"(Ec),
                      [Fd] "=&v"(F[d]), [fw] "+&v"(fw)
                    : [Hc] "v"(Hc), [Hn] "v"(Hn), [Fn] "v"(Fn), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv),
                      [k1] "v"(K.k1), [k2] "v"(K.k2), [k3] "v"(K.k3), [k4] "v"(K.k4));
            else
                asm(MP_BAND_CORE MP_BAND_BITS MP_BAND_COLMAX
                    : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [Hd] "+&v"(H[d]), [Ec] "+&v"(Ec),
                      [Fd] "=&v"(F[d]), [fw] "+&v"(fw), [colmax] "+&v"(colmax)
                    : [Hc] "v"(Hc), [Hn] "v"(Hn), [Fn] "v"(Fn), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv),
                      [k1] "v"(K.k1), [k2] "v"(K.k2), [k3] "v"(K.k3), [k4] "v"(K.k4));
            if ((MODE == 0 || MODE == 2) && (d % 3 == 2 || d == W - 1)) {
                fcol[(uint64_t)(j - 1) * (NW * 64) + (d / 3) * 64] = __builtin_bit_cast(uint32_t, fw);
                fw = pk1(0);
            }
        }
        // Does this column hold an eligible value at or above the qualifying score?  Only the last GB or so columns of a
        // read can; there the band is walked once more for the first row of the maximum and the number of its ties.
        const v2s ex = pk_subsat(colmax, thr) & pk(el2[0], el2[1]);
        if (MODE != 2 && __any(as_int(ex) != 0)) {
            short lim2[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {                      // rows past the window's end (d > m - j + GB) do not count
                int lim = 127 - ((int)((info[q] >> 9) & 511u) - j - rbase[q]);
                lim2[q] = (short)(lim < 0 ? 0 : (lim > 128 ? 128 : lim));
            }
            const v2s lim = pk(lim2[0], lim2[1]);
            v2s cntv = pk1(127), first = pk1(0), neq = pk1(0);
#pragma unroll
            for (int d = 0; d < W; d++) {
                // opaque packed ops: the optimiser would otherwise unpack this into per-half compares and selects
                const v2s b = pk_minu((H[d] ^ colmax) | pk_subsat(lim, cntv), K.k1);   // 0: equals the maximum, inside the window
                neq = as_v2s(as_int(neq) + as_int(b));
                first = pk_maxu(first, pk_sub(b, K.k1) & cntv);
                cntv = as_v2s(as_int(cntv) - as_int(K.k1));
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int u = (int)(unsigned short)(q == 0 ? ex.x : ex.y);
                if (u == 0) continue;
                const int ties = W - (int)(unsigned short)(q == 0 ? neq.x : neq.y);
                const int row = j + (127 - (int)(unsigned short)(q == 0 ? first.x : first.y)) + rbase[q];
                if (ties == 0) { info[q] |= 1u << 21; continue; }      // the column's maximum sits on a pad row: leave it to the full table
                const uint32_t k2 = ((uint32_t)u << 16) | (uint32_t)(0xFFFF - row);
                const uint32_t told = (key[q] >> 16) == (uint32_t)u ? colties[q] >> 9 : 0u;
                const uint32_t tnew = min(told + (uint32_t)ties, 0xFFFFu);
                if (k2 > key[q]) { key[q] = k2; colties[q] = (uint32_t)j | (tnew << 9); }
                else if ((key[q] >> 16) == (uint32_t)u) colties[q] = (colties[q] & 511u) | (tnew << 9);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t slot = 2 * pair + q;
        if (MODE == 2 || slot >= n_tasks) continue;
        const uint32_t inst = list ? list[slot] : slot;
        mp_fill_result r;
        // score -1: "not decided by the band" -- below the qualifying score, outside the band's shape, or the first
        // maximum fell on a pad row past the window
        const int L = (int)(info[q] & 511u), m = (int)((info[q] >> 9) & 511u);
        const int u = (int)(key[q] >> 16), row = 0xFFFF - (int)(key[q] & 0xFFFFu);
        const uint32_t ties = colties[q] >> 9;
        const bool ok = ((info[q] >> 19) & 1u) && !((info[q] >> 21) & 1u) && u >= 1 && row >= 1 && row <= m;
        r.score = ok ? u + L - GB - 1 : (((info[q] >> 20) & 1u) ? 0 : -1);
        r.row = ok ? (uint16_t)row : 0; r.col = ok ? (uint16_t)(colties[q] & 511u) : 0;
        r.count = ok ? (ties > 255u ? 255u : ties) : 0u;       // num_sameScore, capped as in dp_fill_kernel
        out[inst] = r;
    }
}

// Before any table is filled, one lane per instance looks at exact matches of read blocks in the window.
//
// An alignment that qualifies for the band (score S >= L - 13, kernels above) loses at most 13 against a perfect match.
// A mismatch costs 3 of that (a column without its +1, and -2), a gap at least 3, so it has at most 4 of them and its
// matches -- at least S of them -- come in at most 5 exact runs: one run has R = ceil((L - 13) / 5) bases or more, and a
// run of 2b - 1 bases contains a whole read block [kb, kb + b).  So with b = (R + 1) / 2 every qualifying alignment
// shows up as a block of the read that equals the window on the run's diagonal, and all its cells lie within 11
// diagonals of that one (its gaps have at most 11 bases in all).  The kernel tests every block against every diagonal
// -13 .. (m - L) + 13 (rows outside the window do not count) and classifies:
//   no block matches anywhere            -> no qualifying alignment exists: straight to the full-table kernels (score -1);
//   matching diagonals span <= 4         -> every qualifying alignment lies within 13 diagonals of their centre: the
//                                           first band pass runs over 27 diagonals instead of 87 (MODE 3); if the read
//                                           equals the window on one of those diagonals the instance is finished here
//                                           (score L, smallest exact row, ties = exact diagonals);
//   otherwise (repeats inside the window) -> the first pass over the whole band.
// Blocks shorter than MP_PRE_MINB bases would match by chance too often: such (short) reads take the whole band.
#define MP_BAND_RW (MP_BAND_MAXL / 16)
#define MP_BAND_TW ((MP_BAND_MAXL + MP_BAND_DMAX) / 16 + 2)
#define MP_PRE_MINB 8
#define MP_PRE_NONE 3u
#define MP_PRE_NARROW15 4u         // as MP_PRE_NARROW, and a diagonal with at most two mismatches exists: 15 diagonals decide it
#define MP_PRE_WIDE 2u
#define MP_PRE_NARROW 1u
#define MP_PRE_DONE 0u
// GBUD: what a qualifying alignment may lose against a perfect match (13: the first tier; 25: the second tier, run on the
// instances the first tier left undecided, with band kernels of 51 diagonals).  With a list, slot s stands for instance
// list[s]; cls is indexed by slot, fill by instance.
// MAXL: longest read of the batch the instance is compiled for (160 for the usual short-read batches: half the words to orient and stage)
template <int GBUD, int MINB, int MAXL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
dp_prefilter_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
                    const mp_dp_task *__restrict__ tasks, uint32_t n, mp_fill_result *__restrict__ fill, uint32_t *__restrict__ cls,
                    const uint32_t *__restrict__ list = nullptr, int one_mismatch = 2 /* test switch: 0 such instances take the first pass, 1 only one-mismatch instances are finished here */)
{
    constexpr int RW = MAXL / 16, TW = (MAXL + MP_BAND_DMAX) / 16 + 2;
    __shared__ uint32_t s_a[RW + 1][64];        // the read in alignment orientation, LSB-first words
    __shared__ uint32_t s_t[TW + 1][64];        // the window, word 0 starts at its first base
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const uint32_t t = list ? list[slot] : slot;
    const int lane = threadIdx.x;
    const mp_dp_task task = tasks[t];
    const int L = (int)lens[task.read_id], m = (int)task.win_len;
    mp_fill_result r; r.score = -1; r.row = 0; r.col = 0; r.count = 0;
    cls[slot] = MP_PRE_WIDE;
    if (L < 30 || L > MAXL || m < 1 || m - L > MP_BAND_DMAX) { fill[t] = r; return; }      // the first pass decides what to do with it
    constexpr int RUNS = GBUD / 3 + 1;               // at most GBUD / 3 mismatches and gaps
    const int R = (L - GBUD + RUNS - 1) / RUNS, b = min(16, (R + 1) / 2);
    if (b < MINB) { fill[t] = r; return; }
    const uint32_t *rd = reads + (uint64_t)task.read_id * Wr;
    const int nw = (L + 15) / 16;
    // ---- oriented read.  Reverse strand: reverse the 2-bit groups of every word, take the words in reverse order,
    // complement, and shift out the 16 nw - L pad bases that the reversal brought to the front.
    {
        uint32_t a[RW + 1];
#pragma unroll
        for (int w = 0; w <= RW; w++) a[w] = 0;
        if (task.strand == 1) {
#pragma unroll
            for (int w = 0; w < RW; w++) if (w < nw) a[w] = rd[w];
        } else {
            uint32_t z[RW + 1];
#pragma unroll
            for (int w = 0; w <= RW; w++) {
                z[w] = 0;
                if (w < nw) { const uint32_t x = __brev(rd[nw - 1 - w]); z[w] = ~(((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u)); }
            }
            const uint32_t s2 = 2u * (uint32_t)(16 * nw - L);
#pragma unroll
            for (int w = 0; w < RW; w++) a[w] = (uint32_t)((((uint64_t)z[w + 1] << 32) | z[w]) >> s2);
        }
        const uint32_t tail = (L & 15) ? ((1u << (2 * (L & 15))) - 1u) : 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w <= RW; w++) { if (w == nw - 1) a[w] &= tail; if (w >= nw) a[w] = 0; s_a[w][lane] = a[w]; }
    }
    // ---- window words, aligned to the window's first base
    {
        const uint64_t q0 = task.win_start >> 4;
        const uint32_t sh = 2u * (uint32_t)(task.win_start & 15);
        const int tw = (m + 15) / 16;
        for (int w = 0; w <= TW; w++) {
            uint32_t v = 0;
            if (w < tw) { const uint64_t two = ((uint64_t)text[q0 + w + 1] << 32) | text[q0 + w]; v = (uint32_t)(two >> sh); }
            s_t[w][lane] = v;
        }
    }
    // b bases starting at base `at` of a word array in LDS (b <= 16)
    const uint32_t bmask = b == 16 ? 0xFFFFFFFFu : ((1u << (2 * b)) - 1u);
    auto bases_a = [&](int at) -> uint32_t {
        const uint64_t two = ((uint64_t)s_a[(at >> 4) + 1][lane] << 32) | s_a[at >> 4][lane];
        return (uint32_t)(two >> (2 * (at & 15))) & bmask;
    };
    // ---- every block against every diagonal: window row (0-based) of read base u on diagonal dl is u + dl
    const int nb = L / b;
    int dlo = 1 << 20, dhi = -(1 << 20);
    for (int k = 0; k < nb; k++) {
        const uint32_t vb = bases_a(k * b);
        int lo = -GBUD, hi = m - L + GBUD;
        if (lo < -k * b) lo = -k * b;                           // the block's rows stay inside the window
        if (hi > m - k * b - b) hi = m - k * b - b;
        // window positions at = k b + dl, one 16-base word of the window at a time: the two words that cover a position's b <= 16 bases
        // are read once per word, the sixteen positions in it are shifts of that pair, their matches one bit each
        const int at_lo = k * b + lo, at_hi = k * b + hi;
        for (int w = at_lo >> 4; w <= (at_hi >> 4); w++) {
            const uint32_t w_lo = s_t[w][lane], w_hi = s_t[w + 1][lane];
            const int j0 = at_lo > 16 * w ? at_lo - 16 * w : 0, j1 = at_hi < 16 * w + 15 ? at_hi - 16 * w : 15;
            // all sixteen positions of the word, shifts known at compile time (v_alignbit_b32), no loop control; the positions outside
            // at_lo .. at_hi are masked off afterwards
            uint32_t mbits = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t v = j == 0 ? w_lo : __builtin_amdgcn_alignbit(w_hi, w_lo, 2 * j);
                mbits |= (v & bmask) == vb ? 1u << j : 0u;
            }
            mbits &= ((2u << j1) - 1u) & ~((1u << j0) - 1u);
            if (mbits) {
                const int d_first = 16 * w + (__ffs((int)mbits) - 1) - k * b, d_last = 16 * w + (31 - __clz((int)mbits)) - k * b;
                dlo = d_first < dlo ? d_first : dlo; dhi = d_last > dhi ? d_last : dhi;
            }
        }
    }
    if (dhi < dlo) { cls[slot] = MP_PRE_NONE; fill[t] = r; return; }
    if (dhi - dlo > 4) { fill[t] = r; return; }                 // wide
    const int centre = (dlo + dhi) >> 1;                        // floor: dlo + 2 >= centre >= dhi - 2
    // ---- exact diagonals among the matching ones; fewest mismatches of the read laid on one of them
    // does the read, bases u0 .. u1 (0-based), equal the window on diagonal dl, 0 <= dl <= m - L?  (rows u + dl)
    auto exact_on = [&](int dl, int u0, int u1) -> bool {
        for (int w = u0 >> 4; w <= (u1 >> 4); w++) {
            const int at = 16 * w + dl;
            const uint64_t two = (((uint64_t)s_t[(at >> 4) + 1][lane] << 32) | s_t[at >> 4][lane]) >> (2 * (at & 15));      // (dl >= 0 here)
            const uint32_t x = (uint32_t)two ^ s_a[w][lane];
            uint32_t mm = (x | (x >> 1)) & 0x55555555u;
            const int lo = u0 > 16 * w ? u0 - 16 * w : 0, hi = u1 < 16 * w + 15 ? u1 - 16 * w : 15;
            mm &= (hi >= 15 ? 0xFFFFFFFFu : ((1u << (2 * (hi + 1))) - 1u)) & ~((1u << (2 * lo)) - 1u);
            if (mm) return false;
        }
        return true;
    };
    int first = -1, exact = 0, kmin = 1 << 20, mcol = 0, mcol2 = 0;       // mcol, mcol2: columns (1-based) of the first two mismatches on the last diagonal looked at
    const uint32_t tail = (L & 15) ? ((1u << (2 * (L & 15))) - 1u) : 0xFFFFFFFFu;
    for (int dl = max(dlo, 0); dl <= min(dhi, m - L); dl++) {
        int kk = 0;
        for (int w = 0; w < nw; w++) {
            const int at = 16 * w + dl;
            const uint64_t two = ((uint64_t)s_t[(at >> 4) + 1][lane] << 32) | s_t[at >> 4][lane];
            uint32_t x = (uint32_t)(two >> (2 * (at & 15))) ^ s_a[w][lane];
            if (w == nw - 1) x &= tail;
            const uint32_t mm = (x | (x >> 1)) & 0x55555555u;
            if (kk == 0 && mm) mcol = 16 * w + ((__ffs((int)mm) - 1) >> 1) + 1;
            { const uint32_t m2 = kk == 0 ? (mm & (mm - 1)) : mm; if (kk <= 1 && kk + __popc(mm) >= 2) mcol2 = 16 * w + ((__ffs((int)m2) - 1) >> 1) + 1; }
            kk += __popc(mm);
            if (kk > 2) break;                                  // only 0, 1, 2 matter
        }
        if (kk == 0) { if (first < 0) first = dl; exact++; }
        kmin = kk < kmin ? kk : kmin;
    }
    // The two short cuts below (one and two mismatches on the only matching diagonal: score L - 3 / L - 6, one tie, no table) are case analyses
    // over THIS scoring: a match scores 1, a mismatch -2 (an event costs 3), the first base of a gap -3 and every further one -1 (a gap of k
    // bases costs k + 2 >= 3).  The scores are compiled in (mp_align_params carries none; the executable refuses any other ini, soap4_main.cpp);
    // should they ever become parameters, these short cuts must be switched off for anything else.
    static_assert(MP_DP_MISMATCH == -2 && MP_DP_OPEN == -3 && MP_DP_EXT == -1, "dp_prefilter_kernel's no-table short cuts are proven for +1/-2/-3/-1 only");
    if (first >= 0) {
        r.score = L; r.row = (uint16_t)(first + L); r.col = (uint16_t)L; r.count = exact > 255 ? 255u : (uint32_t)exact;
        cls[slot] = MP_PRE_DONE;
    } else if (one_mismatch && dlo == dhi && dlo >= 0 && dlo <= m - L && kmin == 1 && mcol >= 4 && mcol <= L - 3 &&
               (mcol <= L - 2 * b + 2 || dlo + 1 > m - L || !exact_on(dlo + 1, mcol - 1, L - 1)) &&
               (mcol >= 2 * b - 1 || dlo - 1 < 0 || !exact_on(dlo - 1, 0, mcol - 1))) {
        // ONE matching diagonal d, on it the read differs from the window in exactly one base, at column c, 4 <= c <= L - 3:
        // the table's maximum is L - 3, in the single cell (d + L, L).  A path of score >= L - 3 has at most one event (each costs >= 3):
        // none -- an exact run of >= L - 3 >= 2b - 1 bases holds a whole block, so it lies on d, where the longest exact run has
        // max(c - 1, L - c) <= L - 4 bases; one mismatch -- L - 1 matches, i.e. a whole diagonal with one mismatch, whose longer exact
        // half holds a block: d itself; one gap -- its k bases cost k + 2, so k = 1, a skipped WINDOW base, and all L read bases match:
        // read[1..j] on one diagonal, read[j+1..L] on the next, both exact; the longer piece holds a block, so it lies on d and avoids c:
        // either read[1..j] on d with j < c and read[j+1..L] on d + 1 -- then read[c..L] is exact on d + 1, which a block rules out when
        // that piece has 2b - 1 bases or more (c <= L - 2b + 2) and the comparison above otherwise -- or read[j+1..L] on d with j >= c and
        // read[1..j] on d - 1: read[1..c] exact on d - 1, ruled out by a block (c >= 2b - 1) or by the comparison.  (A diagonal whose rows
        // leave the window carries no path.)  Every other cell stays below: on d a path ending at column j < L scores at most j - 3.
        // dp_band_classify_kernel then finds the plain diagonal (mp_diag_alignment: c > 3) and writes the result: no table at all.
        r.score = L - 3; r.row = (uint16_t)(dlo + L); r.col = (uint16_t)L; r.count = 1;
        cls[slot] = MP_PRE_DONE;
    } else if (one_mismatch >= 2 && dlo == dhi && dlo >= 0 && dlo <= m - L && kmin == 2 && mcol >= 2 * b - 1 && mcol <= L - 2 * (2 * b - 1) &&
               mcol2 >= 2 * (2 * b - 1) + 1 && mcol2 <= L - (2 * b - 1) + 1 && L >= 3 * (2 * b - 1) + 6) {
        // The same with TWO mismatches on the one matching diagonal d, at columns c1 < c2 well inside the read.  T = 2b - 1: an exact piece of T
        // bases holds a block, so it lies on d; conditions T <= c1 <= L - 2T, 2T + 1 <= c2 <= L - T + 1, L >= 3T + 6.  The maximum is L - 6, in
        // the single cell (d + L, L).  A path that loses at most 6 against L matches has at most two events and three exact pieces with >= L - 6
        // bases between them, so one piece has >= T bases: it lies on d -- before c1, between c1 and c2, or behind c2 -- and every piece on
        // another diagonal has fewer than T.  The cases: only clipped bases (no exact run of L - 6 bases on d); one mismatch, <= 3 clipped (one
        // diagonal: d; it holds c1 or c2 and clips everything from the other one on: >= T bases); two mismatches (one diagonal, d: the path itself);
        // one gap of 1..4 window bases or 1..2 read bases, <= 3 clipped (two pieces: the one on d ends before c1 and the other covers [c1 + 2, L - 3],
        // or starts behind c2 and the other covers [4, c2 - 2]: >= T bases either way); a one-base gap and a mismatch, or two one-base gaps, nothing
        // clipped (three pieces: with the piece(s) on d before c1, what follows on the next diagonal(s) is [c1, L] cut at most once: a piece of
        // >= (L - c1) / 2 >= T; behind c2: [1, c2] cut at most once, >= c2 / 2 >= T; between c1 and c2, or c1 taken as the mismatch: [1, c1] resp.
        // [c2, L] in one piece, >= T).  Each time a second matching diagonal would exist: there is none.
        r.score = L - 6; r.row = (uint16_t)(dlo + L); r.col = (uint16_t)L; r.count = 1;
        cls[slot] = MP_PRE_DONE;
    } else {
        r.score = -2; r.row = (uint16_t)(centre + 256); r.col = 256; r.count = 0;      // row - col = centre, read by the MODE 3 pass
        // At most two mismatches on one of those diagonals: the best score is >= L - 6, and an alignment that loses at most 7 has
        // at most 5 gap bases (a gap of k bases costs k + 2), i.e. stays within 2 + 5 diagonals of the centre: the first pass over
        // 15 diagonals (qualifying score L - 7) is certain to decide it.
        cls[slot] = (GBUD == MP_BAND_GB && kmin <= 2) ? MP_PRE_NARROW15 : MP_PRE_NARROW;
    }
    fill[t] = r;
}

// flag[t] = (cls[t] == which)
__global__ void dp_class_flag_kernel(const uint32_t *__restrict__ cls, uint32_t n, uint32_t which, uint32_t *__restrict__ flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flag[t] = cls[t] == which ? 1u : 0u;
}

// tasks the band did not decide (score -1) are selected for the full kernels; decided ones are traced from the band slab
__global__ void dp_band_select_kernel(const mp_fill_result *__restrict__ fill, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flag[t] = fill[t].score < 0 ? 1u : 0u;
}

// Classes of the instances after the first pass of the two-pass scheme, by how far the best alignment can stray from the
// best cell's diagonal.  score == L: every base matches, the path is the diagonal itself and the result is written here
// (what dp_trace_kernel would find: L diagonal steps, hit_loc = row - L).  score >= L - MP_BAND_NEAR: gaps cost at most
// MP_BAND_NEAR, one gap of at most MP_BAND_NEAR - 2 bases: second pass over 2 MP_BAND_GB1 + 1 diagonals.  Otherwise the
// full second pass (2 MP_BAND_GB + 1 diagonals).  Undecided instances (score < 0) go to the full-table kernels.
#define MP_BAND_NEAR 5
#define MP_BAND_GB1 4
#define MP_BAND_W1 (2 * MP_BAND_GB1 + 1)
#define MP_BAND_NW1 ((MP_BAND_W1 + 2) / 3)
// in between: score >= L - MP_BAND_MID, gaps of at most MP_BAND_MID - 2 bases in all: 2 MP_BAND_GBM + 1 diagonals
#define MP_BAND_MID 9
#define MP_BAND_GBM 7
#define MP_BAND_WM (2 * MP_BAND_GBM + 1)
// Is the alignment the traceback would report for best cell (row, col) with score S simply the diagonal through that cell --
// read bases 1..col against window rows hit+1..hit+col, hit = row - col, no gap, no left clip?  Sufficient condition, from the read and
// the window alone: every prefix of the diagonal scores above zero (the k-th mismatch lies right of column 3k) and the whole of it
// scores S.  Then for every cell c of that diagonal  prefix(c) <= H(c)  (the diagonal from column 0 is one of the paths into it) and
// H(c) <= S - (score of the rest of the diagonal) = prefix(c)  (continuing along the diagonal reaches the best cell, whose value is S):
// H equals the diagonal's own prefix score everywhere on it, so the traceback (CPU_DP.cpp:622-786: diagonal first whenever
// H(r,c) = H(r-1,c-1) + s) takes the diagonal at every step, and its exit test never fires before column 1 -- a predecessor whose value
// is the clip floor would need H = 0, the prefixes are >= 1.  What dp_trace_kernel writes in that case: col diagonal steps after
// L - col clipped bases, hit_loc = row - col.
__device__ __forceinline__ bool mp_diag_alignment(const uint32_t *__restrict__ text, const uint32_t *__restrict__ rd, int L, uint32_t strand, uint64_t win_start, int m,
                                                  int hit, int col, int S)
{
    if (hit < 0 || col < 1 || col > L || hit + col > m) return false;
    const int nw = (L + 15) / 16, nwc = (col + 15) / 16;
    int k = 0;
    for (int w = 0; w < nwc; w++) {
        const uint64_t p = win_start + (uint64_t)(16 * w + hit);
        const uint32_t t = (uint32_t)((((uint64_t)text[(p >> 4) + 1] << 32) | text[p >> 4]) >> (2 * (uint32_t)(p & 15)));
        const uint32_t x = t ^ mp_oriented_word(rd, nw, L, strand, w);
        uint32_t mm = (x | (x >> 1)) & 0x55555555u;
        if (16 * w + 16 > col) mm &= (1u << (2 * (col - 16 * w))) - 1u;
        while (mm) {
            const int b = __ffs((int)mm) - 1;
            mm &= mm - 1;
            k++;
            if (16 * w + (b >> 1) + 1 <= 3 * k) return false;
        }
    }
    return col - 3 * k == S;
}

__global__ void dp_band_classify_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ lens, const mp_fill_result *__restrict__ fill,
                                        uint32_t n, uint32_t *__restrict__ flag_near, uint32_t *__restrict__ flag_far, uint32_t *__restrict__ flag_full,
                                        mp_dp_result *__restrict__ res, const uint32_t *__restrict__ list = nullptr,
                                        uint32_t *__restrict__ flag_vfar = nullptr, uint32_t *__restrict__ flag_mid = nullptr,
                                        const uint32_t *__restrict__ text = nullptr /* with reads: finish plain diagonal alignments here */,
                                        const uint32_t *__restrict__ reads = nullptr, uint32_t Wr = 0)
{
    // flags are indexed by slot; with a list, slot s stands for instance list[s].  The second tier (scores down to L - 25)
    // has a third class, `flag_vfar`: scores below L - MP_BAND_GB, second pass over 51 diagonals.
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const uint32_t t = list ? list[slot] : slot;
    const mp_fill_result fr = fill[t];
    const int L = (int)lens[tasks[t].read_id];
    uint32_t fn = 0, ff = 0, fu = 0, fv = 0, fm = 0;
    mp_dp_result z; z.score = 0; z.hit_loc = 0; z.max_score_count = 0; z.n_ops = 0;
    if (fr.score < 0) fu = 1;
    else if (fr.score == 0 || fr.score < mp_dp_cutoff(L)) res[t] = z;              // nothing to align / below the cutoff
    else if (fr.score == L && fr.col == L && fr.row >= L) {
        z.score = L; z.hit_loc = (uint32_t)(fr.row - L); z.max_score_count = fr.count; z.n_ops = (uint32_t)L;
        res[t] = z;
    } else if (text && mp_diag_alignment(text, reads + (uint64_t)tasks[t].read_id * Wr, L, tasks[t].strand, tasks[t].win_start, (int)tasks[t].win_len,
                                         (int)fr.row - (int)fr.col, (int)fr.col, (int)fr.score)) {
        // mismatches only, on one diagonal, no clip on the left: no second pass, no traceback
        z.score = fr.score; z.hit_loc = (uint32_t)((int)fr.row - (int)fr.col); z.max_score_count = fr.count; z.n_ops = (uint32_t)L;
        res[t] = z;
    } else if (fr.score >= L - MP_BAND_NEAR) fn = 1;
    else if (flag_mid && fr.score >= L - MP_BAND_MID) fm = 1;
    else if (fr.score >= L - MP_BAND_GB || !flag_vfar) ff = 1;
    else fv = 1;
    flag_near[slot] = fn; flag_far[slot] = ff; flag_full[slot] = fu;
    if (flag_vfar) flag_vfar[slot] = fv;
    if (flag_mid) flag_mid[slot] = fm;
}

// list[idx[t] - 1] = t for the flagged entries (idx = inclusive prefix sum of flag)
__global__ void dp_band_list_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ list,
                                    const uint32_t *__restrict__ src = nullptr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && flag[t]) list[idx[t] - 1] = src ? src[t] : t;
}

// tasks / results of the instances of a list, for the full-table kernels
__global__ void dp_gather_tasks_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ list, uint32_t n, mp_dp_task *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = tasks[list[t]];
}
__global__ void dp_scatter_list_kernel(const mp_dp_result *__restrict__ sel, const uint32_t *__restrict__ list, uint32_t n, mp_dp_result *__restrict__ res)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) res[list[t]] = sel[t];
}

// Measurement aid (mp_ctx_set_dp_census): cells a kernel family computes for the instances of a list (nullptr: all n tasks).
// band_w > 0: a band kernel computes band_w cells per read column; band_w == 0: a full-table kernel computes
// (window + 1) x read length.  slot 0 of `out` counts the instances with a window, slots 1.. the cells per family.
__global__ void __launch_bounds__(256)
dp_census_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ list, uint32_t n, const uint32_t *__restrict__ lens, uint32_t band_w,
                 unsigned long long *__restrict__ out_cells, unsigned long long *__restrict__ out_instances)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0, inst = 0;
    if (t < n) {
        const mp_dp_task k = tasks[list ? list[t] : t];
        if (k.win_len) {
            const unsigned long long L = lens[k.read_id];
            cells = band_w ? (unsigned long long)band_w * L : ((unsigned long long)k.win_len + 1ull) * L;
            inst = 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { cells += __shfl_down(cells, o); inst += __shfl_down(inst, o); }
    __shared__ unsigned long long s_c[4], s_i[4];
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = cells; s_i[threadIdx.x >> 6] = inst; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c = s_c[0] + s_c[1] + s_c[2] + s_c[3], i = s_i[0] + s_i[1] + s_i[2] + s_i[3];
        if (c) atomicAdd(out_cells, c);
        if (i && out_instances) atomicAdd(out_instances, i);
    }
}
