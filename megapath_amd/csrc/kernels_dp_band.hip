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
//   band can tie or beat it: the true best cell (first maximum by row, then column), every value on its optimal path and every
//   tie the traceback bits record are identical to the full table's.  Instances that do not reach that score (or whose
//   shape does not fit the band) are marked and re-run by the full kernel -- see run_dp_band in api.hip.
//
// Mapping: ONE LANE per instance pair (the two instances in the halves of packed 16-bit registers, as in dp_fill_kernel),
// sweeping the read columns; the band's W cells of a column live in registers (H and F, in place), the vertical gap
// state E is a running value.  No cross-lane traffic at all, ~31 VALU instructions per cell pair, 1 wave per SIMD.
// Window base codes are staged in LDS, one byte per row and lane holding both instances' 4-bit codes ([row][lane]: a
// wavefront reads 64 consecutive bytes); 16 KB per block keeps two waves per SIMD resident, which the dependent chain of
// a single lane needs to fill the VALU.
#pragma once

#define MP_BAND_MAXL 160            // longest read the band kernel is instantiated for
#define MP_BAND_DMAX 60             // 2 x DP2_MARGIN: widest window surplus of a deep-DP instance
#define MP_BAND_GB 13               // guard diagonals either side; qualifies scores >= L - 13

// cells of the band in one column, words of traceback bits per column (3 cells per 16-bit half)
#define MP_BAND_W (MP_BAND_DMAX + 2 * MP_BAND_GB + 1)
#define MP_BAND_NW ((MP_BAND_W + 2) / 3)

// One band cell of two instances.  Hd enters as H of (row-1, col-1) [same d, previous column] and leaves as H of this
// cell; Hn / Fn are the previous column's H / F one row down (d + 1): the horizontal neighbour; Hc / Ec the cell above
// in this column.  Frame arithmetic as in MP_DP_CELL_CORE (values are x + BIAS - column).
#define MP_BAND_CELL \
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
        "v_pk_max_u16 %[Hd], %[t1], %[floorv]\n\t"                /* H */ \
        "v_xor_b32 %[bits], %[Hd], %[t1]\n\t" \
        "v_pk_min_u16 %[bits], %[bits], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[bits], %[wz], %[fw]\n\t"           /* Z */ \
        "v_xor_b32 %[t1], %[Hd], %[dg]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[wd], %[fw]\n\t"             /* nD */ \
        "v_xor_b32 %[t1], %[Hd], %[Ec]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[we], %[fw]\n\t"             /* nE */ \
        "v_xor_b32 %[t1], %[Ec], %[eo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[weo], %[fw]\n\t"            /* nEO */ \
        "v_xor_b32 %[t1], %[Fd], %[fo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[wfo], %[fw]\n\t"            /* nFO */ \
        "v_pk_sub_u16 %[t1], %[Hd], %[thr] clamp\n\t"             /* score above the qualifying threshold, 0 below it */ \
        "v_pk_mad_u16 %[t1], %[t1], %[k128], %[cnt]\n\t"          /* (excess << 7) | (127 - d): first maximum of the column */ \
        "v_pk_max_u16 %[colcode], %[colcode], %[t1]\n\t" \
        "v_sub_u32 %[cnt], %[cnt], %[k1]"

template <int W, int GB>
__global__ void __launch_bounds__(64)
dp_band_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, int clip_left, int clip_right,
               uint32_t *__restrict__ flags, uint32_t ncols /* columns of NW x 64 words reserved per block */,
               mp_fill_result *__restrict__ out)
{
    constexpr int NW = (W + 2) / 3;
    constexpr int R = MP_BAND_MAXL + W + 1;        // LDS rows: row i of the window sits at index i + GB
    constexpr int B = MP_DP_BIAS;
    __shared__ uint8_t s_code[R][64];                 // code of instance 0 | code of instance 1 << 4
    const int lane = threadIdx.x;
    const uint32_t pair = blockIdx.x * 64 + lane;
    int L[2], m[2], clipL[2], clipR[2], strand[2];
    uint32_t rid[2];
    bool fit[2], empty[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t inst = 2 * pair + q;
        mp_dp_task task; task.read_id = 0; task.strand = 1; task.win_start = 0; task.win_len = 0; task.reserved = 0;
        const bool live = inst < n_tasks;
        if (live) task = tasks[inst];
        L[q] = live ? (int)lens[task.read_id] : 0;
        m[q] = (int)task.win_len;
        strand[q] = (int)task.strand; rid[q] = task.read_id;
        clipL[q] = task.strand == 1 ? clip_left : clip_right;
        clipR[q] = task.strand == 1 ? clip_right : clip_left;
        // the band covers this instance: read short enough, window surplus inside the band, rows <= 0 only under the left clip
        fit[q] = live && L[q] >= 1 && L[q] <= MP_BAND_MAXL && m[q] >= 1 && m[q] - L[q] <= W - 1 - 2 * GB && m[q] + GB < R && clipL[q] >= GB;
        empty[q] = live && task.win_len == 0;          // nothing to align: decided, score 0 (as the full kernels report it)
        if (!fit[q]) { L[q] = 0; m[q] = 0; }
        if (q == 0)
            for (int r = 0; r < R; r++) s_code[r][lane] = (uint8_t)(12 | (15 << 4));      // pads: match nothing
        for (int w = 0; w * 16 < m[q]; w++) {
            uint64_t p = task.win_start + 16ull * (uint64_t)w;
            uint32_t sh = 2 * (uint32_t)(p & 15);
            uint64_t two = (uint64_t)text[p >> 4];
            if (sh) two |= (uint64_t)text[(p >> 4) + 1] << 32;
            uint32_t bases = (uint32_t)(two >> sh);
            for (int x = 0; x < 16 && w * 16 + x < m[q]; x++) {
                uint8_t &cell = s_code[GB + 1 + w * 16 + x][lane];
                const uint32_t code = (uint32_t)MP_DP_CODE(bases & 3u);
                cell = q == 0 ? (uint8_t)((cell & 0xF0u) | code) : (uint8_t)((cell & 0x0Fu) | (code << 4));
                bases >>= 2;
            }
        }
    }
    int Lw = L[0] > L[1] ? L[0] : L[1];
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(Lw, o); Lw = v > Lw ? v : Lw; }
    __syncthreads();

    dp_consts K;
    K.k1 = pk1(1); K.k2 = pk1(2); K.k3 = pk1(3); K.k4 = pk1(4); K.k16 = pk1(16);
    const v2s k128 = pk1(128);
    dp_weights Wt[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        Wt[q].z = pk1((short)(TBP_Z << (5 * q))); Wt[q].d = pk1((short)(TBP_ND << (5 * q))); Wt[q].e = pk1((short)(TBP_NE << (5 * q)));
        Wt[q].eo = pk1((short)(TBP_NEO << (5 * q))); Wt[q].fo = pk1((short)(TBP_NFO << (5 * q)));
    }
    v2s H[W], F[W];
#pragma unroll
    for (int d = 0; d < W; d++) { H[d] = pk1((short)B); F[d] = pk1(MP_DP_SENT); }     // column 0: H = 0, no insertion state
    const v2s sent = pk1(MP_DP_SENT);
    int bu[2] = {0, 0}, brow[2] = {0, 0}, bcol[2] = {0, 0};
    uint32_t rw[2] = {0, 0};                                  // 16 read bases of each instance
    uint32_t *fcol = flags + (uint64_t)blockIdx.x * ncols * (NW * 64) + lane;

    for (int j = 1; j <= Lw; j++) {
        short b2[2], fl2[2], th2[2], el2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            // read base of column j in alignment orientation (packRead, DV-DPfunctions.cpp:3009-3040)
            const int p = strand[q] == 1 ? j - 1 : L[q] - j;
            int b = 9 + q;
            if (j <= L[q]) {
                const uint32_t *rd = reads + (uint64_t)rid[q] * Wr;
                if (j == 1 || (strand[q] == 1 ? (p & 15) == 0 : (p & 15) == 15)) rw[q] = rd[p >> 4];
                b = (int)((rw[q] >> (2 * (p & 15))) & 3u);
                if (strand[q] != 1) b = 3 - b;
                b = MP_DP_CODE(b);
            }
            b2[q] = (short)b;
            fl2[q] = (short)(j <= clipL[q] ? B - j : 0);
            th2[q] = (short)(L[q] - GB - 1 + B - j);          // score L - GB - 1 in this column's frame
            int mc = L[q] - clipR[q]; if (mc < 0) mc = 0;
            el2[q] = (short)((j >= mc && j <= L[q]) ? -1 : 0);
        }
        const v2s rb = pk(b2[0], b2[1]), floorv = pk(fl2[0], fl2[1]), thr = pk(th2[0], th2[1]);
        v2s Ec = sent, colcode = pk1(0), cnt = pk1(127);
        v2s fw = pk1(0);
        const uint8_t *c0 = &s_code[j][lane];
#pragma unroll
        for (int d = 0; d < W; d++) {
            const uint32_t cb = c0[d * 64];                                    // row j + d - GB of both windows
            const v2s base = as_v2s((int)((cb | (cb << 12)) & 0x000F000Fu));
            v2s eo, fo, dg, t1, bits;
            const v2s Hc = d == 0 ? sent : H[d - 1];
            const v2s Hn = d + 1 < W ? H[d + 1] : sent;
            const v2s Fn = d + 1 < W ? F[d + 1] : sent;
            asm(MP_BAND_CELL
                : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [Hd] "+&v"(H[d]), [Ec] "+&v"(Ec),
                  [Fd] "=&v"(F[d]), [fw] "+&v"(fw), [colcode] "+&v"(colcode), [cnt] "+&v"(cnt)
                : [Hc] "v"(Hc), [Hn] "v"(Hn), [Fn] "v"(Fn), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv), [thr] "v"(thr),
                  [k1] "v"(K.k1), [k2] "v"(K.k2), [k3] "v"(K.k3), [k4] "v"(K.k4), [k128] "v"(k128),
                  [wz] "v"(Wt[d % 3].z), [wd] "v"(Wt[d % 3].d), [we] "v"(Wt[d % 3].e), [weo] "v"(Wt[d % 3].eo), [wfo] "v"(Wt[d % 3].fo));
            if (d % 3 == 2 || d == W - 1) {
                fcol[(uint64_t)(j - 1) * (NW * 64) + (d / 3) * 64] = __builtin_bit_cast(uint32_t, fw);
                fw = pk1(0);
            }
        }
        // first maximum of this column against the best so far: larger score, then smaller row (columns ascend)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int code = (int)(unsigned short)(q == 0 ? colcode.x : colcode.y) & (int)(unsigned short)el2[q];
            const int u = code >> 7, row = j + (127 - (code & 127)) - GB;
            if (u > bu[q] || (u == bu[q] && u > 0 && row < brow[q])) { bu[q] = u; brow[q] = row; bcol[q] = j; }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t inst = 2 * pair + q;
        if (inst >= n_tasks) continue;
        mp_fill_result r;
        // score -1: "not decided by the band" -- below the qualifying score, outside the band's shape, or the first
        // maximum fell on a pad row past the window
        const bool ok = fit[q] && bu[q] >= 1 && brow[q] >= 1 && brow[q] <= m[q];
        r.score = ok ? bu[q] + L[q] - GB - 1 : (empty[q] ? 0 : -1);
        r.row = ok ? (uint16_t)brow[q] : 0; r.col = ok ? (uint16_t)bcol[q] : 0; r.count = 0;
        out[inst] = r;
    }
}

// tasks the band did not decide (score -1) are selected for the full kernels; decided ones are traced from the band slab
__global__ void dp_band_select_kernel(const mp_fill_result *__restrict__ fill, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flag[t] = fill[t].score < 0 ? 1u : 0u;
}
