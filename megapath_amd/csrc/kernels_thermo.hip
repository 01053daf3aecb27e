// megapath_amd/csrc/kernels_thermo.hip -- product code: an exact, cheaper rejection in front of the score-only filter of the single-end
// rescue (dp_score_kernel): a CERTIFICATE that an instance cannot reach its cutoff.  Included by api.hip.
//
// The instances of the single-end rescue (DV-DPForSingleReads.cpp:121-215, SingleEndAlgnBatch::pack DV-DPfunctions.cpp:401-445) are windows
// around a seed: read (L <= 160) x window (m = L + 60) with an exact match of >= 17 bases on diagonal 30.  With 17-base seeds nearly all of
// them are chance matches whose table holds nothing else; the filter still fills all 31 500 cells.  What this kernel computes instead
// (proof: DESIGN.md section 4, "The rescue filter"; numerically: scripts/dp_decomposition_check.py, scripts/thermo_bound_proto.py):
//
//   T'   the table with the cells of the longest exact run R on the seed's diagonal turned into mismatches;
//   H'   its LOCAL scores (+1 / -2, gaps -3 / -1, floor 0 in every cell: never below what callDP computes, CPU_DP.cpp:122-619);
//   max(T) <= max( max H',  max over the cells of R but its last, and the cell diagonally before R, of H'  +  |R|  +  max of H' right of and below R's first cell )
//
// because a path either never uses a cell of R diagonally (a path of T') or is a path of T' into R, a stretch of R (leaving R and coming
// back costs two gaps and skips matches), and a path of T' out of R.  If that bound is below the cutoff the instance is decided: no alignment.
//
// H' is never formed as numbers.  PLANES H_t = [H' >= t], t = 1 .. T (10), one bit per instance: a word holds the SAME cell of 32 instances, so
//   E_t = Hup_{t+3} | Eup_{t+1}        F_t = Hleft_{t+3} | Fleft_{t+1}        D_t = (M & Hdiag_{t-1}) | Hdiag_{t+2}        H_t = D_t | E_t | F_t
// are four logic instructions per plane and cell for 32 instances at once, with no shifts and no carries (M = the instances whose window base
// equals their read base in this cell).  A plane T bit anywhere means max H' >= T: no certificate (the filter runs as before).  T = 10: a
// random 150 x 210 table reaches 10 in about 2 % of the instances (12: 0.2 %), and two planes fewer are 20 % fewer instructions.
// Mapping: a wavefront takes 64 instances, 32 per half; lane l of a half owns read columns 5l .. 5l+4 and walks the window rows on an
// anti-diagonal (row = step - l): what it needs from the column to its left comes from lane l-1's previous step by DPP.  The reads and
// windows are transposed into bit-sliced form with one ballot per word (both halves at once), straight into LDS.
// Measured (profiles/r03_*): 60 single-end instances per pair and chunk of the NT stage, 95 % certified; 6.6 ms for 3.95 M instances where
// dp_score_kernel took 18.5 ms -- 230 instructions per step of five cells x 32 instances, 1.4 per cell and instance.
//
// Round 4: the same planes for what the single-end windows of 101 .. 320-base reads left out.  MARGIN = 25: reads of up to 100 bases (their
// windows are read + 2 x 25, DV-DPfunctions.cpp:1760).  RUNLESS: the windows of the MATE RESCUE (HalfEndAlgnBatch::pack,
// DV-DPfunctions.cpp:1151-1231: up to insert_high - insert_low + L + 1 = 751 rows) hold no seed at all -- where the mate is not there the table
// is a random one, no plane T bit appears, max H' <= T - 1 < 30 <= cutoff and the instance is decided; no run, no bound, one accumulator.
#pragma once

#define MP_TH_T 10                 // planes: values 1 .. T; plane T set anywhere = overflow, no certificate (E and F never exceed max H' - 3: T - 3 planes)
#define MP_TH_T_WIDE 11             // the wide variant's tables are up to four times larger: one plane more keeps the overflows at a few per cent
#define MP_TH_COLS 5               // read columns per lane: 32 lanes x 5 = 160 (two sets of 32 instances per wavefront), or 64 lanes x 5 = 320 (one set)
#define MP_TH_MAXL 160
#define MP_TH_MAXM 224
#define MP_TH_MAXL_WIDE 320
#define MP_TH_MAXM_WIDE 384
#define MP_TH_MARGIN 30            // DP2_MARGIN for reads longer than 100 bases (DV-DPfunctions.cpp:1760): the seed's diagonal
#define MP_TH_MARGIN_SHORT 25      // ... and for reads of up to 100 bases
#define MP_TH_MAXL_SHORT 100
#define MP_TH_MAXM_SHORT 160       // (reads of <= 100 bases: windows of <= 150)
#define MP_TH_MAXM_RESCUE 768      // mate-rescue windows: 751 rows with the pipeline's -u 750 (insert_low = the longest read of the first batch)
#define MP_TH_T_RESCUE 11          // their tables are 3.6 times larger than a single-end window's: one plane more keeps the overflows at a few per cent
#define MP_TH_T_RESCUE_WIDE 12
#define MP_TH_MINRUN 12            // a run shorter than this is no seed: no certificate

// Lane l receives lane l-1's planes 1 .. N AND-ed with the mask of "has a left neighbour", one v_and_b32_dpp each (lane 0, which has no source, reads 0).
// A DPP operand read within two wait states of the VALU instruction that wrote the register returns the register's OLD content (gfx9 data hazard; five
// wait states after a VALU write of EXEC), and the compiler's hazard recognizer does not look into asm statements.  Until round 4 these were N separate
// one-instruction asm statements, safe only by the luck of the schedule: a build whose loop put a producer next to one of them certified differently
// from run to run (found by comparing two builds instance by instance).  Hence ONE statement per group: every input is complete when it begins, nothing
// can be scheduled into it, and it opens with the wait states itself.  (The compiler's own DPP builtin handles the hazard too, but turns the AND into a
// select, does not fold the move, and the kernel then spills.)
// In place (destination = source register of the own lane: the sources of all lanes are read before any is written), so the group needs no register more.
#define TH_L(k) "v_and_b32_dpp %[x" #k "], %[x" #k "], %[m] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define TH_X(k) [x##k] "+v"(x[k])
#define TH_LINES_7 TH_L(1) TH_L(2) TH_L(3) TH_L(4) TH_L(5) TH_L(6) TH_L(7)
#define TH_REGS_7 TH_X(1), TH_X(2), TH_X(3), TH_X(4), TH_X(5), TH_X(6), TH_X(7)
#define TH_COMMA ,
#define TH_GROUP(lines, regs) asm("s_nop 4\n\t" lines : regs : [m] "v"(m))
template <int N>
__device__ __forceinline__ void th_from_left_and(uint32_t (&x)[N + 1], uint32_t m)
{
    static_assert(N >= 7 && N <= 12, "planes per group");
    if constexpr (N == 7) TH_GROUP(TH_LINES_7, TH_REGS_7);
    else if constexpr (N == 8) TH_GROUP(TH_LINES_7 TH_L(8), TH_REGS_7 TH_COMMA TH_X(8));
    else if constexpr (N == 9) TH_GROUP(TH_LINES_7 TH_L(8) TH_L(9), TH_REGS_7 TH_COMMA TH_X(8) TH_COMMA TH_X(9));
    else if constexpr (N == 10) TH_GROUP(TH_LINES_7 TH_L(8) TH_L(9) TH_L(10), TH_REGS_7 TH_COMMA TH_X(8) TH_COMMA TH_X(9) TH_COMMA TH_X(10));
    else if constexpr (N == 11) TH_GROUP(TH_LINES_7 TH_L(8) TH_L(9) TH_L(10) TH_L(11), TH_REGS_7 TH_COMMA TH_X(8) TH_COMMA TH_X(9) TH_COMMA TH_X(10) TH_COMMA TH_X(11));
    else TH_GROUP(TH_LINES_7 TH_L(8) TH_L(9) TH_L(10) TH_L(11) TH_L(12), TH_REGS_7 TH_COMMA TH_X(8) TH_COMMA TH_X(9) TH_COMMA TH_X(10) TH_COMMA TH_X(11) TH_COMMA TH_X(12));
}

// cert[t] = 1: instance t cannot reach its cutoff (no alignment); 0: undecided, the filter has to look at it.
// `list` (may be null): slot s stands for instance list[s].
// LPS = lanes per set of 32 instances: 32 (reads of up to 160 bases, two sets per wavefront) or 64 (up to 320 bases, one set: lanes 0 .. 31 bring the instances)
// MARGIN: the diagonal the seed lies on (window row = read column + MARGIN); RUNLESS: no seed (mate-rescue windows); MAXM_P: longest window
template <int T, int LPS, int MARGIN, bool RUNLESS, int MAXM_P>
__device__ __forceinline__ void
dp_thermo_body(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, uint8_t *__restrict__ cert, const uint32_t *__restrict__ list,
               unsigned long long *__restrict__ stats /* [0] certified [1] overflow [2] bound too high [3] not eligible */)
{
    constexpr int TE = T - 3, NC = MP_TH_COLS;
    static_assert(T >= 10 && T <= 12, "planes (th_from_left_and: groups of T and T - 3)");
    static_assert(LPS == 32 || LPS == 64, "lanes per set");
    constexpr int SETS = 64 / LPS, MAXL = NC * LPS, MAXM = MAXM_P;
    static_assert(MAXM % 16 == 0, "whole words of window");
    constexpr int OFF = LPS, ROWS = MAXM + 2 * LPS + 8;      // window rows are stored OFF further down, with empty rows on either side: the anti-diagonal
                                                             // sweep reads rows -(LPS - 1) .. mmax + LPS + 4 without a test
    // bit-sliced inputs of the sets (index h): one word = one column (row) of 32 instances
    constexpr int XL = RUNLESS ? 1 : MAXL, XR = RUNLESS ? 1 : ROWS;             // (the run's planes do not exist without a run)
    __shared__ uint32_t s_r0[SETS][MAXL], s_r1[SETS][MAXL], s_rv[SETS][MAXL], s_run[SETS][XL], s_lc[SETS][XL], s_gt[SETS][XL];
    __shared__ uint32_t s_w0[SETS][ROWS], s_w1[SETS][ROWS], s_wv[SETS][ROWS], s_rg[SETS][XR];
    const int lane = threadIdx.x, h = LPS == 32 ? lane >> 5 : 0, l = lane & (LPS - 1);
    const int bit = lane & 31;                                // the instance this lane brings (and speaks for at the end), if it brings one
    const bool carrier = lane < 32 * SETS;
    const uint32_t slot = blockIdx.x * (32 * SETS) + lane;
    const bool have = carrier && slot < n_tasks;
    const uint32_t t_id = have ? (list ? list[slot] : slot) : 0;
    // ---------------------------------------------------------------- this lane's instance: oriented read, window, the seed's run
    int L = 0, m = 0, j0 = 0, rlen = 0;
    bool ok = false;
    uint32_t a[MAXL / 16 + 1], w[MAXM / 16 + 2];
#pragma unroll
    for (int k = 0; k <= MAXL / 16; k++) a[k] = 0;
#pragma unroll
    for (int k = 0; k < MAXM / 16 + 2; k++) w[k] = 0;
    if (have) {
        const mp_dp_task task = tasks[t_id];
        L = (int)lens[task.read_id]; m = (int)task.win_len;
        if (RUNLESS) ok = L >= MP_MIN_ALIGNABLE_LEN && L <= MAXL && m >= 1 && m <= MAXM;
        else if (MARGIN == MP_TH_MARGIN) ok = L > 100 && L <= MAXL && m >= L && m <= MAXM && m <= L + 2 * MARGIN;
        else ok = L >= MP_MIN_ALIGNABLE_LEN && L <= 100 && L <= MAXL && m >= L && m <= MAXM && m <= L + 2 * MARGIN;
        if (ok) {
            const uint32_t *rd = reads + (uint64_t)task.read_id * Wr;
            const int nw = (L + 15) / 16;
#pragma unroll
            for (int k = 0; k < MAXL / 16; k++) a[k] = mp_oriented_word(rd, nw, L, task.strand, k);
            const uint64_t q0 = task.win_start >> 4;
            const uint32_t sh = 2u * (uint32_t)(task.win_start & 15);
            const int tw = (m + 15) / 16;
#pragma unroll
            for (int k = 0; k < MAXM / 16 + 1; k++)
                if (k < tw) {
                    uint64_t two = (uint64_t)text[q0 + k];
                    if (sh) two |= (uint64_t)text[q0 + k + 1] << 32;              // (as dp_score_kernel: never a word past the window's last)
                    w[k] = (uint32_t)(two >> sh);
                }
            // the longest run of matches on diagonal MARGIN (window row = read column + MARGIN)
            int best = 0, best_at = 0, cur = 0;
            const int lim = min(L, m - MARGIN);
            if (!RUNLESS)
#pragma unroll
            for (int k = 0; k < MAXL / 16; k++) {
                const int at = 16 * k + MARGIN;                              // window base of read base 16 k
                const uint64_t two = ((uint64_t)w[(at >> 4) + 1] << 32) | w[at >> 4];
                uint32_t x = (uint32_t)(two >> (2 * (at & 15))) ^ a[k];
                x = (x | (x >> 1)) & 0x55555555u;                           // bit 2u set: base 16 k + u differs
                for (int u = 0; u < 16; u++) {
                    const int j = 16 * k + u;
                    const bool eq = j < lim && !((x >> (2 * u)) & 1u);
                    cur = eq ? cur + 1 : 0;
                    if (cur > best) { best = cur; best_at = j - cur + 1; }
                }
            }
            j0 = best_at; rlen = best;
            if (!RUNLESS && rlen < MP_TH_MINRUN) ok = false;
        }
    }
    // (the lanes that own read columns: see `steps` below)
    int lmax = ok ? L : 0;
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o));
    const int n_lanes = __builtin_amdgcn_readfirstlane(min(LPS, (lmax + NC - 1) / NC));
    // ---------------------------------------------------------------- transposition: one ballot per word, both halves at once
#pragma unroll
    for (int k = 0; k < MAXL / 16; k++) {
        uint32_t x = a[k];
        for (int u = 0; u < 16; u++, x >>= 2) {
            const int j = 16 * k + u;
            const bool v = ok && j < L;
            const unsigned long long b0 = __ballot(v && (x & 1u)), b1 = __ballot(v && (x & 2u)), bv = __ballot(v);
            if (l == 0) { s_r0[h][j] = (uint32_t)(b0 >> (32 * h)); s_r1[h][j] = (uint32_t)(b1 >> (32 * h)); s_rv[h][j] = (uint32_t)(bv >> (32 * h)); }
            if (!RUNLESS) {
                const unsigned long long br = __ballot(ok && j >= j0 && j < j0 + rlen);                 // a cell of the run sits in this column
                const unsigned long long bl = __ballot(ok && j >= j0 - 1 && j < j0 + rlen - 1);         // the cells whose H' enters the bound on the run's left
                const unsigned long long bg = __ballot(ok && j > j0);                                   // right of the run's first cell
                if (l == 0) { s_run[h][j] = (uint32_t)(br >> (32 * h)); s_lc[h][j] = (uint32_t)(bl >> (32 * h)); s_gt[h][j] = (uint32_t)(bg >> (32 * h)); }
            }
        }
    }
    int mmax = 0;
    for (int hh = 0; hh < SETS; hh++) {
        for (int x = lane; x < OFF; x += 64) { s_w0[hh][x] = 0; s_w1[hh][x] = 0; s_wv[hh][x] = 0; if (!RUNLESS) s_rg[hh][x] = 0; }
        for (int x = OFF + MAXM + lane; x < ROWS; x += 64) { s_w0[hh][x] = 0; s_w1[hh][x] = 0; s_wv[hh][x] = 0; if (!RUNLESS) s_rg[hh][x] = 0; }
    }
#pragma unroll
    for (int k = 0; k < MAXM / 16; k++) {
        uint32_t x = w[k];
        for (int u = 0; u < 16; u++, x >>= 2) {
            const int i = 16 * k + u;
            const bool v = ok && i < m;
            const unsigned long long b0 = __ballot(v && (x & 1u)), b1 = __ballot(v && (x & 2u)), bv = __ballot(v);
            if (l == 0) { s_w0[h][OFF + i] = (uint32_t)(b0 >> (32 * h)); s_w1[h][OFF + i] = (uint32_t)(b1 >> (32 * h)); s_wv[h][OFF + i] = (uint32_t)(bv >> (32 * h)); }
            if (!RUNLESS) {
                const unsigned long long bg = __ballot(ok && i > j0 + MARGIN);                      // rows below the run's first cell
                if (l == 0) s_rg[h][OFF + i] = (uint32_t)(bg >> (32 * h));
            }
            if (bv) mmax = i + 1;                                              // (wave-uniform)
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- this lane's columns
    uint32_t R0[NC], R1[NC], RV[NC], RUN[NC], LC[NC], GT[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int j = NC * l + c;
        R0[c] = s_r0[h][j]; R1[c] = s_r1[h][j]; RV[c] = s_rv[h][j];
        if (RUNLESS) { RUN[c] = 0; LC[c] = 0; GT[c] = 0; } else { RUN[c] = s_run[h][j]; LC[c] = s_lc[h][j]; GT[c] = s_gt[h][j]; }
    }
    uint32_t H[NC][T + 1], E[NC][TE + 1];          // [t], t = 1 ..; index 0 unused
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int t = 0; t <= T; t++) H[c][t] = 0;
#pragma unroll
        for (int t = 0; t <= TE; t++) E[c][t] = 0;
    }
    uint32_t inH[T + 1], inG[TE + 1];              // from the lane to the left: H planes of its last column one row up; G_t = H_{t+3} | F_{t+1} of that column, this row
#pragma unroll
    for (int t = 0; t <= T; t++) inH[t] = 0;
#pragma unroll
    for (int t = 0; t <= TE; t++) inG[t] = 0;
    uint32_t outH[T + 1], outG[TE + 1];
#pragma unroll
    for (int t = 0; t <= T; t++) outH[t] = 0;
#pragma unroll
    for (int t = 0; t <= TE; t++) outG[t] = 0;
    uint32_t accR[T + 1], accL[T + 1], ov = 0;     // accR: planes 5 .. T-1 (a region maximum below 4 counts as 4); accL: planes 1 .. 5
#pragma unroll
    for (int t = 0; t <= T; t++) { accR[t] = 0; accL[t] = 0; }
    const uint32_t nb = l ? 0xFFFFFFFFu : 0u;       // the first lane of a half has no left neighbour: its column -1 is the table's edge (H = 0)
    // The sweep ends when the last lane that OWNS read columns has done the last row: the lanes right of the longest read of the wavefront's instances hold no
    // cell of any table (data only flows to the right and down, and what a gap carries into such a column is below the cell it left, which the accumulators
    // have seen), so they need not finish.  150-base reads: 30 lanes, 210 + 29 = 239 steps = 40 turns of six instead of 41.
    // (the loop below is a do-while on purpose: as a for loop with this bound the narrow variants spill registers into scratch memory inside the sweep)
    const int steps = mmax + max(n_lanes, 1) - 1;
    // Rows above the table (a lane starts l steps late) need no test: every input there is 0 and stays 0.  Rows below it are computed as rows
    // that match nothing: what appears there is never above the cell it came from, so the accumulators may see it.
    // The seed's diagonal meets lane l in its column K at step s = (NC + 1) l + MARGIN + K: with the steps taken NC + 1 at a time, K is a
    // compile-time constant of each of them (K = NC: between two lanes) and only ONE lane of a half is on the diagonal at all.
    // (another margin: the same with the columns rotated -- K = step number - MARGIN modulo NC + 1)
    auto step = [&](auto kc, const int s) __attribute__((always_inline)) {
        constexpr int K = RUNLESS ? NC + 1 : (decltype(kc)::value + (NC + 1) - MARGIN % (NC + 1)) % (NC + 1);      // (run-less: no column is ever on a seed's diagonal)
        // what the left neighbour published at the end of the previous step: its last column in MY row (G) and, kept from the step before, one row up (H)
        uint32_t gotH[T + 1], gotG[TE + 1];
#pragma unroll
        for (int t = 1; t <= T; t++) gotH[t] = outH[t];
#pragma unroll
        for (int t = 1; t <= TE; t++) gotG[t] = outG[t];
        th_from_left_and<T>(gotH, nb);
        th_from_left_and<TE>(gotG, nb);
        const int ri = s - l + OFF;                                             // my window row (i = s - l), as stored
        const uint32_t wv = s_wv[h][ri], w0 = s_w0[h][ri], w1 = s_w1[h][ri], rowgt = RUNLESS ? 0u : s_rg[h][ri];
        const uint32_t onl = !RUNLESS && (NC + 1) * l + MARGIN + K == s ? 0xFFFFFFFFu : 0u;   // this lane's column K is on the seed's diagonal
        // diagonal predecessor of column 0 = the neighbour's last column one row up (kept in inH); F-input of column 0 = gotG
        uint32_t dH[T + 3];                           // H planes of the cell diagonally before the current column; [0] = all instances, [T+1 ..] = 0
        dH[0] = 0xFFFFFFFFu; dH[T + 1] = 0; dH[T + 2] = 0;
#pragma unroll
        for (int t = 1; t <= T; t++) dH[t] = inH[t];
        // G_t = F_t of the current cell = H_{t+3} | F_{t+1} of the cell to its left; for column 0 it arrives from the neighbour
        uint32_t G[TE + 2];
        G[TE + 1] = 0;
#pragma unroll
        for (int t = 1; t <= TE; t++) G[t] = gotG[t];
        uint32_t U[T + 1];                            // planes 5 .. T-1 of this row, OR-ed over the lane's columns
#pragma unroll
        for (int t = 0; t <= T; t++) U[t] = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            uint32_t M = ~((R0[c] ^ w0) | (R1[c] ^ w1)) & RV[c] & wv;
            if (c == K) M &= ~(RUN[c] & onl);                                   // T': the run's cells are mismatches
            // vertical gaps first: they read this column's planes of the row above, which are overwritten below
#pragma unroll
            for (int t = 1; t <= TE; t++) E[c][t] = H[c][t + 3] | (t < TE ? E[c][t + 1] : 0u);
            uint32_t nH[T + 4];
            nH[T + 1] = 0; nH[T + 2] = 0; nH[T + 3] = 0;
#pragma unroll
            for (int t = T; t >= 1; t--) {
                const uint32_t d = (M & dH[t - 1]) | dH[t + 2];
                nH[t] = t <= TE ? (d | E[c][t] | G[t]) : d;
            }
            // next column: its diagonal predecessor is this column's row above, the cell to its left this column's new row
#pragma unroll
            for (int t = 1; t <= T; t++) { dH[t] = H[c][t]; H[c][t] = nH[t]; }
#pragma unroll
            for (int t = 1; t <= TE; t++) G[t] = nH[t + 3] | G[t + 1];         // ascending: G[t + 1] is still the current cell's F_{t+1}
            // the bound's ingredients
            ov |= nH[T];
            if (!RUNLESS)
#pragma unroll
            for (int t = 5; t < T; t++) U[t] |= nH[t];
            if (!RUNLESS && c == K) {
                const uint32_t lcm = LC[c] & onl;
#pragma unroll
                for (int t = 1; t <= 5; t++) accL[t] |= nH[t] & lcm;
            }
        }
        // the region right of and below the run's first cell, a lane's columns taken together (up to NC - 1 columns more than the region:
        // a larger maximum can only withhold a certificate)
        const uint32_t reg = rowgt & GT[NC - 1];
        if (!RUNLESS)
#pragma unroll
        for (int t = 5; t < T; t++) accR[t] |= U[t] & reg;
        // publish for the right neighbour's next step; remember what the left neighbour's row was for my next step's diagonal
#pragma unroll
        for (int t = 1; t <= T; t++) { inH[t] = gotH[t]; outH[t] = H[NC - 1][t]; }
#pragma unroll
        for (int t = 1; t <= TE; t++) outG[t] = G[t];
    };
    static_assert(NC == 5, "six steps per turn");
    int s = 0;
    do {                                              // (the last turn may run up to five rows further down: rows that match nothing)
        step(std::integral_constant<int, 0>{}, s); step(std::integral_constant<int, 1>{}, s + 1); step(std::integral_constant<int, 2>{}, s + 2);
        step(std::integral_constant<int, 3>{}, s + 3); step(std::integral_constant<int, 4>{}, s + 4); step(std::integral_constant<int, 5>{}, s + 5);
        s += NC + 1;
    } while (s < steps);
    // ---------------------------------------------------------------- per instance: OR over the lanes of its half, then read its bit
    auto half_or = [&](uint32_t v) -> uint32_t {
        for (int o = 1; o < LPS; o <<= 1) v |= (uint32_t)__shfl_xor((int)v, o);
        return v;
    };
    ov = half_or(ov);
    int Rs = 4, Lm = 0;
    bool l_big = false;
#pragma unroll
    for (int t = 5; t < T; t++) { const uint32_t v = half_or(accR[t]); if ((v >> bit) & 1u) Rs = t; }
#pragma unroll
    for (int t = 1; t <= 5; t++) { const uint32_t v = half_or(accL[t]); if ((v >> bit) & 1u) { Lm = t; if (t == 5) l_big = true; } }
    if (!have) return;
    // the lane that brought an instance speaks for it
    const bool over = (ov >> bit) & 1u;
    const int cutoff = mp_dp_cutoff(L);
    const bool certified = RUNLESS ? (ok && !over && T - 1 < cutoff)                                 // no cell reaches T: max H' <= T - 1 < cutoff
                                   : (ok && !over && !l_big && T - 1 < cutoff && Lm + rlen + Rs < cutoff);       // (max H' <= T - 1 when there is no overflow)
    cert[t_id] = certified ? 1 : 0;
    if (stats) {
        const unsigned long long c0 = __ballot(certified), c1 = __ballot(ok && over), c2 = __ballot(ok && !over && !certified), c3 = __ballot(!ok);
        if (lane == 0) {
            if (c0) atomicAdd(&stats[0], (unsigned long long)__popcll(c0));
            if (c1) atomicAdd(&stats[1], (unsigned long long)__popcll(c1));
            if (c2) atomicAdd(&stats[2], (unsigned long long)__popcll(c2));
            if (c3) atomicAdd(&stats[3], (unsigned long long)__popcll(c3));
        }
    }
}

template <int T, int LPS, int MARGIN = MP_TH_MARGIN, bool RUNLESS = false, int MAXM_P = (LPS == 32 ? MP_TH_MAXM : MP_TH_MAXM_WIDE)>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
dp_thermo_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
                 const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, uint8_t *__restrict__ cert, const uint32_t *__restrict__ list, unsigned long long *__restrict__ stats)
{
    dp_thermo_body<T, LPS, MARGIN, RUNLESS, MAXM_P>(text, reads, lens, Wr, tasks, n_tasks, cert, list, stats);
}

// key[s] = window length of the instance in slot s of a class list: a wavefront sweeps as many rows as the LONGEST window among its instances, so the
// lists of classes with mixed read lengths are handed over sorted by window length (run_dp_filtered)
__global__ void thermo_window_key_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ list, uint32_t n, uint32_t *__restrict__ key)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) key[s] = tasks[list[s]].win_len;
}

// flag[t] = 1 for the instances the certificate left undecided; the certified ones get the filter's answer "score 0"
__global__ void thermo_flag_kernel(const uint8_t *__restrict__ cert, uint32_t n, uint32_t *__restrict__ flag, mp_fill_result *__restrict__ fill)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool c = cert[t] != 0;
    flag[t] = c ? 0u : 1u;
    if (c) { mp_fill_result z; z.score = 0; z.row = 0; z.col = 0; z.count = 0; fill[t] = z; }
}
