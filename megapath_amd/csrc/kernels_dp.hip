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

// Packed 16-bit arithmetic: every lane computes the same cell of TWO instances at once (low / high half of a VGPR) with
// v_pk_add_i16 / v_pk_max_i16 / v_pk_min_u16 / v_pk_mad_u16.  There is no packed compare on gfx950, so every decision
// bit is formed arithmetically: for x >= 0, min_u16(x, 1) is "x != 0".  Stored bits are therefore the *negations*
// nD, nE, nEO, nFO next to Z; the traceback inverts them.
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

#define TBP_Z 1u
#define TBP_ND 2u
#define TBP_NE 4u
#define TBP_NEO 8u
#define TBP_NFO 16u
#define MP_DP_LPI_MAX 32       // lanes per instance pair: 16 (reads <= 192 bases) or 32
#define MP_DP_NEG16 ((short)-16000)

// One instruction each, opaque to the optimiser (it would otherwise turn min_u16(x,1) into an unpacked compare+select)
#define MP_PKOP2(name, ins) \
    __device__ __forceinline__ v2s name(v2s a, v2s b) { v2s r; asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MP_PKOP2(pk_add, "v_pk_add_i16")
MP_PKOP2(pk_sub, "v_pk_sub_i16")
MP_PKOP2(pk_max, "v_pk_max_i16")
MP_PKOP2(pk_min, "v_pk_min_i16")
MP_PKOP2(pk_minu, "v_pk_min_u16")
__device__ __forceinline__ v2s pk_mad(v2s a, v2s b, v2s c) { v2s r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ v2s pk_xor(v2s a, v2s b) { return a ^ b; }
// lane k receives lane k-1's value (wavefront shift right by one; the first lane of each 32-lane group is overridden by the caller)
// (lane 0, which has no source, keeps its own value: it is a first lane, overridden anyway -- no zero to materialise per call)
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false); }

// Number representation inside dp_fill_kernel.  A cell value x of read column j is held as the unsigned 16-bit number
//   x + MP_DP_BIAS - j
// in its half of a VGPR.  (1) Unsigned and never near 0 or 2^16 (|x| < ~1500, j <= 384), so one 32-bit v_sub_u32 on the
// whole register subtracts small constants from both halves without a borrow crossing them: the five adds of the
// recurrences run at full rate instead of half rate.  (2) The column term cancels the +1 of a match along the diagonal:
// diag - {0 match, 3 mismatch} is the diagonal candidate, with the penalty read straight off min_u16(base XOR read, 3)
// because the base codes are 0, 3, 5, 6 (every pair of different codes XORs to >= 3; the pad codes 9, 10, 12, 15 too).
// Within a column all values share the frame, so the equality tests behind the traceback bits are unaffected; across
// columns (best-cell tracking) the column term is added back inside the one multiply-add that builds the row code.
#define MP_DP_BIAS 2048
#define MP_DP_SENT 8            // "minus infinity" for E of row 0 and F of column 0: consumed once, against real values

// constants the cell update needs in registers (VOP3P / VOP2 operands), both halves equal
struct dp_consts { v2s k1, k2, k3, k4, k16; };
// weights of the five traceback bits inside a flag word, for the three cells sharing a 16-bit half
struct dp_weights { v2s z, d, e, eo, fo; };

// One DP cell of two instances in ONE asm statement (hipcc pads every separate asm with s_nop).
//   in : Hup/Eup (row above), Hl/Fl (cell to the left), diag, base (window bases), rb (read bases), floorv, elig
//   out: e, f, h for this cell; fw (flag accumulator); COUNT=false: rowcode = max over the row of
//        ((eligible H + bias) << 4 | 15-cc) so that the first maximum of the row wins; COUNT=true: exact per-cell
//        first-maximum tracking with tie counts (jadd returns the value to the column-free frame first).
// Decision bits are arithmetic: x != 0  <=>  min_u16(x, 1); differences are formed with v_xor_b32 (full rate on gfx950;
// v_pk_* and every min/max run at half rate -- scripts/valu_rate.hip) and stored negated (see TBP_*).
// The update is done in place: Hup / Eup enter as the cell above and leave as this cell's H / E, Fl enters as the left
// cell's F and leaves as this one's, dnext receives the old Hup (the next column's diagonal).
#define MP_DP_CELL_CORE \
        "v_sub_u32 %[eo], %[Hup], %[k3]\n\t"                      /* open a deletion: H - 3   (same column) */ \
        "v_mov_b32 %[dn], %[Hup]\n\t" \
        "v_sub_u32 %[Eup], %[Eup], %[k1]\n\t"                     /* extend it:       E - 1 */ \
        "v_pk_max_u16 %[Eup], %[eo], %[Eup]\n\t"                  /* E */ \
        "v_sub_u32 %[fo], %[Hl], %[k4]\n\t"                       /* open an insertion: H - 3, one column to the right: -4 in this frame */ \
        "v_sub_u32 %[Fl], %[Fl], %[k2]\n\t"                       /* extend it:         F - 1                         : -2 */ \
        "v_pk_max_u16 %[Fl], %[fo], %[Fl]\n\t"                    /* F */ \
        "v_xor_b32 %[t1], %[base], %[rb]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k3]\n\t"                    /* 0 match / 3 mismatch */ \
        "v_sub_u32 %[dg], %[diag], %[t1]\n\t"                     /* +1 / -2 along the diagonal (the +1 is the column term) */ \
        "v_pk_max_u16 %[t1], %[dg], %[Eup]\n\t" \
        "v_pk_max_u16 %[t1], %[t1], %[Fl]\n\t"                    /* H before the left-clip floor */ \
        "v_pk_max_u16 %[Hup], %[t1], %[floorv]\n\t"               /* H */ \
        "v_xor_b32 %[bits], %[Hup], %[t1]\n\t" \
        "v_pk_min_u16 %[bits], %[bits], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[bits], %[wz], %[fw]\n\t"           /* Z */ \
        "v_xor_b32 %[t1], %[Hup], %[dg]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[wd], %[fw]\n\t"             /* nD */ \
        "v_xor_b32 %[t1], %[Hup], %[Eup]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[we], %[fw]\n\t"             /* nE */ \
        "v_xor_b32 %[t1], %[Eup], %[eo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[weo], %[fw]\n\t"            /* nEO */ \
        "v_xor_b32 %[t1], %[Fl], %[fo]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k1]\n\t" \
        "v_pk_mad_i16 %[fw], %[t1], %[wfo], %[fw]\n\t"            /* nFO */ \
        "v_pk_min_u16 %[t1], %[Hup], %[elig]\n\t"                 /* eligible H (true -1 in columns that cannot end an alignment) */

// the same recurrences without the traceback bits: a score-only pass decides cheaply which instances can pass their
// cutoff at all (most rescue instances cannot); only those are run again with the bits
#define MP_DP_CELL_SCORE \
        "v_sub_u32 %[eo], %[Hup], %[k3]\n\t" \
        "v_mov_b32 %[dn], %[Hup]\n\t" \
        "v_sub_u32 %[Eup], %[Eup], %[k1]\n\t" \
        "v_pk_max_u16 %[Eup], %[eo], %[Eup]\n\t" \
        "v_sub_u32 %[fo], %[Hl], %[k4]\n\t" \
        "v_sub_u32 %[Fl], %[Fl], %[k2]\n\t" \
        "v_pk_max_u16 %[Fl], %[fo], %[Fl]\n\t" \
        "v_xor_b32 %[t1], %[base], %[rb]\n\t" \
        "v_pk_min_u16 %[t1], %[t1], %[k3]\n\t" \
        "v_sub_u32 %[dg], %[diag], %[t1]\n\t" \
        "v_pk_max_u16 %[t1], %[dg], %[Eup]\n\t" \
        "v_pk_max_u16 %[t1], %[t1], %[Fl]\n\t" \
        "v_pk_max_u16 %[Hup], %[t1], %[floorv]\n\t"

// MODE 0: traceback bits + row code (text output); 1: traceback bits + exact first-maximum tracking with tie counts
// (BAM output, mp_sgdp_batch); 2: scores only
template <int MODE>
__device__ __forceinline__ void dp_cell(const dp_consts &K, const dp_weights &Wt, v2s &Hup, v2s &Eup, v2s Hl, v2s &Fl, v2s diag, v2s base, v2s rb,
                                        v2s floorv, v2s elig, v2s gate, v2s colk, v2s jadd, v2s &dnext, v2s &fw, v2s &best, v2s &bpos,
                                        v2s &pos, v2s &cnt, v2s &rowcode)
{
    v2s eo, fo, dg, t1, bits;
    if (MODE == 2) {
        // score-only filter: `rowcode` is this COLUMN's running maximum of H over all rows (one frame per column).  Rows past
        // the window and columns that cannot end an alignment are included: the filter may over-estimate (the instance is
        // then filled in full and decided exactly), it must never under-estimate.
        asm(MP_DP_CELL_SCORE
            "v_pk_max_u16 %[rowcode], %[rowcode], %[Hup]"
            : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [dn] "=&v"(dnext), [Hup] "+&v"(Hup), [Eup] "+&v"(Eup),
              [Fl] "+&v"(Fl), [rowcode] "+&v"(rowcode)
            : [Hl] "v"(Hl), [diag] "v"(diag), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv), [k1] "v"(K.k1), [k2] "v"(K.k2),
              [k3] "v"(K.k3), [k4] "v"(K.k4));
    } else if (MODE == 0) {
        // ((H + bias) << 4) | (15 - cc): one max per cell keeps the row's best score and, among equals, its first column;
        // colk = 16 j + 15 - cc puts the column term back
        asm(MP_DP_CELL_CORE
            "v_pk_mad_i16 %[t1], %[t1], %[k16], %[colk]\n\t"
            "v_pk_max_u16 %[rowcode], %[rowcode], %[t1]"
            : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [dn] "=&v"(dnext), [Hup] "+&v"(Hup), [Eup] "+&v"(Eup),
              [Fl] "+&v"(Fl), [fw] "+&v"(fw), [rowcode] "+&v"(rowcode)
            : [Hl] "v"(Hl), [diag] "v"(diag), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv), [elig] "v"(elig), [k1] "v"(K.k1), [k2] "v"(K.k2),
              [k3] "v"(K.k3), [k4] "v"(K.k4), [k16] "v"(K.k16), [colk] "v"(colk),
              [wz] "v"(Wt.z), [wd] "v"(Wt.d), [we] "v"(Wt.e), [weo] "v"(Wt.eo), [wfo] "v"(Wt.fo));
    } else {
        v2s eq, ng;
        asm(MP_DP_CELL_CORE
            "v_add_u32 %[t1], %[t1], %[jadd]\n\t"                     // H + bias, comparable across columns
            "v_pk_min_u16 %[t1], %[t1], %[gate]\n\t"                  // rows past this instance's window: <= true 0
            "v_pk_max_u16 %[bits], %[t1], %[best]\n\t"
            "v_sub_u32 %[bits], %[bits], %[best]\n\t"
            "v_pk_min_u16 %[bits], %[bits], %[k1]\n\t"                // gt: strictly better than the best so far
            "v_pk_max_u16 %[best], %[best], %[t1]\n\t"
            "v_pk_sub_i16 %[dg], %[pos], %[bpos]\n\t"
            "v_pk_mad_i16 %[bpos], %[bits], %[dg], %[bpos]\n\t"
            "v_pk_add_i16 %[pos], %[pos], %[k1]\n\t"
            "v_sub_u32 %[eq], %[best], %[t1]\n\t"                     // cnt = gt ? 1 : cnt + (eligible H == best)
            "v_pk_min_u16 %[eq], %[eq], %[k1]\n\t"
            "v_pk_sub_i16 %[eq], %[k1], %[eq]\n\t"
            "v_pk_sub_i16 %[ng], %[k1], %[bits]\n\t"
            "v_pk_mad_i16 %[cnt], %[cnt], %[ng], %[eq]"
            : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [dn] "=&v"(dnext), [Hup] "+&v"(Hup), [Eup] "+&v"(Eup),
              [Fl] "+&v"(Fl), [fw] "+&v"(fw), [eq] "=&v"(eq), [ng] "=&v"(ng), [best] "+&v"(best), [bpos] "+&v"(bpos), [pos] "+&v"(pos), [cnt] "+&v"(cnt)
            : [Hl] "v"(Hl), [diag] "v"(diag), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv), [elig] "v"(elig), [k1] "v"(K.k1), [k2] "v"(K.k2),
              [k3] "v"(K.k3), [k4] "v"(K.k4), [gate] "v"(gate), [jadd] "v"(jadd),
              [wz] "v"(Wt.z), [wd] "v"(Wt.d), [we] "v"(Wt.e), [weo] "v"(Wt.eo), [wfo] "v"(Wt.fo));
    }
}

__device__ __forceinline__ v2s pk(short a, short b) { v2s r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2s pk1(short a) { v2s r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2u nz(v2s x) { return __builtin_elementwise_min(__builtin_bit_cast(v2u, x), (v2u){1, 1}); }   // x >= 0: (x != 0)
__device__ __forceinline__ v2s umax(v2s a, v2s b) { return __builtin_bit_cast(v2s, __builtin_elementwise_max(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b))); }
__device__ __forceinline__ v2s umin(v2s a, v2s b) { return __builtin_bit_cast(v2s, __builtin_elementwise_min(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b))); }
__device__ __forceinline__ int as_int(v2s v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ v2s as_v2s(int v) { return __builtin_bit_cast(v2s, v); }

// flag words per lane-step for C columns: 3 cells (15 bits) per 16-bit half
#define MP_DP_NW(C) (((C) + 2) / 3)
#define MP_DP_FLUSH 8          // steps of flag words staged per lane before one contiguous global write

template <int LPI, int C, int MODE>
__global__ void __launch_bounds__(MP_DP_BLOCK)
dp_fill_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t W,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, int clip_left, int clip_right,
               uint32_t *__restrict__ flags, uint32_t flag_rows /* rows of 32 x NW words reserved per instance PAIR */,
               mp_fill_result *__restrict__ out)
{
    constexpr int GROUPS = MP_DP_BLOCK / LPI;          // lane groups per block, two instances each
    constexpr int NW = MP_DP_NW(C);
    constexpr bool COUNT = MODE == 1, FLAGS = MODE != 2;
    __shared__ uint32_t s_win[GROUPS][2][MP_DP_MAX_WIN / 16 + 1];
    // per-lane staging of 8 steps of flag words (odd row stride: conflict-free ds_write_b32); a lane only ever reads its
    // own row, so no barrier is involved -- LDS is used as a transposing scratchpad
    __shared__ uint32_t s_fl[FLAGS ? GROUPS : 1][FLAGS ? LPI : 1][MP_DP_FLUSH * NW + 1];
    const int g = threadIdx.x / LPI;
    const int k = threadIdx.x % LPI;
    const uint32_t pair = blockIdx.x * GROUPS + g;
    int L[2], m[2], clipL[2], clipR[2], cutoff[2], strand[2];
    uint32_t rid[2];
    uint64_t wstart[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t inst = 2 * pair + q;
        mp_dp_task task; task.read_id = 0; task.strand = 1; task.win_start = 0; task.win_len = 0; task.reserved = 0;
        const bool live = inst < n_tasks;
        if (live) task = tasks[inst];
        L[q] = live ? (int)lens[task.read_id] : 0;
        m[q] = (int)task.win_len;
        if (m[q] > MP_DP_MAX_WIN) m[q] = MP_DP_MAX_WIN;
        if (L[q] > LPI * C) m[q] = 0;                  // host guarantees the template covers the longest read
        strand[q] = (int)task.strand; rid[q] = task.read_id; wstart[q] = task.win_start;
        clipL[q] = task.strand == 1 ? clip_left : clip_right;      // soft clips follow the strand (:2908-2911)
        clipR[q] = task.strand == 1 ? clip_right : clip_left;
        cutoff[q] = mp_dp_cutoff(L[q]);
        for (int w = k; w * 16 < m[q]; w += LPI) {     // stage the window: word w = bases 16w..16w+15, LSB-first
            uint64_t p = task.win_start + 16ull * (uint64_t)w;
            uint32_t sh = 2 * (uint32_t)(p & 15);
            uint64_t two = (uint64_t)text[p >> 4];
            if (sh) two |= (uint64_t)text[(p >> 4) + 1] << 32;
            s_win[g][q][w] = (uint32_t)(two >> sh);
        }
    }
    // per-column constants and row-0 boundary, both instances packed
    v2s rb[C], Hup[C], Eup[C], floorv[C], elig[C];
    constexpr int B = MP_DP_BIAS;
    // base codes 0, 3, 5, 6 (XOR of two of them is 0 or >= 3), looked up in a nibble table
#define MP_DP_CODE(b) ((int)((0x6530u >> (4u * (uint32_t)(b))) & 15u))
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
        const int j = k * C + cc + 1;
        short b2[2], h2[2], fl2[2], el2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            int b = 9 + q;                              // pad: never matches a window base or a window pad
            if (j <= L[q]) {
                const uint32_t *rd = reads + (uint64_t)rid[q] * W;
                int p = strand[q] == 1 ? j - 1 : L[q] - j;
                b = (int)((rd[p >> 4] >> (2 * (p & 15))) & 3u);
                if (strand[q] != 1) b = 3 - b;
                b = MP_DP_CODE(b);
            }
            b2[q] = (short)b;
            h2[q] = (short)((j <= clipL[q] ? 0 : MP_DP_OPEN + (j - clipL[q] - 1) * MP_DP_EXT) + B - j);   // CPU_DP.cpp:394-432
            fl2[q] = (short)(j <= clipL[q] ? B - j : 0);                                       // floor inside the left clip (:505-510)
            int mc = L[q] - clipR[q]; if (mc < 0) mc = 0;
            el2[q] = (short)((j >= mc && j <= L[q]) ? -1 /* 0xFFFF */ : B - 1 - j);            // best-cell columns (:538-591)
        }
        rb[cc] = pk(b2[0], b2[1]); Hup[cc] = pk(h2[0], h2[1]); Eup[cc] = pk1(MP_DP_SENT);
        floorv[cc] = pk(fl2[0], fl2[1]); elig[cc] = pk(el2[0], el2[1]);
    }
    v2s diag_carry;
    {
        const int j0 = k * C;
        diag_carry = pk((short)((j0 <= clipL[0] ? 0 : MP_DP_OPEN + (j0 - clipL[0] - 1) * MP_DP_EXT) + B - j0),
                        (short)((j0 <= clipL[1] ? 0 : MP_DP_OPEN + (j0 - clipL[1] - 1) * MP_DP_EXT) + B - j0));
    }
    __syncthreads();

    dp_consts K;
    K.k1 = pk1(1); K.k2 = pk1(2); K.k3 = pk1(3); K.k4 = pk1(4); K.k16 = pk1(16);
    dp_weights Wt[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        Wt[q].z = pk1((short)(TBP_Z << (5 * q))); Wt[q].d = pk1((short)(TBP_ND << (5 * q))); Wt[q].e = pk1((short)(TBP_NE << (5 * q)));
        Wt[q].eo = pk1((short)(TBP_NEO << (5 * q))); Wt[q].fo = pk1((short)(TBP_NFO << (5 * q)));
    }
    const v2s kZero = pk1(0);
    // best so far: COUNT: score + bias; otherwise the row code ((score + bias) << 4 | 15 - cc) -- both start at score 0
    v2s best = COUNT ? pk1((short)B) : pk1((short)(B << 4)), bpos = kZero;
    v2s cnt = kZero;
    v2s colmax[MODE == 2 ? C : 1];                     // score-only pass: running maximum of every column
#pragma unroll
    for (int cc = 0; cc < (MODE == 2 ? C : 1); cc++) colmax[cc] = kZero;
    v2s outH = pk1((short)B), outF = pk1(MP_DP_SENT);
    int outB = 0;
    // slab of one instance pair: [lane][step][word] -- the traceback walks a lane's steps, 8 of them per 64-byte line
    uint32_t *frow = flags + (uint64_t)pair * flag_rows * (LPI * NW) + (uint64_t)k * flag_rows * NW;
    const int mmax = m[0] > m[1] ? m[0] : m[1];
    int steps = mmax > 0 ? mmax + LPI - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(steps, o); steps = v > steps ? v : steps; }
    const v2s mrow = pk((short)m[0], (short)m[1]);
    uint32_t wq0 = 0, wq1 = 0;                     // 16 window bases of each instance, consumed two bits per step

    for (int t = 1; t <= steps; t++) {
        int inHi = wave_shr1(as_int(outH));
        int inFi = wave_shr1(as_int(outF));
        int inB = wave_shr1(outB);
        // the first lane of a group handles row t: its window bases come from LDS, one word per 16 steps (uniform in t)
        if (((t - 1) & 15) == 0) { wq0 = s_win[g][0][(t - 1) >> 4]; wq1 = s_win[g][1][(t - 1) >> 4]; }
        const int r0 = t <= m[0] ? MP_DP_CODE(wq0 & 3u) : 12, r1 = t <= m[1] ? MP_DP_CODE(wq1 & 3u) : 15;   // pads match nothing
        wq0 >>= 2; wq1 >>= 2;
        if (k == 0) { inHi = as_int(pk1((short)B)); inFi = as_int(pk1(MP_DP_SENT)); inB = r0 | (r1 << 16); }   // column 0: H = 0, no insertion state
        const int i = t - k;                       // row handled now
        if (i >= 1 && i <= mmax) {
            const v2s base = as_v2s(inB);
            // rows past an instance's own window must not count: no cap while i <= m, afterwards score 0 at most (best
            // starts at score 0 and only a strictly greater value replaces it)
            const v2s g2 = vmax(vmin(kZero, mrow - pk1((short)i)), pk1(-1));            // 0 in range, -1 past the window
            const v2s gatecap = pk1(-1) - g2 * pk1((short)(B + 1));                     // 0xFFFF in range, bias (score 0) past it
            const v2s gatecode = (g2 + pk1(1)) * pk1(-1);                               // 0xFFFF in range, 0 past the window
            v2s Hl = as_v2s(inHi), Fl = as_v2s(inFi), diag = diag_carry;
            v2s pos = pk1((short)(i * 16));
            v2s rowcode = kZero;
            v2s fw[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) fw[w] = kZero;
#pragma unroll
            for (int cc = 0; cc < C; cc++) {
                v2s dnext;
                const int j = k * C + cc + 1;
                dp_cell<MODE>(K, Wt[cc % 3], Hup[cc], Eup[cc], Hl, Fl, diag, base, rb[cc], floorv[cc], elig[cc], gatecap,
                               pk1((short)(16 * j + 15 - cc)), pk1((short)j), dnext, fw[cc / 3], best, bpos, pos, cnt, MODE == 2 ? colmax[cc] : rowcode);
                diag = dnext;
                Hl = Hup[cc];
            }
            if (MODE == 0) {
                // once per row: does this row beat the best so far?  (strictly: code > best code with its column bits saturated)
                const v2s rc = umin(rowcode, gatecode);                        // rows past the instance's window never win
                const v2s b15 = best | pk1(15);
                const v2s gt = __builtin_bit_cast(v2s, nz(umax(rc, b15) - b15));
                best += gt * (rc - best);
                bpos += gt * (pk1((short)i) - bpos);
            }
            diag_carry = as_v2s(inHi);
            outH = Hl; outF = Fl;
            if (FLAGS) {
#pragma unroll
                for (int w = 0; w < NW; w++) s_fl[g][k][((t - 1) & (MP_DP_FLUSH - 1)) * NW + w] = __builtin_bit_cast(uint32_t, fw[w]);
            }
        }
        outB = inB;
        if (FLAGS && mmax > 0 && ((t & (MP_DP_FLUSH - 1)) == 0 || t == steps)) {
            // flush this lane's last (up to) 8 steps as one contiguous run.  `steps` is the maximum over the whole wavefront (the
            // DPP hand-over needs every lane in the loop), so lane groups without an instance -- the tail of a launch -- come
            // through here too: they own no slot of the slab (it holds the launch's instance pairs, not the wavefront's) and
            // must not write
            uint32_t *dst = frow + (uint64_t)((t - 1) & ~(MP_DP_FLUSH - 1)) * NW;
#pragma unroll
            for (int w = 0; w < MP_DP_FLUSH * NW; w += 4) {
                uint4 v;
                v.x = s_fl[g][k][w]; v.y = s_fl[g][k][w + 1]; v.z = s_fl[g][k][w + 2]; v.w = s_fl[g][k][w + 3];
                *reinterpret_cast<uint4 *>(dst + w) = v;
            }
        }
    }

    // ---- per instance: reduce over the 32 lanes -- max score, then smallest row, then smallest column; ties counted
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int b, brow, bcol;
        if (MODE == 2) {
            b = 0; brow = 0; bcol = 0;
#pragma unroll
            for (int cc = 0; cc < C; cc++) {             // back to the column-free frame
                const int v = (int)(unsigned short)(q == 0 ? colmax[cc].x : colmax[cc].y) + (k * C + cc + 1) - MP_DP_BIAS;
                b = v > b ? v : b;
            }
        } else if (COUNT) {
            const int pos = q == 0 ? bpos.x : bpos.y;
            b = (int)(unsigned short)(q == 0 ? best.x : best.y) - MP_DP_BIAS; brow = pos >> 4; bcol = k * C + (pos & 15) + 1;
        } else {
            const int code = (int)(unsigned short)(q == 0 ? best.x : best.y);      // ((score + bias) << 4) | (15 - cc)
            b = (code >> 4) - MP_DP_BIAS; brow = q == 0 ? bpos.x : bpos.y; bcol = k * C + (15 - (code & 15)) + 1;
        }
        unsigned long long key = b >= cutoff[q] && b > 0 ? ((unsigned long long)(uint32_t)b << 32) | ((uint32_t)(0xFFFF - brow) << 16) | (uint32_t)(0xFFFF - bcol) : 0ull;
        unsigned long long kmax = key;
        for (int o = LPI / 2; o > 0; o >>= 1) { unsigned long long v = __shfl_xor(kmax, o, LPI); kmax = v > kmax ? v : kmax; }
        const int gbest = (int)(kmax >> 32);
        int c2 = 0;
        if (COUNT) {
            c2 = (b == gbest && gbest > 0) ? (int)(unsigned short)(q == 0 ? cnt.x : cnt.y) : 0;
            for (int o = LPI / 2; o > 0; o >>= 1) c2 += __shfl_xor(c2, o, LPI);
        }
        const uint32_t inst = 2 * pair + q;
        if (inst < n_tasks && k == 0) {
            mp_fill_result r;
            r.score = gbest;
            r.row = (uint16_t)(0xFFFF - ((kmax >> 16) & 0xFFFF));
            r.col = (uint16_t)(0xFFFF - (kmax & 0xFFFF));
            r.count = c2 > 255 ? 255u : (uint32_t)c2;
            if (gbest == 0) { r.row = 0; r.col = 0; r.count = 0; }
            out[inst] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ traceback
// BW > 0: the slab was written by dp_band_kernel<BW, GB> (kernels_dp_band.hip): [block of 64 pairs][column][word][lane]
// ------------------------------------------------------------------------------------------------ score-only filter
// dp_score_kernel<LPI, C>: the pre-pass of run_dp_filtered for reads of up to LPI x C bases.  It answers one question --
// can this instance reach its cutoff? -- and may err on the high side only (an instance it lets through is filled in full
// and decided exactly).  That freedom buys a leaner kernel than dp_fill_kernel<.., 2>:
//  * cell values are held relative to the ROW (x + MP_DP_BIAS - i) instead of the column: the same cancellation of the
//    match bonus along the diagonal, but now every cell of a row shares the frame, so the left-clip floor is one
//    register per row and the maximum is one running register -- no per-column floor / eligibility arrays, which lets a lane
//    own 19 or 20 columns (8 lanes per instance pair) and halves the per-row work per cell;
//  * the mismatch penalty of both packed instances is ONE v_perm_b32: per column a 4-byte table per instance (penalty for each of
//    the four window bases), the row's selector (byte 0: base of instance 0, byte 2: 4 + base of instance 1, travelling from lane
//    to lane with the row) picks both -- instead of an XOR of base codes and a packed min (13 instructions per two cells);
//  * the floor is applied in every column and row 0 is all zeros (as if the whole read could be clipped on the left),
//    rows past the shorter window of a pair and columns that cannot end an alignment are not excluded: all of these can
//    only raise scores.
#define MP_DP_SCORE_CELL \
        "v_sub_u32 %[eo], %[Hup], %[k4]\n\t"                      /* open a deletion: H - 3, one row down: -4 in this frame */ \
        "v_mov_b32 %[dn], %[Hup]\n\t" \
        "v_sub_u32 %[Eup], %[Eup], %[k2]\n\t"                     /* extend it:       E - 1                : -2 */ \
        "v_pk_max_u16 %[Eup], %[eo], %[Eup]\n\t" \
        "v_sub_u32 %[fo], %[Hl], %[k3]\n\t"                       /* open an insertion: H - 3 (same row) */ \
        "v_sub_u32 %[Fl], %[Fl], %[k1]\n\t" \
        "v_pk_max_u16 %[Fl], %[fo], %[Fl]\n\t" \
        "v_perm_b32 %[t1], %[tb], %[ta], %[base]\n\t"             /* mismatch penalty of both instances: one byte look-up each */ \
        "v_sub_u32 %[dg], %[diag], %[t1]\n\t"                     /* the row term swallows the +1 of a match */ \
        "v_pk_max_u16 %[t1], %[dg], %[Eup]\n\t" \
        "v_pk_max_u16 %[t1], %[t1], %[Fl]\n\t" \
        "v_pk_max_u16 %[Hup], %[t1], %[floorr]\n\t" \
        "v_pk_max_u16 %[rowmax], %[rowmax], %[Hup]"

template <int LPI, int C>
__global__ void __launch_bounds__(MP_DP_BLOCK)
dp_score_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t W,
                const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, mp_fill_result *__restrict__ out,
                const uint32_t *__restrict__ list = nullptr /* slot s of the launch stands for instance list[s] (a length class of a mixed batch) */)
{
    constexpr int GROUPS = MP_DP_BLOCK / LPI;
    constexpr int B = MP_DP_BIAS;
    __shared__ uint32_t s_win[GROUPS][2][MP_DP_MAX_WIN / 16 + 1];
    const int g = threadIdx.x / LPI;
    const int k = threadIdx.x % LPI;
    const uint32_t pair = blockIdx.x * GROUPS + g;
    int L[2], m[2], strand[2];
    uint32_t rid[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint32_t slot = 2 * pair + q;
        mp_dp_task task; task.read_id = 0; task.strand = 1; task.win_start = 0; task.win_len = 0; task.reserved = 0;
        const bool live = slot < n_tasks;
        if (live) task = tasks[list ? list[slot] : slot];
        L[q] = live ? (int)lens[task.read_id] : 0;
        m[q] = (int)task.win_len;
        if (m[q] > MP_DP_MAX_WIN) m[q] = MP_DP_MAX_WIN;
        if (L[q] > LPI * C) m[q] = 0;                  // host guarantees the template covers the longest read
        strand[q] = (int)task.strand; rid[q] = task.read_id;
        for (int w = k; w * 16 < m[q]; w += LPI) {
            uint64_t p = task.win_start + 16ull * (uint64_t)w;
            uint32_t sh = 2 * (uint32_t)(p & 15);
            uint64_t two = (uint64_t)text[p >> 4];
            if (sh) two |= (uint64_t)text[(p >> 4) + 1] << 32;
            s_win[g][q][w] = (uint32_t)(two >> sh);
        }
    }
    // ta / tb[cc]: per column the penalty (0 match, 3 mismatch) of instance 0 / 1 for each of the four window bases, one byte each:
    // v_perm_b32 picks both with the row's selector (byte 0: base of instance 0, byte 2: 4 + base of instance 1, 0x0D = pad row: 255)
    uint32_t ta[C], tb[C];
    v2s Hup[C], Eup[C];
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
        const int j = k * C + cc + 1;
        uint32_t tab[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            tab[q] = 0x03030303u;                              // a column past the read's end matches nothing
            if (j <= L[q]) {
                const uint32_t *rd = reads + (uint64_t)rid[q] * W;
                int p = strand[q] == 1 ? j - 1 : L[q] - j;
                int b = (int)((rd[p >> 4] >> (2 * (p & 15))) & 3u);
                if (strand[q] != 1) b = 3 - b;
                tab[q] &= ~(0xFFu << (8 * b));
            }
        }
        ta[cc] = tab[0]; tb[cc] = tab[1]; Hup[cc] = pk1((short)B); Eup[cc] = pk1(MP_DP_SENT);
    }
    v2s diag_carry = pk1((short)B);
    __syncthreads();
    const v2s k1 = pk1(1), k2 = pk1(2), k3 = pk1(3), k4 = pk1(4);
    v2s best = pk1((short)B);                          // score 0
    v2s outH = pk1((short)B), outF = pk1(MP_DP_SENT);
    int outB = 0;
    const int mmax = m[0] > m[1] ? m[0] : m[1];
    int steps = mmax > 0 ? mmax + LPI - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { int v = __shfl_xor(steps, o); steps = v > steps ? v : steps; }
    uint32_t wq0 = 0, wq1 = 0;
    for (int t = 1; t <= steps; t++) {
        int inHi = wave_shr1(as_int(outH));
        int inFi = wave_shr1(as_int(outF));
        int inB = wave_shr1(outB);
        if (((t - 1) & 15) == 0) { wq0 = s_win[g][0][(t - 1) >> 4]; wq1 = s_win[g][1][(t - 1) >> 4]; }
        const int r0 = t <= m[0] ? (int)(wq0 & 3u) : 0x0D, r1 = t <= m[1] ? 4 + (int)(wq1 & 3u) : 0x0D;   // pad rows: penalty 255
        wq0 >>= 2; wq1 >>= 2;
        const int i = t - k;                           // row handled now
        if (k == 0) { inHi = as_int(pk1((short)(B - i))); inFi = as_int(pk1(MP_DP_SENT)); inB = r0 | (r1 << 16) | 0x0C000C00; }   // column 0: H = 0
        if (i >= 1 && i <= mmax) {
            const v2s base = as_v2s(inB);
            const v2s floorr = pk1((short)(B - i));    // score 0 in this row's frame
            v2s Hl = as_v2s(inHi), Fl = as_v2s(inFi), diag = diag_carry, rowmax = pk1(0);
#pragma unroll
            for (int cc = 0; cc < C; cc++) {
                v2s eo, fo, dg, t1, dnext;
                asm(MP_DP_SCORE_CELL
                    : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [dn] "=&v"(dnext), [Hup] "+&v"(Hup[cc]), [Eup] "+&v"(Eup[cc]),
                      [Fl] "+&v"(Fl), [rowmax] "+&v"(rowmax)
                    : [Hl] "v"(Hl), [diag] "v"(diag), [base] "v"(base), [ta] "v"(ta[cc]), [tb] "v"(tb[cc]), [floorr] "v"(floorr), [k1] "v"(k1), [k2] "v"(k2),
                      [k3] "v"(k3), [k4] "v"(k4));
                diag = dnext;
                Hl = Hup[cc];
            }
            best = umax(best, rowmax + pk1((short)i));     // back to the row-free frame
            diag_carry = as_v2s(inHi);
            outH = Hl; outF = Fl;
        }
        outB = inB;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        int b = (int)(unsigned short)(q == 0 ? best.x : best.y) - B;
        for (int o = LPI / 2; o > 0; o >>= 1) { const int v = __shfl_xor(b, o, LPI); b = v > b ? v : b; }
        const uint32_t slot = 2 * pair + q;
        if (slot < n_tasks && k == 0) {
            mp_fill_result r;
            r.score = b; r.row = 0; r.col = 0; r.count = 0;
            out[list ? list[slot] : slot] = r;
        }
    }
}

// CENTER: the band is the one of dp_band_kernel's second pass, centred on the diagonal of the best cell
template <int LPI, int C, int BW = 0, int GB = 0, bool CENTER = false>
__global__ void __launch_bounds__(64)
dp_trace_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ lens, uint32_t n_tasks, int clip_left, int clip_right,
                const uint32_t *__restrict__ flags, uint32_t flag_rows, const mp_fill_result *__restrict__ fill,
                mp_dp_result *__restrict__ out, uint8_t *__restrict__ ops, uint32_t ops_stride, const uint32_t *__restrict__ list = nullptr)
{
    constexpr int NW = MP_DP_NW(C);
    // `slot` is the position in the launch (and in the slab); with a list, slot s stands for instance list[s]
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tasks) return;
    const uint32_t inst = list ? list[slot] : slot;
    const mp_dp_task task = tasks[inst];
    const mp_fill_result fr = fill[inst];
    const int L = (int)lens[task.read_id];
    const int clipL = task.strand == 1 ? clip_left : clip_right;
    const int cutoff = mp_dp_cutoff(L);
    mp_dp_result res; res.score = 0; res.hit_loc = 0; res.max_score_count = 0; res.n_ops = 0;
    if (fr.score < cutoff || fr.score <= 0 || task.win_len == 0) { out[inst] = res; return; }
    res.max_score_count = fr.count;
    constexpr int NWB = (BW + 2) / 3;
    const uint32_t *fpair = BW > 0 ? flags + (uint64_t)(slot >> 7) * flag_rows * (NWB * 64) + ((slot >> 1) & 63u)
                                   : flags + (uint64_t)(slot >> 1) * flag_rows * (LPI * NW);
    const int half = (int)(slot & 1u) * 16;
    uint8_t *op = ops ? ops + (uint64_t)inst * ops_stride : nullptr;
    uint32_t n = 0;
    auto emit = [&](uint8_t ch, int rep) { for (int q = 0; q < rep; q++) { if (op && n < ops_stride) op[n] = ch; n++; } };
    // traceback bits of (row r, column c) as TB_* (positive logic)
    auto cell = [&](int r, int c) -> uint32_t {
        if (c == 0) return TB_Z;
        if (r == 0) return c <= clipL ? TB_Z : 0u;
        if (BW > 0) {
            const int d = r - c + GB - (CENTER ? (int)fr.row - (int)fr.col : 0);
            if (d < 0 || d >= BW) return 0u;          // never reached for an instance the band kernel qualified
            const uint32_t w = fpair[((uint64_t)(c - 1) * NWB + (uint32_t)(d / 3)) * 64];
            // the band kernel shifts its bits in: first cell of a word highest, within a cell Z, nD, nE, nEO, nFO from the top
            const uint32_t b = (w >> (half + 5 * (2 - d % 3))) & 31u;
            return ((b >> 4) & 1u) | ((~b & 8u) >> 2) | (~b & 4u) | ((~b & 2u) << 2) | ((~b & 1u) << 4);
        }
        const int lane = (c - 1) / C, cc = (c - 1) % C;
        const uint32_t w = fpair[((uint64_t)lane * flag_rows + (uint32_t)(r + lane - 1)) * NW + cc / 3];
        const uint32_t b = (w >> (half + 5 * (cc % 3))) & 31u;
        return (b & TBP_Z) | ((b ^ 30u) & 30u);       // invert nD, nE, nEO, nFO
    };
    bool valid = true;
    int hit = 0;
    if (L < cutoff) valid = false;                                    // CPU_DP.cpp:830-831
    else {
        int c = fr.col, r = fr.row;
        emit('S', L - c);
        enum { NORMAL, I_EXT, D_EXT } state = NORMAL;
        bool ended = false, sm_exit = false;
        // the cell looked at for the exit test of a diagonal step is the cell the next step starts from: it is kept (one flag word
        // per step instead of two -- the walk is a chain of dependent loads)
        uint32_t kept = 0;
        bool have_kept = false;
        while (c > 0 && r > 0) {
            const uint32_t b = have_kept ? kept : cell(r, c);
            have_kept = false;
            if (state == NORMAL) {
                uint32_t pred = 0;
                if ((b & TB_D) && c != 1) pred = cell(r - 1, c - 1);
                if ((b & TB_D) && c != 1 && (pred & TB_Z)) { sm_exit = true; break; }   // DP_BT_SM_EXIT
                else if (b & TB_D) { emit('X', 1); r--; c--; if (c >= 1) { kept = pred; have_kept = true; } }
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

// flag[t] = 1 for the instances whose read has lo < length <= hi (length classes of a mixed batch: the score-only pre-pass
// runs a kernel of matching width over each class)
// `skip` (may be null): instances already decided (dp_thermo_kernel's certificates) belong to no class
__global__ void dp_len_class_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ lens, uint32_t n, uint32_t lo, uint32_t hi, uint32_t *__restrict__ flag,
                                    const uint8_t *__restrict__ skip)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t L = lens[tasks[t].read_id];
    flag[t] = (L > lo && L <= hi && !(skip && skip[t])) ? 1u : 0u;
}

// flag[t] = 1 for the task slots that hold an instance (mate rescue reserves four slots per read and fills those it has a hit for)
__global__ void dp_nonempty_flag_kernel(const mp_dp_task *__restrict__ tasks, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flag[t] = tasks[t].win_len != 0 ? 1u : 0u;
}

// ---- score-only pre-pass bookkeeping: which instances can pass, compaction of their tasks, scatter of their results
__global__ void dp_select_kernel(const mp_dp_task *__restrict__ tasks, const mp_fill_result *__restrict__ fill, const uint32_t *__restrict__ lens,
                                 uint32_t n, uint32_t *__restrict__ flag, mp_dp_result *__restrict__ res)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_fill_result fr = fill[t];
    const int cutoff = mp_dp_cutoff((int)lens[tasks[t].read_id]);
    const bool pass = fr.score >= cutoff && fr.score > 0 && tasks[t].win_len != 0;
    flag[t] = pass ? 1u : 0u;
    mp_dp_result z; z.score = 0; z.hit_loc = 0; z.max_score_count = 0; z.n_ops = 0;
    res[t] = z;                                          // what the traceback kernel writes for an instance below its cutoff
}

__global__ void dp_compact_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ idx,
                                  uint32_t n, mp_dp_task *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && flag[t]) out[idx[t] - 1] = tasks[t];
}

__global__ void dp_scatter_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ idx, const mp_dp_result *__restrict__ sel,
                                  uint32_t n, mp_dp_result *__restrict__ res)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && flag[t]) res[t] = sel[idx[t] - 1];
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

// DP2CPUAlgnThread result assembly (DV-DPfunctions.cpp:3432-3530), positions and scores only.  Two passes over the candidates so
// that the hit list leaves the device GROUPED BY PAIR in ascending pair order (the host then only orders the few hits inside a
// group): COUNT adds up each pair's hits, an exclusive scan turns the counts into group offsets, the second pass stores every hit
// at its group's offset plus a per-pair cursor.
// The cursor is NOT an atomic counter: equal hits of a pair (same positions and scores, different tie counts or windows) are de-duplicated
// on the host by "first one wins", so their order must not depend on which wavefront stored first.  A pair's candidates are contiguous in
// `cands` (pair_kernel emits them from one lane), so the cursor of candidate t is the number of passing candidates of its pair in front
// of it: hit_idx (exclusive scan of the COUNT pass's flags) minus its value at the pair's first candidate.
__global__ void cand_first_kernel(const mp_candidate *__restrict__ cands, uint32_t n, uint32_t *__restrict__ cand_first)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t p = cands[t].read_id_left >> 1;
    if (t == 0 || (cands[t - 1].read_id_left >> 1) != p) cand_first[p] = t;
}

template <bool COUNT>
__global__ void collect_pair_hits_kernel(const mp_candidate *__restrict__ cands, uint32_t n, const uint32_t *__restrict__ lens,
                                         const mp_dp_task *__restrict__ lt, const mp_dp_result *__restrict__ lr,
                                         const mp_dp_task *__restrict__ rt, const mp_dp_result *__restrict__ rr,
                                         mp_pair_hit *__restrict__ hits, uint32_t *__restrict__ pair_cnt, const uint32_t *__restrict__ pair_off, uint32_t cap,
                                         uint32_t *__restrict__ pair_done, uint32_t *__restrict__ hit_flag, const uint32_t *__restrict__ hit_idx,
                                         const uint32_t *__restrict__ cand_first)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_candidate c = cands[t];
    const uint32_t left = c.read_id_left, right = left ^ 1u;
    const mp_dp_result a = lr[t], b = rr[t];
    const bool pass = a.score > 0 && b.score > 0 && a.score >= mp_dp_cutoff((int)lens[left]) && b.score >= mp_dp_cutoff((int)lens[right]);
    if (COUNT) { hit_flag[t] = pass ? 1u : 0u; if (pass) atomicAdd(&pair_cnt[left >> 1], 1u); return; }
    if (!pass) return;
    const uint32_t slot = pair_off[left >> 1] + (hit_idx[t] - hit_idx[cand_first[left >> 1]]);
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
    pair_done[h.pair_id] = 1u;
}
