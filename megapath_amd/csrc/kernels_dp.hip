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
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false); }

// constants the cell update needs in registers (VOP3P operands)
struct dp_consts { v2s open, ext, one, zero, neg3, k2, k4, k8, k16; };

// One DP cell of two instances in ONE asm statement (hipcc pads every separate asm with s_nop).
//   in : Hup/Eup (row above), Hl/Fl (cell to the left), diag, base (window bases), rb (read bases), floorv, elig
//   out: e, f, h for this cell; fw (flag accumulator); COUNT=false: rowcode = max over the row of (eligible H << 4 | 15-cc)
//        so that the first maximum of the row wins; COUNT=true: exact per-cell first-maximum tracking with tie counts.
// Decision bits are arithmetic: x != 0  <=>  min_u16(x, 1) (x taken as unsigned); differences are formed with v_xor_b32
// (full rate on gfx950; v_pk_* and v_max run at half rate -- scripts/valu_rate.hip) and stored negated (see TBP_*).
template <bool COUNT>
__device__ __forceinline__ void dp_cell(const dp_consts &K, v2s Hup, v2s Eup, v2s Hl, v2s Fl, v2s diag, v2s base, v2s rb, v2s floorv, v2s elig,
                                        v2s gate, v2s shiftk, v2s colk, v2s &e, v2s &f, v2s &h, v2s &fw, v2s &best, v2s &bpos, v2s &pos, v2s &cnt,
                                        v2s &rowcode)
{
    v2s eo, fo, dg, t1, bits;
    asm("v_pk_add_i16 %[eo], %[Hup], %[kopen]\n\t"
        "v_pk_add_i16 %[e], %[Eup], %[kext]\n\t"
        "v_pk_max_i16 %[e], %[eo], %[e]\n\t"
        "v_pk_add_i16 %[fo], %[Hl], %[kopen]\n\t"
        "v_pk_add_i16 %[f], %[Fl], %[kext]\n\t"
        "v_pk_max_i16 %[f], %[fo], %[f]\n\t"
        "v_xor_b32 %[t1], %[base], %[rb]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[kone]\n\t"
        "v_pk_mad_i16 %[t1], %[t1], %[kneg3], %[kone]\n\t"          // +1 match / -2 mismatch
        "v_pk_add_i16 %[dg], %[diag], %[t1]\n\t"
        "v_pk_max_i16 %[t1], %[dg], %[e]\n\t"
        "v_pk_max_i16 %[t1], %[t1], %[f]\n\t"                       // H before the left-clip floor
        "v_pk_max_i16 %[h], %[t1], %[floorv]\n\t"                   // H
        "v_xor_b32 %[bits], %[h], %[t1]\n\t"
        "v_pk_min_u16 %[bits], %[bits], %[kone]\n\t"                // Z
        "v_xor_b32 %[t1], %[h], %[dg]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[kone]\n\t"
        "v_pk_mad_i16 %[bits], %[t1], %[k2], %[bits]\n\t"           // nD
        "v_xor_b32 %[t1], %[h], %[e]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[kone]\n\t"
        "v_pk_mad_i16 %[bits], %[t1], %[k4], %[bits]\n\t"           // nE
        "v_xor_b32 %[t1], %[e], %[eo]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[kone]\n\t"
        "v_pk_mad_i16 %[bits], %[t1], %[k8], %[bits]\n\t"           // nEO
        "v_xor_b32 %[t1], %[f], %[fo]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[kone]\n\t"
        "v_pk_mad_i16 %[bits], %[t1], %[k16], %[bits]\n\t"          // nFO
        "v_pk_mad_i16 %[fw], %[bits], %[shiftk], %[fw]\n\t"
        "v_pk_min_i16 %[t1], %[h], %[elig]"                         // eligible H (<= -1 in columns that cannot end an alignment)
        : [eo] "=&v"(eo), [fo] "=&v"(fo), [dg] "=&v"(dg), [t1] "=&v"(t1), [bits] "=&v"(bits), [e] "=&v"(e), [f] "=&v"(f), [h] "=&v"(h),
          [fw] "+&v"(fw)
        : [Hup] "v"(Hup), [Eup] "v"(Eup), [Hl] "v"(Hl), [Fl] "v"(Fl), [diag] "v"(diag), [base] "v"(base), [rb] "v"(rb), [floorv] "v"(floorv),
          [elig] "v"(elig), [shiftk] "v"(shiftk), [kopen] "v"(K.open), [kext] "v"(K.ext), [kone] "v"(K.one),
          [kneg3] "v"(K.neg3), [k2] "v"(K.k2), [k4] "v"(K.k4), [k8] "v"(K.k8), [k16] "v"(K.k16));
    if (!COUNT) {
        // (H << 4) | (15 - cc): one max per cell keeps the row's best score and, among equals, its first column
        asm("v_pk_mad_i16 %[t1], %[t1], %[k16], %[colk]\n\t"
            "v_pk_max_i16 %[rowcode], %[rowcode], %[t1]"
            : [t1] "+&v"(t1), [rowcode] "+&v"(rowcode) : [k16] "v"(K.k16), [colk] "v"(colk));
    } else {
        v2s eq, ng;
        asm("v_pk_min_i16 %[t1], %[t1], %[gate]\n\t"                    // rows past this instance's window: <= 0
            "v_pk_sub_i16 %[bits], %[t1], %[best]\n\t"
            "v_pk_max_i16 %[bits], %[bits], %[kzero]\n\t"
            "v_pk_min_u16 %[bits], %[bits], %[kone]\n\t"                // gt: strictly better than the best so far
            "v_pk_max_i16 %[best], %[best], %[t1]\n\t"
            "v_pk_sub_i16 %[dg], %[pos], %[bpos]\n\t"
            "v_pk_mad_i16 %[bpos], %[bits], %[dg], %[bpos]\n\t"
            "v_pk_add_i16 %[pos], %[pos], %[kone]\n\t"
            "v_pk_sub_i16 %[eq], %[best], %[t1]\n\t"                    // cnt = gt ? 1 : cnt + (eligible H == best)
            "v_pk_min_u16 %[eq], %[eq], %[kone]\n\t"
            "v_pk_sub_i16 %[eq], %[kone], %[eq]\n\t"
            "v_pk_sub_i16 %[ng], %[kone], %[bits]\n\t"
            "v_pk_mad_i16 %[cnt], %[cnt], %[ng], %[eq]"
            : [t1] "+&v"(t1), [bits] "=&v"(bits), [dg] "=&v"(dg), [eq] "=&v"(eq), [ng] "=&v"(ng), [best] "+&v"(best), [bpos] "+&v"(bpos),
              [pos] "+&v"(pos), [cnt] "+&v"(cnt)
            : [gate] "v"(gate), [kzero] "v"(K.zero), [kone] "v"(K.one));
    }
}

__device__ __forceinline__ v2s pk(short a, short b) { v2s r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2s pk1(short a) { v2s r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2s vmax(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2u nz(v2s x) { return __builtin_elementwise_min(__builtin_bit_cast(v2u, x), (v2u){1, 1}); }   // x >= 0: (x != 0)
__device__ __forceinline__ int as_int(v2s v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ v2s as_v2s(int v) { return __builtin_bit_cast(v2s, v); }

// flag words per lane-step for C columns: 3 cells (15 bits) per 16-bit half
#define MP_DP_NW(C) (((C) + 2) / 3)
#define MP_DP_FLUSH 8          // steps of flag words staged per lane before one contiguous global write

template <int LPI, int C, bool COUNT>
__global__ void __launch_bounds__(MP_DP_BLOCK)
dp_fill_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t W,
               const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, int clip_left, int clip_right,
               uint32_t *__restrict__ flags, uint32_t flag_rows /* rows of 32 x NW words reserved per instance PAIR */,
               mp_fill_result *__restrict__ out)
{
    constexpr int GROUPS = MP_DP_BLOCK / LPI;          // lane groups per block, two instances each
    constexpr int NW = MP_DP_NW(C);
    __shared__ uint32_t s_win[GROUPS][2][MP_DP_MAX_WIN / 16 + 1];
    // per-lane staging of 8 steps of flag words (odd row stride: conflict-free ds_write_b32); a lane only ever reads its
    // own row, so no barrier is involved -- LDS is used as a transposing scratchpad
    __shared__ uint32_t s_fl[GROUPS][LPI][MP_DP_FLUSH * NW + 1];
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
#pragma unroll
    for (int cc = 0; cc < C; cc++) {
        const int j = k * C + cc + 1;
        short b2[2], h2[2], fl2[2], el2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            int b = 4 + q;                              // never matches a window base (0..3)
            if (j <= L[q]) {
                const uint32_t *rd = reads + (uint64_t)rid[q] * W;
                int p = strand[q] == 1 ? j - 1 : L[q] - j;
                b = (int)((rd[p >> 4] >> (2 * (p & 15))) & 3u);
                if (strand[q] != 1) b = 3 - b;
            }
            b2[q] = (short)b;
            h2[q] = (short)(j <= clipL[q] ? 0 : MP_DP_OPEN + (j - clipL[q] - 1) * MP_DP_EXT);   // CPU_DP.cpp:394-432
            fl2[q] = (short)(j <= clipL[q] ? 0 : -32768);                                      // floor inside the left clip (:505-510)
            int mc = L[q] - clipR[q]; if (mc < 0) mc = 0;
            el2[q] = (short)((j >= mc && j <= L[q]) ? 32767 : -1);                             // best-cell columns (:538-591)
        }
        rb[cc] = pk(b2[0], b2[1]); Hup[cc] = pk(h2[0], h2[1]); Eup[cc] = pk1(MP_DP_NEG16);
        floorv[cc] = pk(fl2[0], fl2[1]); elig[cc] = pk(el2[0], el2[1]);
    }
    v2s diag_carry;
    {
        const int j0 = k * C;
        diag_carry = pk((short)(j0 <= clipL[0] ? 0 : MP_DP_OPEN + (j0 - clipL[0] - 1) * MP_DP_EXT),
                        (short)(j0 <= clipL[1] ? 0 : MP_DP_OPEN + (j0 - clipL[1] - 1) * MP_DP_EXT));
    }
    __syncthreads();

    dp_consts K;
    K.open = pk1(MP_DP_OPEN); K.ext = pk1(MP_DP_EXT); K.one = pk1(1); K.zero = pk1(0); K.neg3 = pk1(-3);
    K.k2 = pk1(2); K.k4 = pk1(4); K.k8 = pk1(8); K.k16 = pk1(16);
    const v2s kShift[3] = {pk1(1), pk1(32), pk1(1024)};
    const v2s kZero = pk1(0);
    v2s best = kZero, bpos = kZero;
    v2s cnt = kZero;
    v2s outH = kZero, outF = pk1(MP_DP_NEG16);
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
        const int r0 = t <= m[0] ? (int)(wq0 & 3u) : 6, r1 = t <= m[1] ? (int)(wq1 & 3u) : 7;
        wq0 >>= 2; wq1 >>= 2;
        if (k == 0) { inHi = 0; inFi = as_int(pk1(MP_DP_NEG16)); inB = r0 | (r1 << 16); }   // column 0: H = 0, no insertion state
        const int i = t - k;                       // row handled now
        if (i >= 1 && i <= mmax) {
            const v2s base = as_v2s(inB);
            // rows past an instance's own window must not count: cap 32767 while i <= m, 0 afterwards (best starts at 0
            // and only a strictly greater value replaces it)
            const v2s g2 = vmax(vmin(kZero, mrow - pk1((short)i)), pk1(-1));            // 0 or -1
            const v2s gatecap = g2 * pk1(32767) + pk1(32767);
            const v2s gatecode = (g2 + pk1(1)) * pk1(32735);                            // large in range, 0 past the window
            v2s Hl = as_v2s(inHi), Fl = as_v2s(inFi), diag = diag_carry;
            v2s pos = pk1((short)(i * 16));
            v2s rowcode = pk1(-32768);
            v2s fw[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) fw[w] = kZero;
#pragma unroll
            for (int cc = 0; cc < C; cc++) {
                v2s e, f, h;
                dp_cell<COUNT>(K, Hup[cc], Eup[cc], Hl, Fl, diag, base, rb[cc], floorv[cc], elig[cc], gatecap, kShift[cc % 3], pk1((short)(15 - cc)),
                               e, f, h, fw[cc / 3], best, bpos, pos, cnt, rowcode);
                diag = Hup[cc];
                Hup[cc] = h; Eup[cc] = e; Hl = h; Fl = f;
            }
            if (!COUNT) {
                // once per row: does this row beat the best so far?  (strictly: code > best code with its column bits saturated)
                const v2s rc = vmin(rowcode, gatecode);                        // rows past the instance's window never win
                const v2s gt = __builtin_bit_cast(v2s, nz(vmax(rc - (best | pk1(15)), kZero)));
                best += gt * (rc - best);
                bpos += gt * (pk1((short)i) - bpos);
            }
            diag_carry = as_v2s(inHi);
            outH = Hl; outF = Fl;
#pragma unroll
            for (int w = 0; w < NW; w++) s_fl[g][k][((t - 1) & (MP_DP_FLUSH - 1)) * NW + w] = __builtin_bit_cast(uint32_t, fw[w]);
        }
        outB = inB;
        if ((t & (MP_DP_FLUSH - 1)) == 0 || t == steps) {
            // flush this lane's last (up to) 8 steps as one contiguous run
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
        if (COUNT) {
            const int pos = q == 0 ? bpos.x : bpos.y;
            b = q == 0 ? best.x : best.y; brow = pos >> 4; bcol = k * C + (pos & 15) + 1;
        } else {
            const int code = q == 0 ? best.x : best.y;      // (score << 4) | (15 - cc)
            b = code >> 4; brow = q == 0 ? bpos.x : bpos.y; bcol = k * C + (15 - (code & 15)) + 1;
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
template <int LPI, int C>
__global__ void __launch_bounds__(64)
dp_trace_kernel(const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ lens, uint32_t n_tasks, int clip_left, int clip_right,
                const uint32_t *__restrict__ flags, uint32_t flag_rows, const mp_fill_result *__restrict__ fill,
                mp_dp_result *__restrict__ out, uint8_t *__restrict__ ops, uint32_t ops_stride)
{
    constexpr int NW = MP_DP_NW(C);
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
    const uint32_t *fpair = flags + (uint64_t)(inst >> 1) * flag_rows * (LPI * NW);
    const int half = (int)(inst & 1u) * 16;
    uint8_t *op = ops ? ops + (uint64_t)inst * ops_stride : nullptr;
    uint32_t n = 0;
    auto emit = [&](uint8_t ch, int rep) { for (int q = 0; q < rep; q++) { if (op && n < ops_stride) op[n] = ch; n++; } };
    // traceback bits of (row r, column c) as TB_* (positive logic)
    auto cell = [&](int r, int c) -> uint32_t {
        if (c == 0) return TB_Z;
        if (r == 0) return c <= clipL ? TB_Z : 0u;
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
