// megapath_amd/csrc/kernels_seed_x2.hip -- product code (measurement variant): GENERATED from seed_kernel's text by scripts/make_seed_x2.py;
// see kernels_seed.hip for every comment.  Included by api.hip after kernels_seed.hip.
#pragma once

// seed_kernel with TWO (read, strand) items per lane: both items' loads are issued before the one wait of the trip, then both are consumed
// (review item 5a of round 2; measurement variant, MP_SEED_ITEMS=2).  The body of each phase is seed_kernel's, applied to item j of the lane.
__global__ void __launch_bounds__(256)
seed_kernel_x2(mp_dev_index ix, const uint32_t *__restrict__ scan, const uint32_t *__restrict__ lens, uint32_t n_items, uint32_t W, mp_dev_mmp P,
            mp_seed_rec *__restrict__ slabs, uint32_t *__restrict__ seeds_per_rs, uint32_t *__restrict__ hits_per_rs,
            uint32_t *__restrict__ overflow_rs, uint32_t *__restrict__ over_slot, uint32_t overflow_cap, unsigned int *queue, mp_counters *ctr,
            uint32_t slab_limit /* <= MP_SEED_SLAB; smaller only to exercise the overflow path */)
{
    extern __shared__ uint32_t s_scan[];                 // [256][S | 1]: the scan row of the item each lane works on
    const int lane = threadIdx.x & (MP_WAVE - 1);
    const uint32_t S = mp_scan_stride(W);
    uint32_t *const sl2[2] = {s_scan + (threadIdx.x * 2u) * (S | 1u), s_scan + (threadIdx.x * 2u + 1u) * (S | 1u)};
    const uint32_t n = ix.n;
    const bool pf = ix.kmer17 && P.seed_min_length >= (int)ix.kmer_k;
    const uint64_t kmask = (1ull << (2 * ix.kmer_k)) - 1;             // (kmer15x3 and the 16-mer tables exist only with 17-mers)
    const bool text_ok = ix.sa_dense && ix.sa_dense_shift == 0;
    const uint32_t cum0 = ix.cum[0], cum1 = ix.cum[1], cum2 = ix.cum[2], cum3 = ix.cum[3];
    mp_lane_events ev = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    mp_lane_state st[2];
    bool hv[2] = {false, false};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        mp_lane_state &s = st[j];
        s.rs = 0; s.len = 0; s.strand = 0; s.i = 0; s.seed_len = 0; s.l = 0; s.r = n; s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
        s.n_seeds = 0; s.n_hits = 0; s.n_located = 0; s.nf = 0; s.pg = -1000; s.upos = 0; s.uvalid = false; s.uprev_valid = false;
        s.kind = MK_NONE; s.aux = 0; s.o1 = 0; s.ptr = nullptr;
    }
    // Work queue: one global counter, but a wavefront takes MP_SEED_CHUNK items per atomic into a private range and hands
    // them to its lanes as they finish (a single word sustains only ~90 returning atomics per microsecond chip-wide).
    uint32_t pool_next = 0, pool_end = 0;       // wave-uniform
    bool drained = false;
    uint32_t wave_trips = 0, lane_busy = 0;
    for (;;) {
        wave_trips++;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            mp_lane_state &s = st[j];
            bool &have = hv[j];
            const unsigned long long need = __ballot(!have);
            if (need) {
                if (pool_next == pool_end && !drained) {
                    uint32_t base = 0;
                    const int leader = __ffsll((long long)need) - 1;
                    if (lane == leader) base = atomicAdd(queue, (unsigned int)MP_SEED_CHUNK);
                    base = __shfl(base, leader);
                    if (base >= n_items) drained = true;
                    else { pool_next = base; pool_end = base + MP_SEED_CHUNK < n_items ? base + MP_SEED_CHUNK : n_items; }
                }
                const uint32_t avail = pool_end - pool_next;
                if (avail) {
                    const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                    if (!have && rank < avail) {
                        // a new item: its first load is its scan row (and its length)
                        const uint32_t rs = pool_next + rank;
                        s.rs = rs; s.strand = (int)(rs & 1); s.len = 0; s.i = 0; s.seed_len = 0;
                        s.l = 0; s.r = n; s.last_l = 0; s.last_r = n; s.last_seed_len = 0;
                        s.n_seeds = 0; s.n_hits = 0; s.n_located = 0; s.nf = 0; s.pg = -1000; s.upos = 0; s.uvalid = false;
                        s.kind = MK_BEGIN; s.aux = 0; s.ptr = scan + (uint64_t)rs * S;
                        have = true;
                    }
                    const uint32_t want = (uint32_t)__popcll(need);
                    pool_next += want < avail ? want : avail;
                }
            }
        }
        if (!__any(hv[0] || hv[1])) { if (drained && pool_next == pool_end) break; else continue; }   // nothing to do this trip
        uint4 Q0[2], Q1[2], Q2[2], Q3[2];
        uint32_t A0[2], A1[2], B0[2], B1[2], ALEN[2];
#pragma unroll
        for (int j = 0; j < 2; j++) asm volatile("" : "=v"(Q0[j]), "=v"(Q1[j]), "=v"(Q2[j]), "=v"(Q3[j]), "=v"(A0[j]), "=v"(A1[j]), "=v"(B0[j]), "=v"(B1[j]), "=v"(ALEN[j]));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
#pragma unroll
        for (int j = 0; j < 2; j++) {
            mp_lane_state &s = st[j];
            bool &have = hv[j];
            uint32_t *const sl = sl2[j];
            uint4 &q0 = Q0[j], &q1 = Q1[j], &q2 = Q2[j], &q3 = Q3[j];
            uint32_t &a0 = A0[j], &a1 = A1[j], &b0 = B0[j], &b1 = B1[j], &alen = ALEN[j];
            (void)sl; (void)q1; (void)q2; (void)q3; (void)a1; (void)b0; (void)b1; (void)alen;
            if (have) {
                lane_busy++;
                const uint32_t kind = s.kind;
                if (kind >= MK_LKT16U) {
                    const uint4 *p = reinterpret_cast<const uint4 *>(s.ptr);
                    q0 = p[0];
                    if (kind != MK_LKT16U) { q1 = p[1]; q2 = p[2]; q3 = p[3]; }
                    if (kind == MK_BEGIN && s.aux == 0) alen = lens[s.rs >> 1];
                } else if (kind != MK_NONE) {
                    const uint32_t *p = reinterpret_cast<const uint32_t *>(s.ptr);
                    const bool one = kind == MK_SA || (kind == MK_PRE && !ix.kmer15x3) || (kind == MK_LKT13 && s.aux);
                    a0 = p[0];
                    a1 = p[one ? 0 : 1];
                    if (kind == MK_PRE && ix.kmer15x3) { const uint2 e2 = *reinterpret_cast<const uint2 *>(ix.kmer15x3 + s.aux); b0 = e2.x; b1 = e2.y; }
                }
            }
        }
        // the one wait of the trip: everything issued above (and the stores of the trip before) has landed
        __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), expcnt and lgkmcnt left alone
        // ---------------------------------------------------------------- consume
#pragma unroll
        for (int j = 0; j < 2; j++) {
            mp_lane_state &s = st[j];
            bool &have = hv[j];
            uint32_t *const sl = sl2[j];
            uint4 &q0 = Q0[j], &q1 = Q1[j], &q2 = Q2[j], &q3 = Q3[j];
            uint32_t &a0 = A0[j], &a1 = A1[j], &b0 = B0[j], &b1 = B1[j], &alen = ALEN[j];
            (void)sl; (void)q1; (void)q2; (void)q3; (void)a1; (void)b0; (void)b1; (void)alen;
            if (have) {
                const uint32_t kind = s.kind;
                uint32_t next = MK_NONE;
                bool fin = false, mid_close = false, text_now = false;
                uint32_t trev = 0;
                if (kind == MK_BEGIN) {
                    const uint32_t base = s.aux * 16u;
                    if (base < S) { sl[base] = q0.x; sl[base + 1] = q0.y; sl[base + 2] = q0.z; sl[base + 3] = q0.w; }
                    if (base + 4 < S) { sl[base + 4] = q1.x; sl[base + 5] = q1.y; sl[base + 6] = q1.z; sl[base + 7] = q1.w; }
                    if (base + 8 < S) { sl[base + 8] = q2.x; sl[base + 9] = q2.y; sl[base + 10] = q2.z; sl[base + 11] = q2.w; }
                    if (base + 12 < S) { sl[base + 12] = q3.x; sl[base + 13] = q3.y; sl[base + 14] = q3.z; sl[base + 15] = q3.w; }
                    if (s.aux == 0) {
                        // A read below the DP cutoff floor (30 bases, definitions.h:166-167) can never pass a DP (its best possible score
                        // is its length), so it gets no seeds and therefore no DP instance: it ends up unaligned like any read without a
                        // hit.  (The reference gives up on the whole SIMD group such an instance falls into, CPU_DP.cpp:294-299 -- not reproduced.)
                        const int ln = (int)alen;
                        s.len = ln < MP_MIN_ALIGNABLE_LEN ? 0 : ln;
                    }
                    s.aux++;
                    if (s.aux * 16u < S) { next = MK_BEGIN; s.ptr = reinterpret_cast<const uint4 *>(s.ptr) + 4; }
                } else if (kind == MK_LKT13 || kind == MK_LKT16 || kind == MK_LKT16U) {
                    // LKT (re)start: the table stands for the first 13 (16) bases of the seed
                    uint32_t nextl, nextr;
                    int k;
                    if (kind == MK_LKT13) { nextl = s.aux ? 1u : a0 + 1; nextr = a1; k = MP_LKT_K - 1; ev.lkt++; }
                    else {
                        if (kind == MK_LKT16U) { nextl = q0.x; nextr = q0.y; } else { nextl = a0 + 1; nextr = a1; }
                        k = 15; ev.lkt16++; ev.skipped += 3;                  // the three extension steps the table stands for
                    }
                    s.i += k; s.seed_len = k;
                    if (nextl <= nextr) {
                        s.l = nextl; s.r = nextr; ++s.seed_len; ++s.i;
                        if (s.i >= s.len) fin = true;
                        else if (kind == MK_LKT16U && nextl == nextr && q0.z != 0xFFFFFFFFu) {
                            // a 16-mer that occurs once: the entry says where, and what stands in front of it -- compared right away
                            s.upos = q0.z; s.uvalid = true; trev = q0.w; text_now = true;
                        }
                    } else mid_close = true;
                } else if (kind == MK_OCC || kind == MK_OCC2) {
                    // one backward extension: rank of the symbol at both ends of the range (one block, or two over two trips)
                    const uint32_t c = s.aux;
                    uint32_t k1 = s.l, k2 = s.r + 1;
                    k1 -= (k1 > ix.inv_sa0); k2 -= (k2 > ix.inv_sa0);
                    const uint32_t b1 = k1 / MP_OCC_SYMS, off1 = k1 - b1 * MP_OCC_SYMS;
                    const uint32_t b2 = k2 / MP_OCC_SYMS, off2 = k2 - b2 * MP_OCC_SYMS;
                    mp_blk_view v;
                    mp_blk_match(q0, q1, q2, q3, c, v);
                    const uint32_t ca = mp_blk_rank(v, kind == MK_OCC ? off1 : off2);
                    uint32_t o1 = ca, o2 = ca;
                    bool done = true;
                    if (kind == MK_OCC) {
                        if (b1 == b2) o2 = mp_blk_rank(v, off2);
                        else { s.o1 = o1; next = MK_OCC2; s.ptr = ix.occ + b2; done = false; }
                    } else o1 = s.o1;
                    if (done) {
                        ev.occ += 2; ev.blk += b1 == b2 ? 1u : 2u;
                        const uint32_t cumc = c == 0 ? cum0 : c == 1 ? cum1 : c == 2 ? cum2 : cum3;     // (ix.cum[c] would be a load from the argument block)
                        const uint32_t nextl = cumc + o1 + 1, nextr = cumc + o2;
                        if (nextl <= nextr) {
                            if (s.seed_len >= P.seed_min_length && nextr - nextl < s.r - s.l) { s.last_r = s.r; s.last_l = s.l; s.last_seed_len = (uint32_t)s.seed_len; }
                            s.l = nextl; s.r = nextr; ++s.seed_len; ++s.i;
                            if (s.i >= s.len) fin = true;
                        } else mid_close = true;
                    }
                } else if (kind == MK_SA) {
                    // the text position of a match whose range became a single row: from here on it is extended by comparing with the text
                    s.upos = a0; s.uvalid = true;
                    if (a0 == 0) { mid_close = true; ev.skipped++; }           // the text starts here: what the failing rank query reports
                } else if (kind == MK_TEXT) {
                    const uint32_t pend = s.upos - 1;                          // text base that must equal scan base s.i
                    uint32_t t16;                                               // text bases pend-15 .. pend, LSB-first
                    if (pend >= 15) t16 = (uint32_t)((((uint64_t)a1 << 32) | a0) >> (2 * ((pend - 15) & 15)));
                    else t16 = a0 << (2 * (15 - pend));
                    const uint32_t r = __brev(t16);
                    trev = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);    // text base pend in bits 1:0, pend - 1 in 3:2, ...
                    text_now = true; ev.text++;
                }
                if (text_now) {
                    // The next base of the backward search is the text base in front of the match, 16 of them per word instead of one
                    // rank block each.  The range cannot shrink any more (the re-seed bookkeeping stands still) and the seed is
                    // reported by its text position (MP_SEED_LOCATED).
                    const int avail = min(s.len - s.i, (int)min(s.upos, (uint32_t)1 << 20));     // scan bases left / text bases in front of the match
                    const int nn = avail < 16 ? avail : 16;
                    const uint32_t wq = (uint32_t)s.i >> 4;
                    const uint32_t scan16 = (uint32_t)((((uint64_t)sl[wq + 1] << 32) | sl[wq]) >> (2 * ((uint32_t)s.i & 15)));
                    const uint32_t x = scan16 ^ trev;
                    int same = x ? (__ffs((int)x) - 1) >> 1 : 16;
                    const bool stop = same < nn;
                    if (!stop) same = nn;
                    s.upos -= (uint32_t)same; s.seed_len += same; s.i += same; ev.skipped += (uint32_t)same;
                    if (stop || same == avail) {
                        if (s.i >= s.len) fin = true;
                        else { mid_close = true; ev.skipped++; }               // the failing step
                    }
                }
                mp_seed_rec *slab = slabs + (uint64_t)s.rs * MP_SEED_SLAB;
                const int i_end = s.i;                                     // (where a seed that ends in this trip ended: the base that did not extend it)
                if (mid_close || fin) {
                    const int x = mid_close ? (s.strand == 0 ? s.len - s.i : s.i - s.seed_len) : (s.strand == 0 ? 0 : s.len - s.seed_len);
                    mp_close_seed<MP_SEED_SLAB>(s, P, n, x, slab);
                    if (mid_close) { ++s.i; fin = s.i >= s.len; }
                }
                // ---- a lane without a seed under way: where does the next one start?  (pre-filter, then the table look-up)
                if (!fin && next == MK_NONE && s.seed_len == 0) {
                    // the scan bases from here on as a 96-bit window: enough for the six positions two pre-filter entries answer and for the
                    // keys of the two entries after them
                    uint32_t wq0 = (uint32_t)s.i >> 4;
                    uint32_t w0 = sl[wq0], w1 = sl[wq0 + 1], w2 = sl[wq0 + 2];
                    auto bases_at = [&](int pos) -> uint64_t {                   // scan bases pos .. pos + 16 at least (pos - 16 wq0 <= 31)
                        const uint32_t off = (uint32_t)pos - 16u * wq0;
                        const uint64_t win = off < 16 ? (((uint64_t)w1 << 32) | w0) : (((uint64_t)w2 << 32) | w1);
                        return win >> (2 * (off & 15));
                    };
                    if (kind == MK_PRE) {
                        // what just arrived, as one bit per position: is the 17-mer that starts there anywhere in the text?  s.nf = present
                        // bits (low half) and known bits (high half) of the positions pg .. pg + 15; the two entries answer three positions each,
                        // from the offsets the lane chose when it asked (kept in s.nf meanwhile)
                        uint32_t nf;
                        if (ix.kmer15x3) {
                            const unsigned long long e1 = ((unsigned long long)a1 << 32) | a0, e2 = ((unsigned long long)b1 << 32) | b0;
                            const int o1 = (int)(s.nf & 255u), o2 = (int)((s.nf >> 8) & 255u);
                            nf = ((7u << o1) | (7u << o2)) << 16;
    #pragma unroll
                            for (int q = 0; q < 6; q++) {
                                const int off = (q < 3 ? o1 : o2 - 3) + q;
                                const uint64_t k17 = bases_at(s.i + off) & ((1ull << 34) - 1);
                                const int aa = q % 3;
                                const uint32_t bit = aa == 0 ? (uint32_t)(k17 & 15) : aa == 1 ? 16u + ((uint32_t)(k17 & 3) | ((uint32_t)(k17 >> 32) & 3u) << 2)
                                                                                              : 32u + (uint32_t)((k17 >> 30) & 15);
                                nf |= (uint32_t)(((q < 3 ? e1 : e2) >> bit) & 1ull) << off;
                            }
                        } else nf = ((a0 >> (uint32_t)(bases_at(s.i) & 31)) & 1u) | 0x10000u;       // (bit = the key's low five bits, whatever its length)
                        s.nf = nf; s.pg = s.i;
                    }
                    const int lim = s.len - P.seed_min_length;                    // the last position a seed may start at (`break` of the scan loop beyond)
                    if (s.i > lim) fin = true;
                    else {
                        bool start_here = true;
                        if (pf) {
                            // pre-filter: is the 17-mer that starts at a scan position anywhere in the text?  A futile start (a seed shorter than
                            // seed_min_length: nothing reported, restart one base on -- CHECK_AND_ADD_RANGE) costs no look-up.  A seed is reported
                            // only from seed_min_length bases on, so a start at p is futile when ANY of the 17-mers at p .. p + seed_min_length - 17
                            // is absent (`need` windows: six with soap4.ini's 22-base seeds, one with 17), and an absent 17-mer at z rules out every
                            // start in [z - (need - 1), z]: the scan jumps behind the highest known such z, again and again.  That is all the
                            // correctness there is to it -- WHICH positions a lane asks about is free: with six windows per start the two entries
                            // of a trip are taken six positions apart (p + 5 .. 7 and p + 11 .. 13: a read's mismatch makes 22 starts futile, two
                            // trips rule them out), and a start is tried as soon as nothing known speaks against it and something known speaks for it.
                            const int need = ix.kmer15x3 ? min(P.seed_min_length - (int)ix.kmer_k + 1, 6) : 1;
                            const uint32_t present = s.nf & 0xFFFFu, known = s.nf >> 16, zeros = known & ~present, wmask = (1u << need) - 1u;
                            int a = s.i - s.pg;
                            if (a >= 0 && a < 16) {
                                const int a_in = a;
                                for (;;) {
                                    const uint32_t win = (zeros >> a) & wmask;     // known absent among the windows of a start at pg + a
                                    if (!win) break;
                                    a += 32 - __clz((int)win);
                                }
                                const int p = s.pg + a;
                                if (p > lim) { ev.kmer += (uint32_t)(lim - s.i + 1); s.i = lim + 1; fin = true; start_here = false; }
                                else {
                                    // (not right after a seed ended: what is known then is mostly about the stretch that just matched)
                                    start_here = ((present >> a) & wmask) != 0 && !(mid_close && need > 1);
                                    ev.kmer += (uint32_t)(a - a_in) + (start_here ? 1u : 0u);
                                    s.i = p;
                                    if (a != a_in) { wq0 = (uint32_t)p >> 4; w0 = sl[wq0]; w1 = sl[wq0 + 1]; w2 = sl[wq0 + 2]; }     // the window moves with the scan
                                }
                            } else start_here = false;
                            if (!fin && !start_here) {
                                next = MK_PRE;
                                if (ix.kmer15x3) {
                                    // two entries per trip.  Six windows per start: the first entry at the window that contains the base the last seed
                                    // ended at (i_end - 16, when that is among this start's windows) or at the start's last window, the second six on.
                                    int o1 = 0, o2 = 3;
                                    if (need == 6) { o1 = mid_close ? min(max(i_end - 16 - s.i, 0), 5) : 5; o2 = o1 + 6; }
                                    s.ptr = ix.kmer15x3 + ((bases_at(s.i + o1) >> 4) & ((1ull << 30) - 1));
                                    s.aux = (uint32_t)((bases_at(s.i + o2) >> 4) & ((1ull << 30) - 1));
                                    s.nf = (uint32_t)o1 | ((uint32_t)o2 << 8);
                                    s.pg = -1000;
                                    ev.kload += 2;
                                } else { s.ptr = ix.kmer17 + ((bases_at(s.i) & kmask) >> 5); s.pg = -1000; ev.kload++; }
                            }
                        }
                        if (start_here) {
                            const uint32_t key16 = (uint32_t)bases_at(s.i);
                            if (pf && ix.lkt16 && (key16 & (MP_LKT_ENTRIES - 1)) != 0) {
                                // the 17-mer is reachable, so its first 16 bases are: their SA range straight from the 16-mer table instead of
                                // the 13-mer table and three extension steps (all below seed_min_length, none observable).  Keys whose 13
                                // rightmost bases are 'A' keep the 13-mer path (kernels_index.hip, lkt16_hist_kernel).
                                if (ix.lkt16u) { next = MK_LKT16U; s.ptr = ix.lkt16u + key16; }
                                else { next = MK_LKT16; s.ptr = ix.lkt16 + (key16 - 1); }
                            } else {
                                const uint32_t key = key16 & (MP_LKT_ENTRIES - 1);
                                next = MK_LKT13; s.aux = key == 0; s.ptr = ix.lkt + (key ? key - 1 : 0);
                            }
                        }
                    }
                }
                // ---- a lane with a seed under way: its next extension
                if (!fin && next == MK_NONE) {
                    if (text_ok && s.l == s.r && s.l != 0) {
                        if (s.uvalid) {
                            const uint32_t pend = s.upos - 1;
                            next = MK_TEXT; s.ptr = ix.text + ((pend >= 15 ? pend - 15 : 0u) >> 4);
                        } else { next = MK_SA; s.ptr = ix.sa_dense + s.l; ev.sa++; }
                    } else {
                        const uint32_t c = (sl[(uint32_t)s.i >> 4] >> (2 * ((uint32_t)s.i & 15))) & 3u;
                        uint32_t k1 = s.l;
                        k1 -= (k1 > ix.inv_sa0);
                        next = MK_OCC; s.aux = c; s.ptr = ix.occ + k1 / MP_OCC_SYMS;
                    }
                }
                if (fin) {
                    seeds_per_rs[s.rs] = s.n_seeds;
                    hits_per_rs[s.rs] = s.n_hits;
                    ev.located += s.n_located;
                    if (s.n_seeds > slab_limit) {
                        uint32_t slot = (uint32_t)atomicAdd(&ctr->overflow, 1ull);
                        if (slot < overflow_cap) { overflow_rs[slot] = s.rs; over_slot[s.rs] = slot; }
                    }
                    have = false; next = MK_NONE;
                }
                s.kind = next;
            }
        }
#pragma clang diagnostic pop
    }
    { uint32_t lb = lane_busy; for (int o = MP_WAVE / 2; o > 0; o >>= 1) lb += __shfl_down(lb, o);
      if (lane == 0) { atomicAdd(&ctr->n_wave_trips, (unsigned long long)wave_trips); atomicAdd(&ctr->n_lane_trips, (unsigned long long)lb); } }
    uint32_t tot[10] = {ev.occ, ev.blk, ev.lkt, ev.lkt16, ev.kmer, ev.kload, ev.text, ev.sa, ev.skipped, ev.located};
#pragma unroll
    for (int q = 0; q < 10; q++)
        for (int o = MP_WAVE / 2; o > 0; o >>= 1) tot[q] += __shfl_down(tot[q], o);
    if (lane == 0) {
        unsigned long long *dst[10] = {&ctr->n_occ, &ctr->n_occ_loaded, &ctr->n_lkt, &ctr->n_lkt16, &ctr->n_kmer, &ctr->n_kmer_loads, &ctr->n_text_words, &ctr->n_sa,
                                       &ctr->n_steps_skipped, &ctr->n_located};
#pragma unroll
        for (int q = 0; q < 10; q++) if (tot[q]) atomicAdd(dst[q], (unsigned long long)tot[q]);
    }
}


