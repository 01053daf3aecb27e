// megapath_amd/csrc/kernels_floor.hip -- product code: the `-top` rule across resident index chunks (included by api.hip).
//
// The annotated FASTQ keeps, per read, the entries that reach top% of the read's best score over ALL chunks (getMappingFromHeader BGS-IO.cpp:1348-1371,
// pairDeepDPOutputFastqAPI :2034-2089, unproperlypairDPOutputFastqAPI :1422-1441; DESIGN.md section 4, "Reduction over chunks": one filter with the final
// best score prints exactly what the chain prints).  With every chunk resident the best score a read is KNOWN to reach (the deep-DP round of every chunk,
// reduced on the device: mp_reduce_chunk_part) is there before the rescue rounds of the chunks start, and a rescue round whose results cannot reach top% of it
// can be left out -- for most pairs that align well somewhere, that is the rescue work of every other chunk: ~60 single-end DP instances per pair and chunk
// around chance 17-mers (DV-DPForSingleReads.cpp:121-215), all failing a cutoff of 30, let alone of 100.
//
// What may be left out is decided per PAIR, not per instance: the mate rescue starts from the LAST four single-end hits of a read in (score descending)
// order (DV-DPfunctions.cpp:1048-1079) and a rescued pair hides its single-end entries (the writers' precedence), so low-scoring hits
// decide what is rescued from and what is printed.  A pair keeps ALL its rescue work in this chunk unless NO entry it could produce can matter:
//   U_x        an upper bound on the score of every single-end instance of mate x in this chunk (below), 0 without instances
//   T_x        = top% of the best score known for mate x (entries below it are never printed); an entry of score s is certainly irrelevant iff 100 s < top B_x - 100
//   single-end entry of x                  irrelevant iff U_x below T_x
//   rescue anchored at x (needs a hit of x) impossible iff U_x < cutoff(x); else its entries are: both mates on one sequence -> U_x + L_y in both headers,
//                                           irrelevant iff below min(T_x, T_y); on different sequences -> each its own score: x's irrelevant as above, the
//                                           rescued mate's at most L_y: irrelevant iff L_y below T_y
// (the rescued mate scores at most its length).  Pairs of the LAST chunk always keep their work: that chunk's classes are the sections of the output.
//
// The bound U: blocks of five read bases (MP_FLOOR_K).  Score <= L - loss, and every block whose 5-mer occurs NOWHERE in the window costs the alignment at
// least 3 -- its five columns cannot all be matched in a row without an event (they would spell the 5-mer in the window), so they hold a mismatch (1 + 2), an
// inserted column (a run of g of them costs 2 + 2 g >= 3 per block it touches), a deletion between two of them (>= 3), or they are clipped (a whole block 5;
// the two blocks the clips end in at least 1 each).  With E such blocks: U = L - 3 max(0, E - 2) - min(E, 2), for each of the five block phases; the smallest is
// taken.  The window's 5-mers go into a 1024-bit set per instance in LDS (one column of words per lane: no bank conflicts, no atomics across lanes): ~2 500
// instructions per instance against the certificate kernel's 44 000.  A random 150 x 210 instance around a 17-20-base seed gets U = 84 +- 4 (true maximum ~20);
// the largest over a pair's ~60 instances is ~93 (blocks of six: the same; of seven: 99), so pairs whose best known score leaves min(T) - L_mate above that --
// mates within ~4.5 % of the reference -- lose their rescue work in every other chunk.  tests/test_gpu_top_floor.py: U >= the exact DP score on the certificate
// tests' adversarial instances; the executable's output with the rule == without it == the chained oracle's.
#pragma once

#define MP_FLOOR_K 5              // block length: 5 and 6 bound equally well (prototype: the largest U over a pair's 60 instances 93.2 / 92.7), 5 needs a quarter of the LDS
#define MP_FLOOR_WORDS ((1 << (2 * MP_FLOOR_K)) / 32)     // 32 words of 32 bits per instance
#define MP_FLOOR_MAXL 320
#define MP_FLOOR_MAXM 384
#define MP_FLOOR_SKIP 0xFFFFu                              // no bound computed (the pair cannot be dropped anyway, or the shape is not covered)

// 100 x score below which an entry of a read whose best known score is `best` is certainly not printed (one score point of slack for the double comparison
// `score >= best * top` of the writers); <= 0: everything may matter
__device__ __forceinline__ long long mp_floor_thr100(uint32_t best, int top_pct) { return (long long)top_pct * (long long)best - 100; }

// One lane per single-end DP instance (task): ub[t] = the bound U, or MP_FLOOR_SKIP.
__global__ void __launch_bounds__(64)
floor_bound_kernel(const uint32_t *__restrict__ text, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
                   const mp_dp_task *__restrict__ tasks, uint32_t n_tasks, const uint32_t *__restrict__ best_known, int top_pct, uint16_t *__restrict__ ub)
{
    __shared__ uint32_t s_set[MP_FLOOR_WORDS][64];       // [word][lane]
    const int lane = threadIdx.x;
    const uint32_t t = blockIdx.x * 64u + (uint32_t)lane;
    bool work = t < n_tasks;
    mp_dp_task task{};
    int L = 0, m = 0;
    if (work) {
        task = tasks[t];
        L = (int)lens[task.read_id]; m = (int)task.win_len;
        // a mate whose pair could not be dropped whatever the bound says (nothing known about it or about its mate's side of the conditions) needs none
        const uint32_t bx = best_known[task.read_id];
        if (mp_floor_thr100(bx, top_pct) <= 0 || L > MP_FLOOR_MAXL || L < MP_FLOOR_K || m > MP_FLOOR_MAXM || m < 1) { ub[t] = MP_FLOOR_SKIP; work = false; }
    }
    if (!__any(work)) return;
    for (int w = 0; w < MP_FLOOR_WORDS; w++) s_set[w][lane] = 0u;
    uint32_t a[MP_FLOOR_MAXL / 16 + 1], win[MP_FLOOR_MAXM / 16 + 2];
#pragma unroll
    for (int k = 0; k <= MP_FLOOR_MAXL / 16; k++) a[k] = 0;
#pragma unroll
    for (int k = 0; k < MP_FLOOR_MAXM / 16 + 2; k++) win[k] = 0;
    if (work) {
        const uint32_t *rd = reads + (uint64_t)task.read_id * Wr;
        const int nw = (L + 15) / 16;
#pragma unroll
        for (int k = 0; k < MP_FLOOR_MAXL / 16; k++) a[k] = mp_oriented_word(rd, nw, L, task.strand, k);
        const uint64_t q0 = task.win_start >> 4;
        const uint32_t sh = 2u * (uint32_t)(task.win_start & 15);
        const int tw = (m + 15) / 16;
#pragma unroll
        for (int k = 0; k < MP_FLOOR_MAXM / 16 + 1; k++)
            if (k < tw) {
                uint64_t two = (uint64_t)text[q0 + k];
                if (sh) two |= (uint64_t)text[q0 + k + 1] << 32;                  // (never a word past the window's last)
                win[k] = (uint32_t)(two >> sh);
            }
        // the window's 6-mers into the set
#pragma unroll
        for (int k = 0; k < MP_FLOOR_MAXM / 16 + 1; k++) {
            if (16 * k + MP_FLOOR_K > m) break;
            const uint64_t two = ((uint64_t)win[k + 1] << 32) | win[k];
            const int last = min(16, m - MP_FLOOR_K + 1 - 16 * k);
            for (int u = 0; u < last; u++) {
                const uint32_t key = (uint32_t)(two >> (2 * u)) & ((1u << (2 * MP_FLOOR_K)) - 1u);
                atomicOr(&s_set[key >> 5][lane], 1u << (key & 31u));             // (this lane's own column: a plain ds_or)
            }
        }
        // the read's 6-mers, every start: start j belongs to block phase j % 6; E of the six phases in six bytes of one word pair
        unsigned long long E = 0;
        int ph = 0;
#pragma unroll
        for (int k = 0; k < MP_FLOOR_MAXL / 16; k++) {
            if (16 * k + MP_FLOOR_K > L) break;
            const uint64_t two = ((uint64_t)a[k + 1] << 32) | a[k];
            const int last = min(16, L - MP_FLOOR_K + 1 - 16 * k);
            for (int u = 0; u < last; u++) {
                const uint32_t key = (uint32_t)(two >> (2 * u)) & ((1u << (2 * MP_FLOOR_K)) - 1u);
                const uint32_t present = (s_set[key >> 5][lane] >> (key & 31u)) & 1u;
                E += (unsigned long long)(present ^ 1u) << (8 * ph);
                ph = ph == MP_FLOOR_K - 1 ? 0 : ph + 1;
            }
        }
        int U = L;
#pragma unroll
        for (int p = 0; p < MP_FLOOR_K; p++) {
            const int e = (int)((E >> (8 * p)) & 255u);
            const int u = L - 3 * max(0, e - 2) - min(e, 2);
            U = min(U, u);
        }
        ub[t] = (uint16_t)max(U, 0);
    }
}

// One lane per failed pair (its mates are the failed reads of slots 2k, 2k + 1; their tasks are task_off[slot] .. task_off[slot + 1]): n_cand of both
// mates becomes 0 when nothing the pair's rescue work in this chunk could produce can matter (header of this file); counts[0] += dropped pairs,
// counts[1] += their instances.
__global__ void floor_decide_kernel(const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const uint32_t *__restrict__ task_off, const uint16_t *__restrict__ ub,
                                    const uint32_t *__restrict__ lens, const uint32_t *__restrict__ best_known, int top_pct, uint32_t *__restrict__ n_cand,
                                    unsigned long long *__restrict__ counts, uint8_t *__restrict__ hopeful /* may be null; [pair] */)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    bool drop = false, hope = false;
    uint32_t n_inst = 0;
    if (2 * k + 1 < n_failed) {
        const uint32_t ra = failed_reads[2 * k], rb = failed_reads[2 * k + 1];
        if ((ra ^ rb) == 1u) {                                                    // (always: both mates of a failed pair are listed, in read order)
            uint32_t U[2] = {0, 0};
            bool known = true;
            for (int x = 0; x < 2; x++)
                for (uint32_t t = task_off[2 * k + x]; t < task_off[2 * k + x + 1]; t++) {
                    const uint32_t u = ub[t];
                    if (u == MP_FLOOR_SKIP) known = false;
                    U[x] = max(U[x], u);
                }
            n_inst = task_off[2 * k + 2] - task_off[2 * k];
            if (known && n_inst) {
                const long long L[2] = {(long long)lens[ra], (long long)lens[rb]};
                const long long T[2] = {mp_floor_thr100(best_known[ra], top_pct), mp_floor_thr100(best_known[rb], top_pct)};
                const long long Tmin = T[0] < T[1] ? T[0] : T[1];
                drop = true;
                bool single_ok = true;
                for (int x = 0; x < 2; x++) {
                    const int y = 1 - x;
                    const bool has = task_off[2 * k + x + 1] > task_off[2 * k + x];
                    if (!has) continue;                                           // no instance of x: no single-end hit of x, nothing anchored at x
                    if (100 * (long long)U[x] >= T[x]) { drop = false; single_ok = false; }      // a single-end entry of x could be printed
                    if ((long long)U[x] >= (long long)mp_dp_cutoff((int)L[x])) {  // x may have a single-end hit: a rescue of y anchored at it
                        if (100 * ((long long)U[x] + L[y]) >= Tmin) drop = false; //   both on one sequence: the sum in both headers
                        if (100 * L[y] >= T[y]) drop = false;                     //   on different sequences: the rescued mate's own score in its header
                    }
                }
                // kept only because a rescued mate was credited with its whole length: floor_mate_kernel looks at what its windows can really give it
                hope = !drop && single_ok;
            }
        }
        if (drop) { n_cand[2 * k] = 0; n_cand[2 * k + 1] = 0; }
        if (hopeful) hopeful[k] = hope ? 1 : 0;
    }
    const unsigned long long b = __ballot(drop);
    unsigned long long inst = drop ? (unsigned long long)n_inst : 0ull;
    for (int o = MP_WAVE / 2; o > 0; o >>= 1) inst += __shfl_down(inst, o);
    if ((threadIdx.x & (MP_WAVE - 1)) == 0 && b) { atomicAdd(&counts[0], (unsigned long long)__popcll(b)); atomicAdd(&counts[1], inst); }
}

// ---- the rescued mate's side of the rule.  floor_decide_kernel credits a mate rescued from a hit of x with its whole length L_y, which keeps every pair whose
// mates are more than ~4.8 % away from their best alignment (150-base reads, -top 95: U_x + L_y against 95 % of the pair's score).  What a rescue anchored at
// instance i of x can give the mate is bounded by the text it searches: [hit, hit + insert_high) to the right of a forward hit, the mirror image to the left of
// a reverse one (mate_rescue_task_kernel, HalfEndAlgnBatch::pack DV-DPfunctions.cpp:1151-1231) -- whatever the hit's offset in x's window, a part of the
// ~960 bases S_i = x's window extended by insert_high on that side.  The same block argument as above with blocks of SIX (a 1 024-base text holds a given
// 6-mer with probability 0.21, a 5-mer with 0.61): U_y(i) = L_y - 3 max(0, E - 2) - min(E, 2), E = blocks of the oriented mate whose 6-mer occurs nowhere in S_i,
// the smallest over the six block phases.  The pair loses its rescue work when, for every instance i of either mate that could be a hit (U_x(i) >= cutoff),
// 100 (U_x(i) + U_y(i)) is below min(T_x, T_y) and 100 U_y(i) below T_y.
// One wavefront per pair that floor_decide_kernel marked (`hopeful`), one lane per instance.  A 4 096-bit set per instance would be 32 KB of LDS per
// wavefront (the reason the bound of x uses blocks of five); here the 6-mers of the four oriented mates (2 mates x 2 strands: 4 x 512 bytes, shared by the
// wavefront) are the filter: a lane streams its ~960 text bases, and only the 6-mers that occur in the mate at all (3.6 % of them) enter the lane's own set --
// 1 024 bits addressed by the key's low ten bits.  Two 6-mers that differ in the last base alone share a bit: a block may count as present that is not,
// which only raises the bound.  The filter is laid out by the FIVE bases two neighbouring 6-mers share (one byte per 5-mer: low nibble = the bases that may
// stand before it, high nibble = the bases that may follow it), so one look-up answers two text positions.
#define MP_FLOOR_KM 6
#define MP_FLOOR_KM_WORDS ((1 << (2 * MP_FLOOR_KM)) / 32)      // 128
#define MP_FLOOR_CORE_WORDS ((1 << (2 * (MP_FLOOR_KM - 1))) / 4)    // 1 024 bytes
#define MP_FLOOR_PM_WORDS 32                                    // the lane's own hashed set
__global__ void __launch_bounds__(64)
floor_mate_kernel(const uint32_t *__restrict__ text, uint64_t text_len, const uint32_t *__restrict__ reads, const uint32_t *__restrict__ lens, uint32_t Wr,
                  const mp_dp_task *__restrict__ tasks, const uint32_t *__restrict__ failed_reads, uint32_t n_failed, const uint32_t *__restrict__ task_off,
                  const uint16_t *__restrict__ ub, const uint32_t *__restrict__ best_known, int top_pct, const uint8_t *__restrict__ hopeful /* null: every pair */,
                  int insert_high, uint32_t *__restrict__ n_cand /* null: no decision (test hook) */, unsigned long long *__restrict__ counts /* [2] pairs looked at, [3] dropped here */,
                  uint16_t *__restrict__ ub_mate /* may be null: U_y per instance */)
{
    __shared__ uint32_t s_y[4][MP_FLOOR_CORE_WORDS];          // [2 * mate + (strand - 1)][5-mer] bytes
    __shared__ uint32_t s_rd[4][MP_FLOOR_MAXL / 16 + 2];      // the oriented mates, 16 bases per word
    __shared__ uint32_t s_p[MP_FLOOR_PM_WORDS][64];           // [word][lane]
    const uint32_t k = blockIdx.x;
    const int lane = threadIdx.x;
    if (2 * k + 1 >= n_failed || (hopeful && !hopeful[k])) return;
    const uint32_t r0 = failed_reads[2 * k];
    if ((r0 ^ failed_reads[2 * k + 1]) != 1u) return;
    const int Lm[2] = {(int)lens[r0], (int)lens[r0 ^ 1u]};
    if (Lm[0] > MP_FLOOR_MAXL || Lm[1] > MP_FLOOR_MAXL) return;                  // (uniform: the pair keeps its work)
    for (int w = lane; w < 4 * MP_FLOOR_CORE_WORDS; w += 64) (&s_y[0][0])[w] = 0u;
    for (int w = lane; w < 4 * (MP_FLOOR_MAXL / 16 + 2); w += 64) {
        const int q = w / (MP_FLOOR_MAXL / 16 + 2), kk = w % (MP_FLOOR_MAXL / 16 + 2);
        const int L = Lm[q >> 1];
        s_rd[q][kk] = mp_oriented_word(reads + (uint64_t)(r0 ^ (uint32_t)(q >> 1)) * Wr, (L + 15) / 16, L, 1u + (uint32_t)(q & 1), kk);
    }
    __syncthreads();
    for (int q = 0; q < 4; q++) {
        const int L = Lm[q >> 1];
        for (int j = lane; j + MP_FLOOR_KM <= L; j += 64) {
            const uint64_t two = ((uint64_t)s_rd[q][(j >> 4) + 1] << 32) | s_rd[q][j >> 4];
            const uint32_t key = (uint32_t)(two >> (2 * (j & 15))) & ((1u << (2 * MP_FLOOR_KM)) - 1u);
            const uint32_t tail = key >> 2, head = key & ((1u << (2 * (MP_FLOOR_KM - 1))) - 1u);       // the 5-mers after the first base / before the last
            atomicOr(&s_y[q][tail >> 2], (1u << (key & 3u)) << (8u * (tail & 3u)));
            atomicOr(&s_y[q][head >> 2], (16u << (key >> (2 * (MP_FLOOR_KM - 1)))) << (8u * (head & 3u)));
        }
    }
    __syncthreads();
    const uint32_t t0 = task_off[2 * k], t1 = task_off[2 * k + 1], t2 = task_off[2 * k + 2];
    const long long T[2] = {mp_floor_thr100(best_known[r0], top_pct), mp_floor_thr100(best_known[r0 ^ 1u], top_pct)};
    const long long Tmin = T[0] < T[1] ? T[0] : T[1];
    bool all_ok = true;
    for (uint32_t base = t0; base < t2; base += 64) {                             // (uniform trip count)
        const uint32_t t = base + (uint32_t)lane;
        const bool have = t < t2;
        bool ok = true;
        if (have) {
            const int x = t >= t1 ? 1 : 0, y = 1 - x;
            const mp_dp_task task = tasks[t];
            const int alen = Lm[x], ulen = Lm[y];
            const uint32_t ux = ub[t];
            int Uy = ulen;
            if (ulen < MP_MIN_ALIGNABLE_LEN) Uy = 0;                              // no rescue instance is made for such a mate
            else if (ulen >= MP_FLOOR_KM) {
                // S_i: every text base a rescue anchored at a hit inside this window can look at
                uint64_t ws, we;
                if (task.strand == 1) { ws = task.win_start; we = task.win_start + task.win_len + (uint64_t)insert_high; }
                else { const uint64_t lo = task.win_start + (uint64_t)alen; ws = lo > (uint64_t)insert_high ? lo - (uint64_t)insert_high : 0; we = task.win_start + task.win_len + (uint64_t)alen; }
                if (we > text_len) we = text_len;
                const int q = 2 * y + (task.strand == 1 ? 1 : 0);                 // the mate is searched on the other strand
                for (int w = 0; w < MP_FLOOR_PM_WORDS; w++) s_p[w][lane] = 0u;
                // whole text words: up to 63 bases more on either side (and 'A's before the first word) only add 6-mers to the set.  Sixteen positions per
                // trip, their sixteen filter look-ups independent of each other; the few that hit (0.6 per trip and lane) enter the lane's set afterwards.
                // (text words in aligned groups of four, one 16-byte load each, eight groups in flight at a time; the packed text is allocated with four
                // words to spare, api.hip mp_index_*)
                const uint8_t *sy = reinterpret_cast<const uint8_t *>(s_y[q]);
                uint32_t prev = 0u;
                const uint64_t w_end = (we + 15) >> 4;
                for (uint64_t wi = (ws >> 4) & ~(uint64_t)3; wi < w_end; wi += 32) {
                    uint4 buf[8];
#pragma unroll
                    for (int b = 0; b < 8; b++) buf[b] = wi + 4 * b < w_end ? *reinterpret_cast<const uint4 *>(text + wi + 4 * b) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        if (wi + 4 * b >= w_end) break;
                        const uint32_t wd[4] = {buf[b].x, buf[b].y, buf[b].z, buf[b].w};
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const uint32_t cur = wd[g];
                            const uint64_t both = ((uint64_t)cur << 32) | prev;
                            uint32_t hits = 0u;
#pragma unroll
                            for (int u = 0; u < 16; u += 2) {
                                // the 6-mers that end at bases u and u + 1 of this word: seven bases, the middle five are the look-up
                                const uint32_t seven = (uint32_t)(both >> (2 * (u + 16 - (MP_FLOOR_KM - 1)))) & ((1u << (2 * MP_FLOOR_KM + 2)) - 1u);
                                const uint32_t byte = sy[(seven >> 2) & ((1u << (2 * (MP_FLOOR_KM - 1))) - 1u)];
                                hits |= ((byte >> (seven & 3u)) & 1u) << u;
                                hits |= ((byte >> (4u + (seven >> (2 * MP_FLOOR_KM)))) & 1u) << (u + 1);
                            }
                            while (hits) {
                                const int u = __ffs((int)hits) - 1;
                                hits &= hits - 1u;
                                const uint32_t key = (uint32_t)(both >> (2 * (u + 16 - (MP_FLOOR_KM - 1)))) & ((1u << (2 * MP_FLOOR_KM)) - 1u);
                                atomicOr(&s_p[(key >> 5) & (MP_FLOOR_PM_WORDS - 1)][lane], 1u << (key & 31u));
                            }
                            prev = cur;
                        }
                    }
                }
                // the mate's own 6-mers, sixteen starts per trip (their look-ups independent of each other): start j belongs to block phase j % 6
                uint32_t Ec[MP_FLOOR_KM] = {0u, 0u, 0u, 0u, 0u, 0u};
                int ph0 = 0;                                                  // phase of this trip's first start
                for (int kw = 0; 16 * kw + MP_FLOOR_KM <= ulen; kw++) {
                    const uint64_t two = ((uint64_t)s_rd[q][kw + 1] << 32) | s_rd[q][kw];
                    const int last = min(16, ulen - MP_FLOOR_KM + 1 - 16 * kw);
                    uint32_t absent = 0u;
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const uint32_t kj = (uint32_t)(two >> (2 * u)) & ((1u << (2 * MP_FLOOR_KM)) - 1u);
                        absent |= (((s_p[(kj >> 5) & (MP_FLOOR_PM_WORDS - 1)][lane] >> (kj & 31u)) & 1u) ^ 1u) << u;
                    }
                    absent &= (1u << last) - 1u;
#pragma unroll
                    for (int pp = 0; pp < MP_FLOOR_KM; pp++) {
                        const int c = pp >= ph0 ? pp - ph0 : pp + MP_FLOOR_KM - ph0;     // the starts u of this trip with (ph0 + u) % 6 == pp: u = c, c + 6, c + 12
                        Ec[pp] += (uint32_t)__popc(absent & ((0x1041u << c) & 0xFFFFu));
                    }
                    ph0 = (ph0 + 16) % MP_FLOOR_KM;
                }
#pragma unroll
                for (int pp = 0; pp < MP_FLOOR_KM; pp++) {
                    const int e = (int)Ec[pp];
                    Uy = min(Uy, ulen - 3 * max(0, e - 2) - min(e, 2));
                }
                Uy = max(Uy, 0);
            }
            if (ub_mate) ub_mate[t] = (uint16_t)Uy;
            if (ux == MP_FLOOR_SKIP) ok = false;
            else if ((long long)ux >= (long long)mp_dp_cutoff(alen) && ulen >= MP_MIN_ALIGNABLE_LEN)
                ok = 100 * ((long long)ux + (long long)Uy) < Tmin && 100 * (long long)Uy < T[y];
        }
        if (__ballot(have && !ok)) all_ok = false;
    }
    if (lane == 0) {
        const bool drop = all_ok && n_cand && t2 > t0;
        if (drop) { n_cand[2 * k] = 0; n_cand[2 * k + 1] = 0; }
        if (counts) {
            atomicAdd(&counts[2], 1ull);
            if (drop) { atomicAdd(&counts[0], 1ull); atomicAdd(&counts[1], (unsigned long long)(t2 - t0)); atomicAdd(&counts[3], 1ull); }
        }
    }
}

// best_known[read] = max(best_known[read], score of the key) for the keys a reduction call has just written (null keys excluded)
__global__ void floor_best_kernel(const unsigned long long *__restrict__ keys, uint64_t n, uint32_t *__restrict__ best_known, uint32_t n_reads)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    if (k == 0xFFFFFFFFFFFFFFFFull) return;
    const uint32_t read = (uint32_t)(k >> 42), score = 1023u - (uint32_t)(k & 1023u);
    if (read < n_reads) atomicMax(&best_known[read], score);
}

// ---- what mp_chunk_stash keeps of a chunk's seed positions, 8 bytes each instead of 16 (the stash of a 256 Ki-pair batch over 88 chunks: 13 GB instead of 25):
// position (low word: a chunk holds < 2^32 bases, and a projection that wrapped below zero comes back by the sign extension key_pos makes of it, kernels_seed.hip)
// | covered length (10 bits; a read has at most MP_SEED_BIG = 384 bases) | pair number (21 bits: batches of up to 2 Mi pairs, mp_chunk_stash checks) | strand.
// (The array has a slot per located hit and only the kept seeds of a read are filled in; what the other slots hold is never read, before or after.)
__device__ __forceinline__ mp_seedpos mp_stash_unpack(unsigned long long w, uint64_t text_len)
{
    mp_seedpos sp;
    const uint64_t o = w & 0xFFFFFFFFull;
    sp.pos = o >= text_len ? (o | 0xFFFFFFFF00000000ull) : o;
    sp.paired_seed_length = (uint32_t)(w >> 32) & 1023u;
    sp.strand_read_id = ((uint32_t)((w >> 42) & 0x1FFFFFu) << 1) | ((uint32_t)(w >> 63) << 31);
    return sp;
}

__global__ void stash_pack_kernel(const mp_seedpos *__restrict__ in, uint64_t n, unsigned long long *__restrict__ out)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mp_seedpos sp = in[i];
    out[i] = (sp.pos & 0xFFFFFFFFull) | ((unsigned long long)(sp.paired_seed_length & 1023u) << 32) |
             ((unsigned long long)((sp.strand_read_id >> 1) & 0x1FFFFFu) << 42) | ((unsigned long long)(sp.strand_read_id >> 31) << 63);
}

__global__ void stash_unpack_kernel(const unsigned long long *__restrict__ in, uint64_t n, uint64_t text_len, mp_seedpos *__restrict__ out)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = mp_stash_unpack(in[i], text_len);
}
