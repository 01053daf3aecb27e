// megapath_amd/csrc/kernels_reduce.hip -- product code: per-reference best scores of a batch, reduced on the device across all the
// index chunks the batch is aligned against ("resolve best-per-reference hits across chunks in one pass instead of
// round-tripping FASTQ comments").
//
// What the annotated-FASTQ writers of the reference keep of a pair's alignments is, per mate, ONE score per reference sequence:
//   pairDeepDPOutputFastqAPI        BGS-IO.cpp:1966-2091   (deep-DP pairs and mate-rescued pairs)
//   unproperlypairDPOutputFastqAPI  BGS-IO.cpp:1384-1446   (single-end hits of the pairs left over)
//   decideTargetChr                 BGS-IO.cpp:1312-1341   (an alignment whose ends lie in different sequences is dropped)
//   normalizeScore                  BGS-IO.cpp:1943-1961   (both mates on the same sequence: each gets the sum of the two scores)
// and the chained runs of the NT loop (runMegaPath.sh:191-227) merge those lists through the header comments
// (getMappingFromHeader, BGS-IO.cpp:1348-1371): the final comment of a read holds the entries of the LAST chunk first, then the chunk
// before, ..., each chunk's entries in ascending sequence order, every entry whose score reaches top% of the read's best score over
// all chunks (the thresholds only rise from run to run, so filtering once with the final best score keeps exactly what the chain
// keeps: DESIGN.md section 4, "Reduction over chunks").
//
// Here every hit of every list becomes one 64-bit key per mate, in a slot fixed by its position in its list (no atomics):
//   read id : 22 | chunks after this one : 10 | sequence id : 22 | 1023 - score : 10
// Sorting the keys of a batch orders them by read, later chunks first, sequence ascending, best score first; the first key of every
// (read, chunk, sequence) run is the entry the writers would print.  One list per read goes back to the host, however many chunks.
#pragma once

#define MP_RED_NULL 0xFFFFFFFFFFFFFFFFull
#define MP_RED_MAX_SEQ ((1u << 22) - 1u)
#define MP_RED_MAX_CHUNKS 1024u

struct mp_dev_translate {          // the .tra table of an index chunk (HSP.c:569-640), for decideTargetChr on the device
    const uint32_t *grid;          // one entry per 2^18 packed positions: a row at or after the right one
    const uint64_t *start;         // per row: first packed position
    const uint32_t *chr;           //          1-based sequence id
    uint32_t n_grid, n_rows;
};

// mp_index_target_chr (api.hip) on the device: 0 when the alignment's two ends lie in different sequences
__device__ __forceinline__ uint32_t mp_dev_target_chr(const mp_dev_translate &t, uint64_t pos, uint32_t len)
{
    uint64_t g = pos >> 18;
    if (g >= t.n_grid) g = t.n_grid - 1;
    uint32_t v = t.grid[g];
    while (v > 0 && t.start[v] > pos) v--;
    const uint32_t c1 = t.chr[v];
    const uint64_t last = pos + len - 1;
    uint32_t c2 = c1;
    if (v + 1 < t.n_rows && t.start[v + 1] <= last) {
        uint64_t g2 = last >> 18;
        if (g2 >= t.n_grid) g2 = t.n_grid - 1;
        uint32_t w = t.grid[g2];
        while (w > 0 && t.start[w] > last) w--;
        c2 = t.chr[w];
    }
    return c1 == c2 ? c1 : 0u;
}

__device__ __forceinline__ unsigned long long mp_red_key(uint32_t read, uint32_t chunk_inv, uint32_t seq, int score)
{
    if (seq == 0 || score <= 0) return MP_RED_NULL;             // dropped alignment, or nothing a writer would list (scores are > 0)
    const uint32_t s = score > 1023 ? 1023u : (uint32_t)score;
    return ((unsigned long long)read << 42) | ((unsigned long long)chunk_inv << 32) | ((unsigned long long)seq << 10) | (unsigned long long)(1023u - s);
}

// One lane per pair of mate alignments (a deep-DP hit, or a mate rescue): Writer::pair of the host code / pairDeepDPOutputFastqAPI.
//   mode 2 (-P): a pair with a dropped mate is dropped altogether (BGS-IO.cpp:2003-2011)
template <bool RESCUE>
__global__ void reduce_pairs_kernel(const void *__restrict__ hits_v, uint32_t n, const uint32_t *__restrict__ lens, mp_dev_translate tr, int mode, uint32_t chunk_inv,
                                    unsigned long long *__restrict__ keys /* 2 per hit */, uint32_t *__restrict__ resc_done)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t q;
    uint64_t a1, a2;
    int s1, s2;
    if (!RESCUE) {
        const mp_pair_hit h = reinterpret_cast<const mp_pair_hit *>(hits_v)[t];
        q = h.pair_id; a1 = h.algnmt_1; a2 = h.algnmt_2; s1 = h.score_1; s2 = h.score_2;
    } else {
        const mp_rescue_hit h = reinterpret_cast<const mp_rescue_hit *>(hits_v)[t];
        q = h.pair_id;
        if ((h.aligned_read & 1u) == 0) { a1 = h.aligned_pos; a2 = h.dp_pos; s1 = h.aligned_score; s2 = h.dp_score; }
        else { a1 = h.dp_pos; a2 = h.aligned_pos; s1 = h.dp_score; s2 = h.aligned_score; }
        resc_done[q] = 1u;
    }
    uint32_t c1 = mp_dev_target_chr(tr, a1, lens[2 * q]), c2 = mp_dev_target_chr(tr, a2, lens[2 * q + 1]);
    if (c1 == 0) s1 = 0;
    if (c2 == 0) s2 = 0;
    if (mode == 2 && (c1 == 0 || c2 == 0)) { c1 = c2 = 0; s1 = s2 = 0; }
    if (c1 == c2 && c1 != 0) s1 = s2 = s1 + s2;                                     // normalizeScore
    keys[2 * (size_t)t] = mp_red_key(2 * q, chunk_inv, c1, s1);
    keys[2 * (size_t)t + 1] = mp_red_key(2 * q + 1, chunk_inv, c2, s2);
}

// One lane per single-end hit of a pair that neither round paired: Writer::single / unproperlypairDPOutputFastqAPI (nothing in -P mode)
__global__ void reduce_singles_kernel(const mp_single_hit *__restrict__ hits, uint32_t n, const uint32_t *__restrict__ lens, mp_dev_translate tr, int mode, uint32_t chunk_inv,
                                      const uint32_t *__restrict__ pair_done, const uint32_t *__restrict__ resc_done, unsigned long long *__restrict__ keys)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const mp_single_hit h = hits[t];
    const uint32_t q = h.read_id >> 1;
    unsigned long long k = MP_RED_NULL;
    if (mode != 2 && !pair_done[q] && !resc_done[q]) k = mp_red_key(h.read_id, chunk_inv, mp_dev_target_chr(tr, h.algnmt, lens[h.read_id]), h.score);
    keys[t] = k;
}

// pairs of the chunk by how they were aligned (the executable's closing line): [0] deep-DP round, [1] mate rescue
// cls (may be null): per pair 0 = deep-DP round, 1 = mate rescue, 2 = neither (the order the reference writes a batch in, alignment.cpp:105-330)
__global__ void reduce_count_kernel(const uint32_t *__restrict__ pair_done, const uint32_t *__restrict__ resc_done, uint32_t n_pairs, unsigned long long *__restrict__ out,
                                    uint8_t *__restrict__ cls)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool d = q < n_pairs && pair_done[q] != 0, r = q < n_pairs && !d && resc_done[q] != 0;
    if (cls && q < n_pairs) cls[q] = d ? 0 : r ? 1 : 2;
    const unsigned long long bd = __ballot(d), br = __ballot(r);
    if ((threadIdx.x & (MP_WAVE - 1)) == 0) {
        if (bd) atomicAdd(&out[0], (unsigned long long)__popcll(bd));
        if (br) atomicAdd(&out[1], (unsigned long long)__popcll(br));
    }
}

// after the sort: flag the first key of every (read, chunk, sequence) run (its best score), nulls excluded
__global__ void reduce_first_flag_kernel(const unsigned long long *__restrict__ keys, uint64_t n, uint32_t *__restrict__ flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { flag[i] = 0; return; }
    const unsigned long long k = keys[i];
    flag[i] = (k != MP_RED_NULL && (i == 0 || (keys[i - 1] >> 10) != (k >> 10))) ? 1u : 0u;
}

__global__ void reduce_compact_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ slot, uint64_t n,
                                      unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[slot[i]] = keys[i];
}
