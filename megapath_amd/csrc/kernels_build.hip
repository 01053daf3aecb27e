// megapath_amd/csrc/kernels_build.hip -- product code: index construction on the device, included by api.hip.
//
// What the reference's 2bwt-builder computes from the packed text (2BWT-Builder.c:199-393, BWTConstruct.c -- an incremental BWT
// construction on one core): the BWT with the '$' row left out, the position of that row, the cumulative symbol counts and the sampled
// suffix array.  Here: the whole suffix array by prefix doubling in HBM -- suffixes are ranked by their first 21 symbols (3 bits each,
// 0 = past the end, so '$' sorts first), then repeatedly by (rank, rank of the suffix h further on) with h = 21, 42, 84, ... until all
// ranks differ.  The sorts are device-wide LSD radix sorts of (64-bit key, 32-bit suffix) pairs (rocPRIM through hipcub); everything
// between them is one streaming kernel per step.  28 bytes of HBM per base, no host loop over bases, texts up to 2^32 - 3 bases (the
// reference's own per-index limit).
#pragma once

#define MP_BUILD_K 21      // symbols in the first key: 21 x 3 bits

// key[i] = the first 21 symbols of suffix i (i = 0 .. n; suffix n is the empty one: '$'), idx[i] = i
__global__ void build_init_kernel(const uint8_t *__restrict__ codes, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i > n) return;
    uint64_t k = 0;
#pragma unroll
    for (int t = 0; t < MP_BUILD_K; t++) {
        const uint64_t p = i + (uint64_t)t;
        k = (k << 3) | (p < n ? (uint64_t)(codes[p] & 3u) + 1u : 0u);
    }
    key[i] = k;
    idx[i] = (uint32_t)i;
}

// flag[j] = 1 where the sorted key differs from its predecessor (j > 0): the inclusive sum of it is the rank of row j
__global__ void build_flag_kernel(const uint64_t *__restrict__ skey, uint64_t m, uint32_t *__restrict__ flag)
{
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (j >= m) return;
    flag[j] = (j > 0 && skey[j] != skey[j - 1]) ? 1u : 0u;
}

// rank[suffix of row j] = rank of row j
__global__ void build_scatter_rank_kernel(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ row_rank, uint64_t m, uint32_t *__restrict__ rank)
{
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (j < m) rank[sidx[j]] = row_rank[j];
}

// next key of row j: (its rank, 1 + rank of the suffix h symbols further on -- 0 when that runs past the end)
__global__ void build_next_key_kernel(const uint32_t *__restrict__ sidx, const uint32_t *__restrict__ row_rank, const uint32_t *__restrict__ rank, uint64_t m,
                                      uint64_t h, uint64_t *__restrict__ key)
{
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t nxt = (uint64_t)sidx[j] + h;                       // m = n + 1 suffixes: indices 0 .. n
    const uint64_t lo = nxt < m ? (uint64_t)rank[nxt] + 1u : 0u;
    key[j] = ((uint64_t)row_rank[j] << 32) | lo;
}

// the row whose suffix is the whole text ('$' row of the BWT: BWT.c:605 leaves it out of the numbering)
__global__ void build_find_row0_kernel(const uint32_t *__restrict__ sa, uint64_t m, unsigned long long *__restrict__ out)
{
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (j < m && sa[j] == 0) *out = j;
}

// One thread per 16 BWT symbols: word w holds rows 16w .. 16w+15 of the BWT without the '$' row, first symbol in the top bits
// (the .bwt layout).  BWT of row j = the symbol in front of suffix sa[j].
__global__ void build_bwt_words_kernel(const uint8_t *__restrict__ codes, const uint32_t *__restrict__ sa, uint64_t n, uint64_t inv_sa0, uint32_t *__restrict__ words)
{
    const uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (w * 16 >= n) return;
    uint32_t v = 0;
#pragma unroll 4
    for (int t = 0; t < 16; t++) {
        const uint64_t q = w * 16 + (uint64_t)t;                      // position in the BWT without the '$' row
        uint32_t c = 0;
        if (q < n) {
            const uint64_t j = q + (q >= inv_sa0 ? 1u : 0u);          // its row in the suffix array
            c = codes[(uint64_t)sa[j] - 1u] & 3u;                     // sa[j] != 0 for every row but inv_sa0
        }
        v |= c << (2 * (15 - t));
    }
    words[w] = v;
}

// every `interval`-th entry of the suffix array, 64 bits wide (what hipapi.Index.from_arrays and the .sa writer take)
__global__ void build_sa_samples_kernel(const uint32_t *__restrict__ sa, uint64_t m, uint32_t interval, uint64_t n_samples, unsigned long long *__restrict__ out)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k < n_samples) out[k] = (unsigned long long)sa[k * (uint64_t)interval];
}

// .pac bytes: four bases per byte, first base in bits 7-6; symbol counts by the way
__global__ void build_pac_kernel(const uint8_t *__restrict__ codes, uint64_t n, uint8_t *__restrict__ pac, unsigned long long *__restrict__ counts /* [4] */)
{
    __shared__ unsigned int s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b * 4 < n) {
        uint32_t v = 0;
        unsigned int c4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint64_t p = b * 4 + (uint64_t)t;
            if (p < n) { const uint32_t c = codes[p] & 3u; v |= c << (2 * (3 - t)); c4[c]++; }
        }
        pac[b] = (uint8_t)v;
#pragma unroll
        for (int c = 0; c < 4; c++) if (c4[c]) atomicAdd(&s_cnt[c], c4[c]);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}
