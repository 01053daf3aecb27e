// megapath_amd/csrc/kernels_cov.hip -- product code: per-reference coverage and the spike filter on the device.
//
// The reference gets there through files (runMegaPath.sh:211-222): samtools merge/sort of the BAM side files -> bedtools genomecov
// -bga (depth per base of every sequence) -> cc/genomeCovFilter (per sequence: mean and population variance of the depth over all
// its bases, cc/genomeCovFilter.cpp:58-84; "spike" = depth > mean + k * stdev) -> bedtools bamtobed + annotate (fraction of every
// alignment that lies on spike bases) -> the names of the reads to drop.  Here the alignments the BAM files would hold are
// accumulated as +1 / -1 events in a difference array over the sequences' own coordinates; one prefix sum gives the depth, one
// segmented reduction per sequence the two sums, a second prefix sum over the spike mask answers every alignment with two loads.
// Streaming kernels over 4 bytes per base: HBM-bound, no re-reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

// one alignment = one interval [start, start + len) of sequence `seq` (0-based ids; offsets of the sequences in `seq_off`)
__global__ void cov_add_kernel(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, const uint32_t *__restrict__ len, uint64_t n,
                               const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ seq_len, int32_t *__restrict__ diff)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t s = seq[t];
    uint64_t a = start[t], b = a + len[t];
    const uint64_t L = seq_len[s];
    if (a > L) a = L;
    if (b > L) b = L;                                 // an alignment never leaves its sequence (the BAM writer trims it)
    if (b <= a) return;
    atomicAdd(&diff[seq_off[s] + a], 1);
    atomicAdd(&diff[seq_off[s] + b], -1);             // the slot behind a sequence's last base belongs to it (seq_off has a gap of 1)
}

// per sequence: sum and sum of squares of the depth over its bases; one block per (sequence, slice)
__global__ void __launch_bounds__(256)
cov_moments_kernel(const int32_t *__restrict__ depth, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ seq_len, uint32_t n_seq,
                   uint32_t slices, unsigned long long *__restrict__ sums /* [n_seq][2] */)
{
    const uint32_t s = blockIdx.x / slices, sl = blockIdx.x % slices;
    if (s >= n_seq) return;
    const uint64_t L = seq_len[s], per = (L + slices - 1) / slices, lo = (uint64_t)sl * per, hi = lo + per < L ? lo + per : L;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) { const unsigned long long d = (unsigned long long)(uint32_t)depth[seq_off[s] + i]; s1 += d; s2 += d * d; }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
    __shared__ unsigned long long a1[4], a2[4];
    if ((threadIdx.x & 63) == 0) { a1[threadIdx.x >> 6] = s1; a2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s1 = a1[0] + a1[1] + a1[2] + a1[3]; s2 = a2[0] + a2[1] + a2[2] + a2[3];
        if (s1) atomicAdd(&sums[2 * (size_t)s], s1);
        if (s2) atomicAdd(&sums[2 * (size_t)s + 1], s2);
    }
}

// mask[i] = depth[i] > max_depth of its sequence (as the comparison of an int with a double in cc/genomeCovFilter.cpp:89);
// the slot behind a sequence's last base is 0
__global__ void cov_mask_kernel(const int32_t *__restrict__ depth, const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ seq_len,
                                const double *__restrict__ max_depth, uint32_t n_seq, uint64_t total, uint32_t *__restrict__ mask)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    // sequence of slot i: binary search over the offsets
    uint32_t lo = 0, hi = n_seq - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (seq_off[mid] <= i) lo = mid; else hi = mid - 1; }
    const uint64_t at = i - seq_off[lo];
    mask[i] = (at < seq_len[lo] && (double)depth[i] > max_depth[lo]) ? 1u : 0u;
}

// spike bases under every alignment: two loads of the inclusive prefix sum of the mask
__global__ void cov_query_kernel(const uint32_t *__restrict__ seq, const uint64_t *__restrict__ start, const uint32_t *__restrict__ len, uint64_t n,
                                 const uint64_t *__restrict__ seq_off, const uint64_t *__restrict__ seq_len, const uint32_t *__restrict__ mask_incl,
                                 uint32_t *__restrict__ out_covered)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t s = seq[t];
    uint64_t a = start[t], b = a + len[t];
    const uint64_t L = seq_len[s];
    if (a > L) a = L;
    if (b > L) b = L;
    uint32_t c = 0;
    if (b > a) {
        const uint64_t base = seq_off[s];
        c = mask_incl[base + b - 1] - (base + a ? mask_incl[base + a - 1] : 0u);
    }
    out_covered[t] = c;
}

// starts (mask 0 -> 1) and ends (1 -> 0) of the spike runs, as global slots, in no particular order
__global__ void cov_edges_kernel(const uint32_t *__restrict__ mask, uint64_t total, unsigned long long *__restrict__ out, uint32_t cap, unsigned int *__restrict__ n_out)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t cur = mask[i], prev = i ? mask[i - 1] : 0u;
    if (cur != prev) {
        const unsigned int k = atomicAdd(n_out, 1u);
        if (k < cap) out[k] = (unsigned long long)i << 1 | cur;          // low bit: 1 = a run starts here, 0 = it ended before here
    }
}
