// oracle/ref_shims/ref_dp_api.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Exposes the *real* reference DP kernel `callDP` (reference soap4/CPU_DP.cpp:881-978) behind a plain C
// entry point so tests can drive it from ctypes.  The reference translation unit is #included from
// where it lies under /root/reference (via -I on the compiler command line, see oracle/Makefile.ref);
// nothing of it is copied here.  This shim only allocates the scratch areas the reference expects
// (DPTable / DPTempRow sized as in soap4/CPU_DPfunctions.cpp:251-262) and pads the per-instance
// arrays to a multiple of the SIMD width because callDP writes the padding lanes (CPU_DP.cpp:923-927).
#include "CPU_DP.cpp"
#include <stdlib.h>
#include <vector>

extern "C" int ref_dp_lanes(void) { return N_DP_PACKED; }

// Same array contracts as callDP: 32-instance interleaved, MSB-first 2-bit packed sequences,
// symbol i (1-based) of instance t at word  (t/32)*32*W + t%32 + 32*(i/16), bits 2*(15 - i%16).
extern "C" int ref_callDP(const uint *dnaPacked, const uint *dnaLen, uint maxDNALength,
                          const uint *readPacked, const uint *readLen, uint maxReadLength,
                          int clipLt, int clipRt, int mismatch, int gapOpen,
                          const int *cutoff, uint n,
                          int *scores, uint *hitLocs, uint *maxScoreCounts, uchar *pattern)
{
    uint padded = ((n + N_DP_PACKED - 1) / N_DP_PACKED) * N_DP_PACKED;
    std::vector<uint> dl(padded), rl(padded), hl(padded), mc(padded);
    std::vector<int> co(padded), sc(padded);
    for (uint i = 0; i < n; i++) { dl[i] = dnaLen[i]; rl[i] = readLen[i]; co[i] = cutoff[i]; }
    void *table = NULL; void *temp = NULL;
    size_t tableBytes = (size_t)(maxDNALength + 1) * (maxReadLength + 1) * sizeof(SIMDtype);
    if (posix_memalign(&table, 64, tableBytes)) return -1;
    if (posix_memalign(&temp, 64, (size_t)(maxReadLength + 2) * sizeof(SIMDtype))) return -1;
    memset(temp, 0, (size_t)(maxReadLength + 2) * sizeof(SIMDtype));
    std::vector<uchar> pat((size_t)padded * (maxDNALength + maxReadLength), 0);
    callDP((uint *)dnaPacked, dl.data(), maxDNALength, (uint *)readPacked, rl.data(), maxReadLength,
           clipLt, clipRt, NULL, NULL, 1, (uint)mismatch, (uint)gapOpen, (uint)-1,
           co.data(), table, (SIMDtype *)temp, n, sc.data(), hl.data(), mc.data(), pat.data());
    for (uint i = 0; i < n; i++) { scores[i] = sc[i]; hitLocs[i] = hl[i]; maxScoreCounts[i] = mc[i]; }
    memcpy(pattern, pat.data(), (size_t)n * (maxDNALength + maxReadLength));
    free(table); free(temp);
    return 0;
}
