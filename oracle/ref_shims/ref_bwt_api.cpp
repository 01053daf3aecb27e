// oracle/ref_shims/ref_bwt_api.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Plain-C entry points around the *real* reference 2BWT query primitives, compiled from the reference
// sources where they lie (oracle/Makefile.ref).  Wrapped functions:
//   BWTLoad            soap4/2bwt-lib/BWT.c:100
//   BWTOccValue        soap4/2bwt-lib/BWT.c:597
//   BWTPsiMinusValue   soap4/2bwt-lib/BWT.c:915
//   BWTSaValue         soap4/2bwt-lib/BWT.c:968
//   LTLoad             soap4/2bwt-flex/LT.c:37   (13-mer lookup table; use as in DV-DPfunctions.cpp:2245-2246)
#include <stdio.h>
#include <stdlib.h>
#include <string>
extern "C" {
}
#include "MemManager.h"
#include "BWT.h"
#include "LT.h"

struct RefIndex { MMPool *pool; BWT *bwt; LT *lkt; };

extern "C" void *ref_index_open(const char *prefix, int loadLkt)
{
    static int inited = 0;
    if (!inited) { MMMasterInitialize(3, 0, FALSE, NULL); inited = 1; }
    RefIndex *ix = new RefIndex();
    ix->pool = MMPoolCreate(2097152);
    std::string p(prefix);
    ix->bwt = BWTLoad(ix->pool, 0, (p + ".bwt").c_str(), (p + ".fmv").c_str(), (p + ".sa").c_str());
    ix->lkt = loadLkt ? LTLoad((p + ".lkt").c_str()) : NULL;
    return ix;
}
extern "C" unsigned long long ref_text_length(void *h) { return ((RefIndex *)h)->bwt->textLength; }
extern "C" unsigned long long ref_inverse_sa0(void *h) { return ((RefIndex *)h)->bwt->inverseSa0; }
extern "C" unsigned long long ref_sa_interval(void *h) { return ((RefIndex *)h)->bwt->saInterval; }
extern "C" unsigned long long ref_cum_freq(void *h, int c) { return ((RefIndex *)h)->bwt->cumulativeFreq[c]; }
extern "C" unsigned long long ref_occ(void *h, unsigned long long idx, unsigned c)
{ return BWTOccValue(((RefIndex *)h)->bwt, idx, c); }
extern "C" unsigned long long ref_psi_minus(void *h, unsigned long long idx)
{ return BWTPsiMinusValue(((RefIndex *)h)->bwt, idx); }
extern "C" unsigned long long ref_sa_value(void *h, unsigned long long idx)
{ return BWTSaValue(((RefIndex *)h)->bwt, idx); }
extern "C" void ref_occ_many(void *h, const unsigned long long *idx, const unsigned *c, unsigned long long n, unsigned long long *out)
{ for (unsigned long long i = 0; i < n; i++) out[i] = BWTOccValue(((RefIndex *)h)->bwt, idx[i], c[i]); }
extern "C" void ref_sa_many(void *h, const unsigned long long *idx, unsigned long long n, unsigned long long *out)
{ for (unsigned long long i = 0; i < n; i++) out[i] = BWTSaValue(((RefIndex *)h)->bwt, idx[i]); }
// SA range of a 13-mer key exactly as the reference seeds it (DV-DPfunctions.cpp:2245-2246)
extern "C" void ref_lkt_range(void *h, unsigned key, unsigned long long *l, unsigned long long *r)
{
    LT *lkt = ((RefIndex *)h)->lkt;
    *l = key ? lkt->table[key - 1] + 1 : 1;
    *r = lkt->table[key];
}
