// megapath_amd/csrc/index_host.cpp -- product code: reads the 2BWT index files and re-lays them out for HBM.
//
// Replaces the load side of the reference index stack:
//   BWTLoad            soap4/2bwt-lib/BWT.c:100-243      (.bwt .fmv .sa headers and payloads)
//   DNALoadPacked      soap4/2bwt-lib/TextConverter.c:427-499 (.pac)
//   HSPLoad            soap4/2bwt-lib/HSP.c:57-300       (.ann, .tra)
//   LTLoad             soap4/2bwt-flex/LT.c:37-57        (.lkt; rebuilt from the text as LTConstruct.c:44-105 when absent)
// The explicit-occurrence file (.fmv) is only header-checked: the 64-byte rank blocks carry their own running counts,
// recomputed on the device (api.hip: upload_index) together with the LSB-first text and, when no .lkt file exists,
// the 13-mer table.  This file only parses; no per-base loop runs on the host.
#include "mp_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>

namespace {

struct File {
    FILE *f = nullptr;
    explicit File(const std::string &p, const char *mode = "rb") { f = fopen(p.c_str(), mode); }
    ~File() { if (f) fclose(f); }
    bool read(void *dst, size_t sz, size_t cnt) { return cnt == 0 || fread(dst, sz, cnt, f) == cnt; }
};

int finish(mp_host_index &out, std::vector<uint32_t> &bwt_words, std::vector<uint8_t> &pac, const uint64_t *lkt64, std::string &err)
{
    if (out.n >= 0xFFFFFFF0ull) { err = "text length must be below 2^32 (split the database into chunks as the reference does)"; return MP_ERANGE; }
    out.bwt_words.swap(bwt_words);
    out.pac.swap(pac);
    if (lkt64) {
        out.lkt.resize(MP_LKT_ENTRIES);
        for (uint64_t k = 0; k < MP_LKT_ENTRIES; k++) out.lkt[k] = (uint32_t)lkt64[k];
    }
    return MP_OK;
}

}  // namespace

int mp_host_index_load(const char *prefix_c, mp_host_index &out, std::string &err)
{
    const std::string prefix(prefix_c);
    uint64_t hdr[5];
    std::vector<uint32_t> bwt_words;
    {
        File f(prefix + ".bwt");
        if (!f.f || !f.read(hdr, 8, 5)) { err = "cannot read " + prefix + ".bwt"; return MP_EIO; }
        out.inv_sa0 = hdr[0];
        out.cum[0] = 0;
        for (int i = 1; i <= 4; i++) out.cum[i] = hdr[i];
        out.n = out.cum[4];
        uint64_t words = (out.n + 15) / 16;
        bwt_words.assign(words + 16, 0u);
        if (!f.read(bwt_words.data(), 4, words)) { err = "short " + prefix + ".bwt"; return MP_EIO; }
    }
    {
        File f(prefix + ".fmv");
        uint64_t h2[5];
        if (!f.f || !f.read(h2, 8, 5) || h2[0] != out.inv_sa0 || h2[4] != out.n) { err = "bad " + prefix + ".fmv"; return MP_EIO; }
    }
    {
        File f(prefix + ".sa");
        uint64_t h2[5], interval = 0;
        if (!f.f || !f.read(h2, 8, 5) || !f.read(&interval, 8, 1) || h2[0] != out.inv_sa0 || interval == 0) { err = "bad " + prefix + ".sa"; return MP_EIO; }
        out.sa_interval = (uint32_t)interval;
        uint64_t cnt = (out.n + interval) / interval;
        std::vector<uint64_t> sa64(cnt);
        if (!f.read(sa64.data(), 8, cnt)) { err = "short " + prefix + ".sa"; return MP_EIO; }
        out.sa.resize(cnt + 1);
        for (uint64_t k = 0; k < cnt; k++) out.sa[k] = (uint32_t)sa64[k];
    }
    std::vector<uint8_t> pac;
    {
        File f(prefix + ".pac");
        if (!f.f) { err = "cannot read " + prefix + ".pac"; return MP_EIO; }
        pac.assign(out.n / 4 + 8, 0);
        size_t got = fread(pac.data(), 1, out.n / 4 + 2, f.f);
        if (got < out.n / 4 + 1) { err = "short " + prefix + ".pac"; return MP_EIO; }
    }
    {
        File f(prefix + ".ann", "r");
        unsigned long long n2; int ns; unsigned seed;
        if (!f.f || fscanf(f.f, "%llu %d %u\n", &n2, &ns, &seed) != 3 || n2 != out.n || ns <= 0) { err = "bad " + prefix + ".ann"; return MP_EIO; }
        for (int s = 0; s < ns; s++) {
            unsigned gi; char line[4096]; unsigned long long a, b; int z;
            if (fscanf(f.f, "%u ", &gi) != 1 || !fgets(line, sizeof line, f.f)) { err = "bad .ann entry"; return MP_EIO; }
            size_t L = strlen(line);
            if (L && line[L - 1] == '\n') line[L - 1] = 0;
            if (fscanf(f.f, "%llu %llu %d\n", &a, &b, &z) != 3) { err = "bad .ann offsets"; return MP_EIO; }
            out.seq_names.emplace_back(line);
            out.seq_start.push_back(a);
            out.seq_len.push_back(b);
        }
    }
    {
        File f(prefix + ".tra", "r");
        unsigned long long n2; int ns; unsigned rem, grid;
        if (!f.f || fscanf(f.f, "%llu %d %u %u\n", &n2, &ns, &rem, &grid) != 4) { err = "bad " + prefix + ".tra"; return MP_EIO; }
        out.grid.resize(grid);
        for (unsigned g = 0; g < grid; g++) { unsigned v; if (fscanf(f.f, "%u\n", &v) != 1) { err = "bad .tra grid"; return MP_EIO; } out.grid[g] = v; }
        out.translate.resize((size_t)ns + rem);
        for (auto &row : out.translate) {
            unsigned long long a, c; unsigned b;
            if (fscanf(f.f, "%llu %u %llu\n", &a, &b, &c) != 3) { err = "bad .tra row"; return MP_EIO; }
            row.start = a; row.chr = b; row.corr = c;
        }
        out.n_removed = rem;
        out.seq_actual_len = out.seq_len;                       // HSP.c:286-302: older indexes lack the tail rows
        for (int s = 0; s < ns && s < (int)out.seq_actual_len.size(); s++) {
            unsigned long long a, b;
            if (fscanf(f.f, "%llu %llu\n", &a, &b) != 2) break;
            out.seq_actual_len[s] = b;
        }
    }
    std::unique_ptr<uint64_t[]> lkt64;
    {
        File f(prefix + ".lkt");
        if (f.f) {
            int k = 0;
            lkt64.reset(new uint64_t[MP_LKT_ENTRIES]);
            if (!f.read(&k, 4, 1) || k != MP_LKT_K || !f.read(lkt64.get(), 8, MP_LKT_ENTRIES)) { err = "bad " + prefix + ".lkt (lookup table size must be 13)"; return MP_EIO; }
        }
    }
    return finish(out, bwt_words, pac, lkt64.get(), err);
}

int mp_host_index_from_arrays(uint64_t n, uint64_t inv_sa0, const uint64_t cum[5], const uint32_t *bwt_words,
                              const uint64_t *sa_samples, uint32_t sa_interval, const uint8_t *pac, const uint64_t *lkt,
                              uint32_t num_seq, const char *const *names, const uint64_t *starts, const uint64_t *lens,
                              mp_host_index &out, std::string &err)
{
    if (!bwt_words || !sa_samples || !pac || !sa_interval || num_seq == 0) { err = "null argument"; return MP_EINVAL; }
    out.n = n; out.inv_sa0 = inv_sa0;
    for (int i = 0; i < 5; i++) out.cum[i] = cum[i];
    out.sa_interval = sa_interval;
    uint64_t cnt = (n + sa_interval) / sa_interval;
    out.sa.resize(cnt + 1);
    for (uint64_t k = 0; k < cnt; k++) out.sa[k] = (uint32_t)sa_samples[k];
    // single-segment translate table per sequence (what the reference builder writes when no N-runs were removed)
    out.grid.assign((size_t)(n >> 18) + 1, 0u);
    for (uint32_t s = 0; s < num_seq; s++) {
        out.seq_names.emplace_back(names[s]);
        out.seq_start.push_back(starts[s]);
        out.seq_len.push_back(lens[s]);
        out.seq_actual_len.push_back(lens[s]);
        mp_translate_row row;
        row.start = starts[s]; row.chr = s + 1; row.corr = starts[s] - 1;   // HSP.c:679-690: tp = ambPos - correction, 1-based
        out.translate.push_back(row);
    }
    for (size_t g = 0; g < out.grid.size(); g++) {
        uint64_t pos = ((uint64_t)g << 18) + (1u << 18) - 1;   // last position of the grid cell: the scan walks downwards
        uint32_t v = 0;
        while (v + 1 < num_seq && starts[v + 1] <= pos) v++;
        out.grid[g] = v;
    }
    std::vector<uint32_t> bw(bwt_words, bwt_words + (n + 15) / 16);
    bw.resize(bw.size() + 16, 0u);
    std::vector<uint8_t> pc(pac, pac + (n + 3) / 4);
    pc.resize(pc.size() + 8, 0);
    return finish(out, bw, pc, lkt, err);
}
