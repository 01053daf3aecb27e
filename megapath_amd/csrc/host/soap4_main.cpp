// megapath_amd/csrc/host/soap4_main.cpp -- product code: the `soap4`-compatible executable of the MI355X path.
//
//   soap4-mi355x pair <indexPrefix> <r1.fq[.gz]> <r2.fq[.gz]> -o <outPrefix> -C <ini> -L <maxReadLen>
//                     [-T threads] [-u maxInsert] [-v minInsert] -F|-P [-nc] [-top N] [-b]
//
// Host side of the path in C++ above the C ABI (include/megapath_hip.h), mirroring the reference's process interface:
//   argument parsing        soap4/IniParam.cpp:542-942 (parseInputArgs), ini keys :242-438 (ParseIniFile)
//   read ingest / packing   soap4/QueryParser.cpp:177-262 (kseq records, trim_readno, appendToQueryArrays), IndexHandler.cpp:26-46
//   batch loop              soap4/SOAP4.cpp:445-766 (<= 2 097 152 reads per batch, insert_low adjusted on the first batch :465-474)
//   per-batch flow          soap4/alignment.cpp:29-355 (deep-DP round, single-end DP, mate rescue, unpaired output)
//   result selection        DV-DPfunctions.h:165-243 (sort + de-duplicate), OutputDPResult.cpp:65-265
//   annotated FASTQ         BGS-IO.cpp:1312-1446, 1943-2091
// All FM-index steps and DP cells run in the HIP kernels; this file only parses, packs, groups hits per pair and prints.
// A reader thread parses and packs the next batch while the GPU works on the current one (the reference's IO thread,
// aio_thread.cpp:247).  BAM side files (-b) are not written by this build (SURVEY 8a row a15 is "next").
#include <zlib.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/megapath_hip.h"

namespace {

const uint64_t NOT_ALIGNED = ~0ull;

struct Params {
    int max_read_length = 120, insert_low = 1, insert_high = 500;   // IniParam.cpp:627-648
    int clip_left = 130, clip_right = 130;
    int mode = 0;                 // 1 = -F, 2 = -P
    double top = 0.95;
    bool skip_default_dp = false, want_bam = false, ignore_comments = false;   // -nc: SOAP4.cpp:537-547
    mp_mmp_params mmp = {30, 17, 6, 5, 27, 18, 0.85, 4, 0.5};       // defaults of ParseIniFile (IniParam.cpp:427-435)
    size_t batch_pairs = 1u << 20;                                   // SOAP4.cpp:206
};

struct Rec {
    std::string name, comment, seq, qual;   // seq = decoded 2-bit codes (what the reference echoes: N -> G, truncated)
    bool has_comment = false;
};

struct Batch {
    std::vector<Rec> recs;                  // 2 per pair
    std::vector<uint32_t> packed, lens;
    uint32_t W = 0;
    bool last = false;
};

unsigned char g_code[256];
void fill_char_map()
{
    memset(g_code, 0, sizeof g_code);                         // INDEXFillCharMap: everything is 'A' unless listed
    const char *b = "ACGT";
    for (int i = 0; i < 4; i++) { g_code[(int)b[i]] = (unsigned char)i; g_code[(int)b[i] + 32] = (unsigned char)i; }
    g_code['U'] = g_code['u'] = 3;
    g_code['N'] = g_code['n'] = 2;
}

// ------------------------------------------------------------------------------------------------ ini
typedef std::map<std::string, std::string> Ini;     // "Section:key" -> value
bool load_ini(const std::string &path, Ini &ini)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[4096];
    std::string sec;
    while (fgets(line, sizeof line, f)) {
        std::string s(line);
        size_t c = s.find_first_of(";#");
        if (c != std::string::npos) s.erase(c);
        auto trim = [](std::string &t) { size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n"); t = a == std::string::npos ? "" : t.substr(a, b - a + 1); };
        trim(s);
        if (s.empty()) continue;
        if (s[0] == '[') { size_t e = s.find(']'); sec = s.substr(1, e == std::string::npos ? std::string::npos : e - 1); continue; }
        size_t eq = s.find('=');
        if (eq == std::string::npos) continue;
        std::string k = s.substr(0, eq), v = s.substr(eq + 1);
        trim(k); trim(v);
        ini[sec + ":" + k] = v;
    }
    fclose(f);
    return true;
}
int ini_int(const Ini &ini, const char *key, int dflt) { auto it = ini.find(key); return it == ini.end() ? dflt : atoi(it->second.c_str()); }
double ini_dbl(const Ini &ini, const char *key, double dflt) { auto it = ini.find(key); return it == ini.end() ? dflt : atof(it->second.c_str()); }

// ------------------------------------------------------------------------------------------------ FASTQ (plain or gz)
struct FqReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    bool open(const char *path) { f = gzopen(path, "rb"); if (f) gzbuffer(f, 1 << 20); buf.resize(1 << 16); return f != nullptr; }
    bool line(std::string &out)
    {
        out.clear();
        for (;;) {
            if (!gzgets(f, buf.data(), (int)buf.size())) return !out.empty();
            size_t n = strlen(buf.data());
            out.append(buf.data(), n);
            if (n && buf[n - 1] == '\n') break;
        }
        while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
        return true;
    }
    // one record; name loses a trailing /1 /2 (QueryParser.cpp:177), comment = rest of the header line (kseq)
    bool next(Rec &r, std::string &raw_seq)
    {
        std::string h, plus;
        do { if (!line(h)) return false; } while (h.empty());
        if (!line(raw_seq)) return false;
        bool fasta = h[0] == '>';
        r.qual.clear();
        if (!fasta) { line(plus); line(r.qual); }
        size_t cut = h.find_first_of(" \t", 1);
        r.name = h.substr(1, cut == std::string::npos ? std::string::npos : cut - 1);
        r.has_comment = cut != std::string::npos && cut + 1 < h.size();
        r.comment = r.has_comment ? h.substr(cut + 1) : std::string();
        size_t L = r.name.size();
        if (L > 2 && r.name[L - 2] == '/' && r.name[L - 1] >= '0' && r.name[L - 1] <= '9') r.name.resize(L - 2);
        return true;
    }
    ~FqReader() { if (f) gzclose(f); }
};

// reads the next batch and packs it as appendToQueryArrays does (QueryParser.cpp:184-234), words contiguous per read
bool fill_batch(FqReader &a, FqReader &b, const Params &p, Batch &out)
{
    out.recs.clear(); out.packed.clear(); out.lens.clear(); out.last = false;
    const uint32_t W = (uint32_t)(p.max_read_length + 15) / 16;
    out.W = W;
    std::string sa, sb;
    Rec ra, rb;
    while (out.recs.size() < 2 * p.batch_pairs) {
        bool ga = a.next(ra, sa), gb = b.next(rb, sb);
        if (ga != gb) { fprintf(stderr, "Error: number of sequences of pair-end files not matched.\n"); exit(1); }
        if (!ga) { out.last = true; break; }
        for (int k = 0; k < 2; k++) {
            Rec &r = k ? rb : ra;
            const std::string &raw = k ? sb : sa;
            size_t keep = std::min(raw.size(), (size_t)(p.max_read_length - 1));          // QueryParser.cpp:188
            size_t base = out.packed.size();
            out.packed.resize(base + W, 0u);
            r.seq.resize(keep);
            for (size_t i = 0; i < keep; i++) {
                unsigned c = g_code[(unsigned char)raw[i]];
                out.packed[base + i / 16] |= c << (2 * (i % 16));
                r.seq[i] = "ACGT"[c];
            }
            if (r.qual.size() > keep) r.qual.resize(keep);
            if (p.ignore_comments) { r.has_comment = false; r.comment.clear(); }
            out.lens.push_back((uint32_t)keep);
            out.recs.push_back(r);
        }
    }
    return !out.recs.empty();
}

// ------------------------------------------------------------------------------------------------ output
struct ChrHit { int chr, neg; bool operator<(const ChrHit &o) const { return chr != o.chr ? chr < o.chr : neg < o.neg; } };
struct HdrMap { int score; std::string text; };

// getMappingFromHeader (BGS-IO.cpp:1348-1371), quirks included (the record filter tests the header's best score)
int mapping_from_header(const Rec &r, std::vector<HdrMap> &v, double top, double scoreT)
{
    v.clear();
    if (!r.has_comment || r.comment == "IGNORE") return 0;
    const char *c = r.comment.c_str();
    if (r.comment.size() < 6) return 0;
    int score = atoi(c + 6);
    if (score < scoreT) return score;
    else if (scoreT < score * top) scoreT = score * top;
    const char *p = strchr(c + 6, ';');
    while (p && *(p + 1) != '\0') {
        const char *s = p + 1;
        const char *q = strchr(s, ';');
        if (!q) break;
        if (score >= scoreT) v.push_back(HdrMap{atoi(s), std::string(s, q)});
        p = q;
    }
    return score;
}

struct Writer {
    const mp_index *ix;
    const Params &p;
    std::string out;
    Writer(const mp_index *i, const Params &pp) : ix(i), p(pp) { out.reserve(1 << 22); }
    void flush() { fwrite(out.data(), 1, out.size(), stdout); out.clear(); }
    void seq_qual(const Rec &r) { out += r.seq; out += "\n+\n"; out += r.qual; out += '\n'; if (out.size() > (1u << 22)) flush(); }
    void score_line(const Rec &r, int best, const std::vector<ChrHit> &hits, const std::vector<HdrMap> &v)
    {
        out += '@'; out += r.name;
        if (r.has_comment && r.comment == "IGNORE") { out += "\tIGNORE\n"; return; }
        char tmp[64];
        snprintf(tmp, sizeof tmp, "\tSCORE:%d;", best); out += tmp;
        if (best > 0) {
            for (size_t i = 0; i < hits.size(); i++) {
                if (i > 0 && hits[i].chr == hits[i - 1].chr) continue;
                if (-hits[i].neg > 0 && -hits[i].neg >= best * p.top) {
                    snprintf(tmp, sizeof tmp, "%d,", -hits[i].neg); out += tmp;
                    out += mp_index_seq_name(ix, (uint32_t)hits[i].chr); out += ';';
                }
            }
        }
        for (const HdrMap &m : v) if (m.score >= best * p.top) { out += m.text; out += ';'; }
        out += '\n';
    }
    // decideTargetChr (BGS-IO.cpp:1312-1341)
    int target_chr(uint64_t pos, uint32_t len) const
    {
        uint64_t tp; uint32_t c1, c2;
        mp_index_chr_and_pos(ix, pos, &tp, &c1);
        mp_index_chr_and_pos(ix, pos + len - 1, &tp, &c2);
        return c1 == c2 ? (int)c1 : -1;
    }
    struct PairRes { uint64_t a1, a2; int s1, s2; };
    // pairDeepDPOutputFastqAPI (BGS-IO.cpp:1966-2091)
    void pair(const Rec &r1, const Rec &r2, uint32_t len1, uint32_t len2, const PairRes *res, size_t n)
    {
        int best1 = 0, best2 = 0;
        std::vector<ChrHit> h1, h2;
        for (size_t i = 0; i < n; i++) {
            uint64_t a1 = res[i].a1, a2 = res[i].a2; int s1 = res[i].s1, s2 = res[i].s2;
            int c1 = a1 != NOT_ALIGNED ? target_chr(a1, len1) : -1, c2 = a2 != NOT_ALIGNED ? target_chr(a2, len2) : -1;
            if (c1 == -1) { a1 = NOT_ALIGNED; s1 = 0; }
            if (c2 == -1) { a2 = NOT_ALIGNED; s2 = 0; }
            if (p.mode == 2 && (a1 == NOT_ALIGNED || a2 == NOT_ALIGNED)) { a1 = a2 = NOT_ALIGNED; s1 = s2 = 0; }
            if (c1 == c2 && a1 != NOT_ALIGNED && a2 != NOT_ALIGNED) s1 = s2 = s1 + s2;        // normalizeScore (:1943-1961)
            best1 = std::max(best1, s1); best2 = std::max(best2, s2);
            if (c1 != -1) h1.push_back(ChrHit{c1, -s1});
            if (c2 != -1) h2.push_back(ChrHit{c2, -s2});
        }
        std::sort(h1.begin(), h1.end()); std::sort(h2.begin(), h2.end());
        std::vector<HdrMap> v1, v2;
        int p1 = mapping_from_header(r1, v1, p.top, best1 * p.top), p2 = mapping_from_header(r2, v2, p.top, best2 * p.top);
        best1 = std::max(best1, p1); best2 = std::max(best2, p2);
        score_line(r1, best1, h1, v1); seq_qual(r1);
        score_line(r2, best2, h2, v2); seq_qual(r2);
    }
    // unproperlypairDPOutputFastqAPI (BGS-IO.cpp:1384-1446)
    void single(const Rec &r, uint32_t len, const std::vector<std::pair<uint64_t, int>> &hits_in)
    {
        if (r.has_comment && r.comment == "IGNORE") { out += '@'; out += r.name; out += "\tIGNORE\n"; seq_qual(r); return; }
        int best = 0;
        std::vector<ChrHit> h;
        if (p.mode != 2)
            for (auto &hs : hits_in) {
                int c = target_chr(hs.first, len), s = hs.second;
                if (c < 0) s = 0; else h.push_back(ChrHit{c, -s});
                best = std::max(best, s);
            }
        std::sort(h.begin(), h.end());
        std::vector<HdrMap> v;
        int prev = mapping_from_header(r, v, p.top, best * p.top);
        best = std::max(best, prev);
        score_line(r, best, h, v); seq_qual(r);
    }
};

void die(int rc, const char *what) { if (rc) { fprintf(stderr, "[soap4-mi355x] %s failed: %s (%d)\n", what, mp_strerror(rc), rc); exit(1); } }

struct Stats { uint64_t pairs = 0, deep = 0, rescued = 0, unpaired = 0, dp = 0; };

void align_batch(mp_index *ix, mp_ctx *ctx, Params &p, Batch &b, bool &first, Writer &w, Stats &st)
{
    const uint32_t n_reads = (uint32_t)b.recs.size(), n_pairs = n_reads / 2;
    if (first) {
        // SOAP4.cpp:465-474 / QueryParser.cpp:2253-2277: the minimum insert size is at least the detected read lengths
        uint32_t d1 = 0, d2 = 0;
        for (uint32_t q = 0; q < n_pairs && q < 999999; q++) { d1 = std::max(d1, b.lens[2 * q]); d2 = std::max(d2, b.lens[2 * q + 1]); }
        p.insert_low = std::max(p.insert_low, (int)std::max(d1, d2));
        first = false;
    }
    for (uint32_t l : b.lens) if (l < 30) { fprintf(stderr, "[soap4-mi355x] reads shorter than the DP cutoff (30) are undefined in the reference (CPU_DP.cpp:316-324)\n"); exit(1); }
    die(mp_set_reads(ctx, b.packed.data(), b.W, b.lens.data(), n_reads), "mp_set_reads");
    mp_align_params ap;
    ap.insert_low = p.insert_low; ap.insert_high = p.insert_high; ap.clip_left = p.clip_left; ap.clip_right = p.clip_right; ap.mmp = p.mmp;
    ap.want_tie_counts = 0; ap.reserved = 0;
    uint64_t nh = 0;
    mp_round_stats rs;
    die(mp_deep_dp_round(ctx, &ap, &nh, &rs), "mp_deep_dp_round");
    const mp_pair_hit *view = nullptr;
    die(mp_get_pair_hits_view(ctx, &view, &nh), "mp_get_pair_hits_view");
    std::vector<mp_pair_hit> hits(view, view + nh);
    st.pairs += n_pairs; st.dp += 2 * rs.n_candidates;
    // OutputBuffer::ready: per pair sort by (pos1, pos2, score1, score2) and drop duplicates (DV-DPfunctions.h:165-196)
    std::sort(hits.begin(), hits.end(), [](const mp_pair_hit &x, const mp_pair_hit &y) {
        if (x.pair_id != y.pair_id) return x.pair_id < y.pair_id;
        if (x.algnmt_1 != y.algnmt_1) return x.algnmt_1 < y.algnmt_1;
        if (x.algnmt_2 != y.algnmt_2) return x.algnmt_2 < y.algnmt_2;
        if (x.score_1 != y.score_1) return x.score_1 < y.score_1;
        return x.score_2 < y.score_2;
    });
    std::vector<char> done(n_pairs, 0);
    uint32_t n_done = 0;
    std::vector<Writer::PairRes> res;
    for (size_t i = 0; i < hits.size();) {
        size_t j = i;
        res.clear();
        while (j < hits.size() && hits[j].pair_id == hits[i].pair_id) {
            const mp_pair_hit &h = hits[j];
            if (res.empty() || res.back().a1 != h.algnmt_1 || res.back().a2 != h.algnmt_2 || res.back().s1 != h.score_1 || res.back().s2 != h.score_2)
                res.push_back(Writer::PairRes{h.algnmt_1, h.algnmt_2, h.score_1, h.score_2});
            j++;
        }
        uint32_t q = hits[i].pair_id;
        w.pair(b.recs[2 * q], b.recs[2 * q + 1], b.lens[2 * q], b.lens[2 * q + 1], res.data(), res.size());
        done[q] = 1; st.deep++; n_done++;
        i = j;
    }
    if (n_done == n_pairs) return;
    // ---- rescue passes (alignment.cpp:137-352): single-end DP + mate rescue on the device, bookkeeping here
    mp_rescue_stats rst;
    die(mp_rescue_round(ctx, &ap, p.skip_default_dp ? 1 : 0, &rst), "mp_rescue_round");
    st.dp += rst.n_single_tasks + rst.n_rescue_tasks;
    uint64_t ns = 0, nr = 0;
    die(mp_get_single_hits(ctx, nullptr, 0, &ns), "mp_get_single_hits");
    std::vector<mp_single_hit> singles(ns + 1);
    die(mp_get_single_hits(ctx, singles.data(), ns, &ns), "mp_get_single_hits");
    die(mp_get_rescue_hits(ctx, nullptr, 0, &nr), "mp_get_rescue_hits");
    std::vector<mp_rescue_hit> resc(nr + 1);
    die(mp_get_rescue_hits(ctx, resc.data(), nr, &nr), "mp_get_rescue_hits");
    std::map<uint32_t, std::vector<std::pair<uint64_t, int>>> single;
    for (uint64_t i = 0; i < ns; i++) single[singles[i].read_id].emplace_back(singles[i].algnmt, singles[i].score);
    std::map<uint32_t, std::vector<Writer::PairRes>> rescued;
    for (uint64_t i = 0; i < nr; i++) {
        const mp_rescue_hit &h = resc[i];
        if ((h.aligned_read & 1u) == 0) rescued[h.pair_id].push_back(Writer::PairRes{h.aligned_pos, h.dp_pos, h.aligned_score, h.dp_score});
        else rescued[h.pair_id].push_back(Writer::PairRes{h.dp_pos, h.aligned_pos, h.dp_score, h.aligned_score});
    }
    auto pr_less = [](const Writer::PairRes &x, const Writer::PairRes &y) {
        if (x.a1 != y.a1) return x.a1 < y.a1;
        if (x.a2 != y.a2) return x.a2 < y.a2;
        if (x.s1 != y.s1) return x.s1 < y.s1;
        return x.s2 < y.s2;
    };
    for (uint32_t q = 0; q < n_pairs; q++) {
        if (done[q]) continue;
        auto it = rescued.find(q);
        if (it != rescued.end()) {
            auto &v = it->second;                           // OutputBuffer::ready(1): sorted, duplicates and failed rescues dropped
            std::sort(v.begin(), v.end(), pr_less);
            v.erase(std::unique(v.begin(), v.end(), [](const Writer::PairRes &x, const Writer::PairRes &y) { return x.a1 == y.a1 && x.a2 == y.a2 && x.s1 == y.s1 && x.s2 == y.s2; }), v.end());
            w.pair(b.recs[2 * q], b.recs[2 * q + 1], b.lens[2 * q], b.lens[2 * q + 1], v.data(), v.size());
            st.rescued++;
        } else {
            for (uint32_t r = 2 * q; r < 2 * q + 2; r++) {
                std::vector<std::pair<uint64_t, int>> hs;
                auto is = single.find(r);
                if (is != single.end()) { hs = is->second; std::sort(hs.begin(), hs.end()); hs.erase(std::unique(hs.begin(), hs.end()), hs.end()); }
                w.single(b.recs[r], b.lens[r], hs);
            }
            st.unpaired++;
        }
    }
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 5 || strcmp(argv[1], "pair") != 0) {
        fprintf(stderr, "usage: %s pair <indexPrefix> <r1.fq[.gz]> <r2.fq[.gz]> -o <outPrefix> -C <ini> -L <maxReadLen> [-T n] [-u maxInsert] [-v minInsert] -F|-P [-nc] [-top N] [-b]\n", argv[0]);
        return 1;
    }
    Params p;
    std::string ini_path;
    int cli_device = -1;
    const char *prefix = argv[2], *fq1 = argv[3], *fq2 = argv[4];
    for (int i = 5; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
        // arities follow parseInputArgs (IniParam.cpp:651-896)
        if (a == "-o" || a == "-T" || a == "-D" || a == "-A" || a == "-R" || a == "-e") (void)val();
        else if (a == "-h") { int t = atoi(val()); if (t < 1 || t > 4) { fprintf(stderr, "The output option should be 1, 2, 3 or 4\n"); return 1; } }
        else if (a == "-s") { if (i + 1 < argc && isdigit((unsigned char)argv[i + 1][0])) i++; }   // IniParam.cpp:738-757
        else if (a == "-c") { cli_device = atoi(val()); if (cli_device < 0) { fprintf(stderr, "The GPU device ID should not be less than 0\n"); return 1; } }
        else if (a == "-p" || a == "-I") {}                                                          // flags without a value
        else if (a == "-C") ini_path = val();
        else if (a == "-L" || a == "-l") p.max_read_length = atoi(val());
        else if (a == "-u") p.insert_high = atoi(val());
        else if (a == "-v") p.insert_low = atoi(val());
        else if (a == "-top") p.top = atoi(val()) / 100.0;
        else if (a == "-F") p.mode = 1;
        else if (a == "-P") p.mode = 2;
        else if (a == "-b") p.want_bam = true;
        else if (a == "-nc") p.ignore_comments = true;
    }
    if (p.insert_low > p.insert_high) { fprintf(stderr, "The minimum value of insert size should not be greater than the maximum value of insert size.\n"); return 1; }
    if (p.mode == 0) { fprintf(stderr, "[soap4-mi355x] only MegaPath modes -F / -P are implemented (plain mode segfaults in the reference, SURVEY App. D15)\n"); return 1; }
    if (ini_path.empty()) ini_path = std::string(argv[0]) + ".ini";       // SOAP4.cpp:150-154
    Ini ini;
    if (!load_ini(ini_path, ini)) { fprintf(stderr, "[soap4-mi355x] cannot read ini file %s\n", ini_path.c_str()); return 1; }
    if (ini_int(ini, "DP:MatchScore", 1) != 1 || ini_int(ini, "DP:MismatchScore", -2) != -2 || ini_int(ini, "DP:GapOpenScore", -3) != -3 ||
        ini_int(ini, "DP:GapExtendScore", -1) != -1) { fprintf(stderr, "only the shipped DP scores (+1/-2/-3/-1) are compiled into the kernels\n"); return 1; }
    { auto it = ini.find("PairEnd:StrandArrangement"); if (it != ini.end() && it->second != "+/-") { fprintf(stderr, "only StrandArrangement=+/- is supported\n"); return 1; } }
    if (ini_int(ini, "OtherSettings:SkipSOAP3Alignment", 1) != 1) { fprintf(stderr, "SkipSOAP3Alignment must be 1 (alignment.cpp:50-67)\n"); return 1; }
    p.clip_left = ini_int(ini, "Clipping:MaxFrontLenClipped", p.clip_left);
    p.clip_right = ini_int(ini, "Clipping:MaxEndLenClipped", p.clip_right);
    p.skip_default_dp = ini_int(ini, "OtherSettings:SkipDefaultDP", 0) != 0;
    p.mmp.sa_size_threshold = ini_int(ini, "MMP:mmpSeedSAsizeThreshold", 30); p.mmp.seed_min_length = ini_int(ini, "MMP:mmpSeedMinLength", 17);
    p.mmp.uniq_threshold = ini_int(ini, "MMP:mmpUniqThreshold", 6); p.mmp.indel_fuzz = ini_int(ini, "MMP:mmpIndelFuzz", 5);
    p.mmp.good_seed_len = ini_int(ini, "MMP:mmpGoodSeedLen", 27); p.mmp.reseed_len = ini_int(ini, "MMP:mmpReseedLen", 18);
    p.mmp.reseed_rlt_ratio = ini_dbl(ini, "MMP:mmpReseedRLTratio", 0.85); p.mmp.reseed_abs_diff = ini_int(ini, "MMP:mmpReseedAbsDiff", 4);
    p.mmp.short_seed_ratio = ini_dbl(ini, "MMP:mmpShortSeedRatio", 0.5);
    if (const char *e = getenv("MP_BATCH_PAIRS")) p.batch_pairs = (size_t)atol(e);
    if (p.want_bam) fprintf(stderr, "[soap4-mi355x] -b: BAM output is not written by this build (annotated FASTQ only)\n");
    int device = 0;
    if (cli_device >= 0) device = cli_device;                              // -c <GPU device id>, IniParam.cpp:822-838
    else if (const char *e = getenv("MP_DEVICE")) device = atoi(e); else if (const char *e2 = getenv("LOCAL_RANK")) device = atoi(e2);
    fill_char_map();

    mp_index *ix = nullptr;
    die(mp_index_open(prefix, device, &ix), "mp_index_open");
    mp_index_info info;
    mp_index_get_info(ix, &info);
    fprintf(stderr, "[Main] index %s in HBM: %.2f GB, %u sequences\n", prefix, info.device_bytes / 1e9, info.num_seq);
    mp_ctx *ctx = nullptr;
    die(mp_ctx_create(ix, &ctx), "mp_ctx_create");

    FqReader ra, rb;
    if (!ra.open(fq1) || !rb.open(fq2)) { fprintf(stderr, "[soap4-mi355x] cannot open read files\n"); return 1; }
    // double buffer: the reader thread parses + packs batch k+1 while the GPU aligns batch k (aio_thread.cpp:247)
    Batch bufs[2];
    std::mutex mu; std::condition_variable cv;
    int ready[2] = {0, 0};           // 0 = free, 1 = filled
    bool reader_done = false;
    std::thread reader([&]() {
        for (int k = 0;; k ^= 1) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[k] == 0; }); }
            bool got = fill_batch(ra, rb, p, bufs[k]);
            { std::lock_guard<std::mutex> lk(mu); if (got) ready[k] = 1; if (!got || bufs[k].last) reader_done = true; }
            cv.notify_all();
            if (!got || bufs[k].last) break;
        }
    });
    Writer w(ix, p);
    Stats st;
    bool first = true;
    for (int k = 0;; k ^= 1) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[k] == 1 || reader_done; }); if (ready[k] != 1) break; }
        align_batch(ix, ctx, p, bufs[k], first, w, st);
        bool last = bufs[k].last;
        { std::lock_guard<std::mutex> lk(mu); ready[k] = 0; }
        cv.notify_all();
        if (last) break;
    }
    reader.join();
    w.flush();
    fflush(stdout);
    fprintf(stderr, "[Main] %llu pairs: %llu aligned by deep DP, %llu by mate rescue, %llu unpaired; %llu DP instances\n", (unsigned long long)st.pairs,
            (unsigned long long)st.deep, (unsigned long long)st.rescued, (unsigned long long)st.unpaired, (unsigned long long)st.dp);
    mp_ctx_destroy(ctx);
    mp_index_close(ix);
    return 0;
}
