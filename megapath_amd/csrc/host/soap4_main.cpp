// megapath_amd/csrc/host/soap4_main.cpp -- product code: the `soap4`-compatible executable of the MI355X path.
//
//   soap4-mi355x pair <indexPrefix> <r1.fq[.gz]> <r2.fq[.gz]> -o <outPrefix> -C <ini> -L <maxReadLen>
//                     [-T threads] [-u maxInsert] [-v minInsert] -F|-P [-nc] [-top N] [-b [-p] [-D rg -A sample [-R opt]]] [-c device]
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
// aio_thread.cpp:247).  With -b the BAM side files are written too (bam_out.h).
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
#include <memory>
#include "../../../include/megapath_hip.h"
#include "bam_out.h"

namespace {

using mpbam::NOT_ALIGNED;
using mpbam::PairAln;
using mpbam::Side;

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
    // pairDeepDPOutputFastqAPI (BGS-IO.cpp:1966-2091).  Like the reference it rewrites the results in place (boundary
    // spanning mates become unaligned, scores are normalised): the BAM writer that runs next sees those values.
    void pair(const Rec &r1, const Rec &r2, uint32_t len1, uint32_t len2, std::vector<PairAln> &res)
    {
        int best1 = 0, best2 = 0;
        h1_.clear(); h2_.clear();
        for (PairAln &r : res) {
            uint64_t &a1 = r.m[0].pos, &a2 = r.m[1].pos; int &s1 = r.m[0].score, &s2 = r.m[1].score;
            int c1 = a1 != NOT_ALIGNED ? target_chr(a1, len1) : -1, c2 = a2 != NOT_ALIGNED ? target_chr(a2, len2) : -1;
            if (c1 == -1) { a1 = NOT_ALIGNED; s1 = 0; }
            if (c2 == -1) { a2 = NOT_ALIGNED; s2 = 0; }
            if (p.mode == 2 && (a1 == NOT_ALIGNED || a2 == NOT_ALIGNED)) { a1 = a2 = NOT_ALIGNED; s1 = s2 = 0; }
            if (c1 == c2 && a1 != NOT_ALIGNED && a2 != NOT_ALIGNED) s1 = s2 = s1 + s2;        // normalizeScore (:1943-1961)
            best1 = std::max(best1, s1); best2 = std::max(best2, s2);
            if (c1 != -1) h1_.push_back(ChrHit{c1, -s1});
            if (c2 != -1) h2_.push_back(ChrHit{c2, -s2});
        }
        std::sort(h1_.begin(), h1_.end()); std::sort(h2_.begin(), h2_.end());
        std::vector<HdrMap> v1, v2;
        int p1 = mapping_from_header(r1, v1, p.top, best1 * p.top), p2 = mapping_from_header(r2, v2, p.top, best2 * p.top);
        best1 = std::max(best1, p1); best2 = std::max(best2, p2);
        score_line(r1, best1, h1_, v1); seq_qual(r1);
        score_line(r2, best2, h2_, v2); seq_qual(r2);
    }
    // unproperlypairDPOutputFastqAPI (BGS-IO.cpp:1384-1446); in -F mode hits that span a sequence boundary are un-aligned in place
    void single(const Rec &r, uint32_t len, std::vector<Side> &hits)
    {
        if (r.has_comment && r.comment == "IGNORE") { out += '@'; out += r.name; out += "\tIGNORE\n"; seq_qual(r); return; }
        int best = 0;
        h1_.clear();
        if (p.mode != 2)
            for (Side &hs : hits) {
                int c = target_chr(hs.pos, len);
                if (c < 0) { hs.pos = NOT_ALIGNED; hs.score = 0; } else h1_.push_back(ChrHit{c, -hs.score});
                best = std::max(best, hs.score);
            }
        std::sort(h1_.begin(), h1_.end());
        std::vector<HdrMap> v;
        int prev = mapping_from_header(r, v, p.top, best * p.top);
        best = std::max(best, prev);
        score_line(r, best, h1_, v); seq_qual(r);
    }
    std::vector<ChrHit> h1_, h2_;
};

void die(int rc, const char *what) { if (rc) { fprintf(stderr, "[soap4-mi355x] %s failed: %s (%d)\n", what, mp_strerror(rc), rc); exit(1); } }

struct Stats { uint64_t pairs = 0, deep = 0, rescued = 0, unpaired = 0, dp = 0; };

// -b only: the special CIGAR of each listed DP instance.  The hits name their instance (read, strand, window); the same
// kernels run it again with the op strings switched on (mp_sgdp_batch), in bounded chunks.
void resolve_cigars(mp_ctx *ctx, const Params &p, const std::vector<mp_dp_task> &tasks, const std::vector<uint32_t> &lens, std::vector<std::string> &out)
{
    out.assign(tasks.size(), std::string());
    const size_t chunk = 1u << 17;
    std::vector<uint8_t> ops;
    std::vector<mp_dp_result> res;
    for (size_t off = 0; off < tasks.size(); off += chunk) {
        const size_t cnt = std::min(chunk, tasks.size() - off);
        uint32_t stride = 0;
        for (size_t t = off; t < off + cnt; t++) stride = std::max(stride, lens[tasks[t].read_id] + tasks[t].win_len + 2);
        stride = (stride + 15) & ~15u;
        ops.assign(cnt * (size_t)stride, 0);
        res.resize(cnt);
        die(mp_sgdp_batch(ctx, tasks.data() + off, (uint32_t)cnt, p.clip_left, p.clip_right, res.data(), ops.data(), stride, nullptr, nullptr), "mp_sgdp_batch");
        for (size_t t = 0; t < cnt; t++) {
            if (res[t].score <= 0 || res[t].n_ops == 0) { fprintf(stderr, "[soap4-mi355x] internal error: a reported hit did not align again\n"); exit(1); }
            // CigarStringEncoder (DV-DPfunctions.h:377-427): runs collected in traceback order, printed from the read's start
            const uint8_t *o = ops.data() + t * (size_t)stride;
            std::string &cg = out[off + t];
            for (int i = (int)res[t].n_ops - 1; i >= 0;) {
                int j = i;
                while (j >= 0 && o[j] == o[i]) j--;
                cg += std::to_string(i - j); cg += (char)o[i];
                i = j;
            }
        }
    }
}

struct BamState {
    std::unique_ptr<mpbam::Writer> w;
    int stage_rescue = 2, stage_unpaired = 3;                // hspaux->dpStageId (alignment.cpp:93,228,308)
};

mpbam::ReadView view_of(const Rec &r, uint32_t len) { return mpbam::ReadView{r.name.c_str(), r.seq.c_str(), r.qual.c_str(), len}; }

// MD text of the best alignments of one pair (-p only): one small device gather per pair
struct PairText {
    std::vector<uint8_t> buf;
    mpbam::Writer::RefText ref[2];
    void fetch(const mp_index *ix, const Side *const best[2])
    {
        uint64_t pos[2]; uint32_t len[2]; uint32_t stride = 16;
        for (int k = 0; k < 2; k++) {
            const bool ok = best[k] && best[k]->pos != NOT_ALIGNED;
            pos[k] = ok ? best[k]->pos : 0; len[k] = ok ? mpbam::Writer::ref_span(best[k]->cigar) + 8 : 0;
            stride = std::max(stride, len[k]);
        }
        buf.assign(2 * (size_t)stride, 0);
        die(mp_index_fetch_text(ix, pos, len, 2, buf.data(), stride), "mp_index_fetch_text");
        for (int k = 0; k < 2; k++) { ref[k].pos = pos[k]; ref[k].codes = buf.data() + (size_t)k * stride; ref[k].len = len[k]; }
    }
};

void align_batch(mp_index *ix, mp_ctx *ctx, Params &p, Batch &b, bool &first, Writer &w, BamState *bam, Stats &st)
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
    ap.want_tie_counts = bam ? 1 : 0; ap.reserved = 0;          // num_sameScore only feeds the BAM MAPQ / X0 / X1
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
    hits.erase(std::unique(hits.begin(), hits.end(), [](const mp_pair_hit &x, const mp_pair_hit &y) {
        return x.pair_id == y.pair_id && x.algnmt_1 == y.algnmt_1 && x.algnmt_2 == y.algnmt_2 && x.score_1 == y.score_1 && x.score_2 == y.score_2; }), hits.end());
    std::vector<std::string> cig;
    std::vector<mp_dp_task> tasks;
    if (bam) {
        tasks.reserve(2 * hits.size());
        for (const mp_pair_hit &h : hits) {
            const uint32_t s1 = h.left_is_mate ? 2u : 1u;       // the left leg is aligned as is, the right leg reverse-complemented
            tasks.push_back(mp_dp_task{2 * h.pair_id, s1, h.start_1, h.dplen_1, 0});
            tasks.push_back(mp_dp_task{2 * h.pair_id + 1, 3u - s1, h.start_2, h.dplen_2, 0});
        }
        resolve_cigars(ctx, p, tasks, b.lens, cig);
    }
    std::vector<char> done(n_pairs, 0);
    uint32_t n_done = 0;
    std::vector<PairAln> res;
    PairText text;
    auto best_of = [](const std::vector<PairAln> &v) {           // first maximum of score_1 + score_2 (OutputDPResult.cpp:156-232)
        size_t best = 0;
        for (size_t i = 1; i < v.size(); i++) if (v[i].m[0].score + v[i].m[1].score > v[best].m[0].score + v[best].m[1].score) best = i;
        return best;
    };
    auto emit_pair = [&](uint32_t q, int file, int stage) {
        const Rec &r1 = b.recs[2 * q], &r2 = b.recs[2 * q + 1];
        const size_t best = best_of(res);
        w.pair(r1, r2, b.lens[2 * q], b.lens[2 * q + 1], res);
        if (!bam) return;
        const mpbam::ReadView rd[2] = {view_of(r1, b.lens[2 * q]), view_of(r2, b.lens[2 * q + 1])};
        if (bam->w->wants_text()) { const Side *bs[2] = {&res[best].m[0], &res[best].m[1]}; text.fetch(ix, bs); }
        bam->w->pair(file, stage, rd, res, best, text.ref);
    };
    for (size_t i = 0; i < hits.size();) {
        size_t j = i;
        res.clear();
        const uint32_t q = hits[i].pair_id;
        for (; j < hits.size() && hits[j].pair_id == q; j++) {
            const mp_pair_hit &h = hits[j];
            PairAln r;
            r.m[0].pos = h.algnmt_1; r.m[1].pos = h.algnmt_2; r.m[0].score = h.score_1; r.m[1].score = h.score_2;
            r.m[0].strand = h.left_is_mate ? 2 : 1; r.m[1].strand = 3 - r.m[0].strand;
            if (bam) {
                // the reference computes both edit distances with the right leg's read length (DV-DPfunctions.cpp:3466, :2999)
                const int right_len = (int)b.lens[2 * q + (h.left_is_mate ? 0 : 1)], right_side = h.left_is_mate ? 0 : 1;
                int dis[2];
                for (int k = 0; k < 2; k++) {
                    r.m[k].cigar = cig[2 * j + k].c_str();
                    r.m[k].editdist = mpbam::edit_dist(r.m[k].cigar, right_len, r.m[k].score, &dis[k]);
                }
                r.m[0].count = h.count_1; r.m[1].count = h.count_2;
                const uint64_t d = h.algnmt_1 < h.algnmt_2 ? h.algnmt_2 - h.algnmt_1 : h.algnmt_1 - h.algnmt_2;
                r.insert = (long long)d + right_len + dis[right_side];                              // :3520-3525
            }
            res.push_back(r);
        }
        emit_pair(q, mpbam::Writer::FILE_DPOUT, 1);
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
    singles.resize(ns);
    die(mp_get_rescue_hits(ctx, nullptr, 0, &nr), "mp_get_rescue_hits");
    std::vector<mp_rescue_hit> resc(nr + 1);
    die(mp_get_rescue_hits(ctx, resc.data(), nr, &nr), "mp_get_rescue_hits");
    resc.resize(nr);
    // mate rescue results as (read, mate) pairs; OutputBuffer::ready(1): sorted, duplicates and failed rescues dropped
    struct RescueRow { uint32_t pair; uint64_t a1, a2; int s1, s2; uint32_t src; };
    std::vector<RescueRow> rows(nr);
    for (uint64_t i = 0; i < nr; i++) {
        const mp_rescue_hit &h = resc[i];
        if ((h.aligned_read & 1u) == 0) rows[i] = RescueRow{h.pair_id, h.aligned_pos, h.dp_pos, h.aligned_score, h.dp_score, (uint32_t)i};
        else rows[i] = RescueRow{h.pair_id, h.dp_pos, h.aligned_pos, h.dp_score, h.aligned_score, (uint32_t)i};
    }
    std::sort(rows.begin(), rows.end(), [](const RescueRow &x, const RescueRow &y) {
        if (x.pair != y.pair) return x.pair < y.pair;
        if (x.a1 != y.a1) return x.a1 < y.a1;
        if (x.a2 != y.a2) return x.a2 < y.a2;
        if (x.s1 != y.s1) return x.s1 < y.s1;
        return x.s2 < y.s2;
    });
    rows.erase(std::unique(rows.begin(), rows.end(), [](const RescueRow &x, const RescueRow &y) {
        return x.pair == y.pair && x.a1 == y.a1 && x.a2 == y.a2 && x.s1 == y.s1 && x.s2 == y.s2; }), rows.end());
    // single-end hits per read, OutputBuffer::ready(): sorted by (position, score), duplicates dropped (DV-DPfunctions.cpp:853-866)
    std::sort(singles.begin(), singles.end(), [](const mp_single_hit &x, const mp_single_hit &y) {
        if (x.read_id != y.read_id) return x.read_id < y.read_id;
        if (x.algnmt != y.algnmt) return x.algnmt < y.algnmt;
        return x.score < y.score;
    });
    singles.erase(std::unique(singles.begin(), singles.end(), [](const mp_single_hit &x, const mp_single_hit &y) {
        return x.read_id == y.read_id && x.algnmt == y.algnmt && x.score == y.score; }), singles.end());
    for (const RescueRow &r : rows) done[r.pair] = 2;
    std::vector<std::string> rcig, scig;
    if (bam) {
        tasks.clear();
        for (const RescueRow &r : rows) {
            const mp_rescue_hit &h = resc[r.src];
            tasks.push_back(mp_dp_task{h.aligned_read, h.aligned_strand, h.aligned_win_start, h.aligned_win_len, 0});
            tasks.push_back(mp_dp_task{h.aligned_read ^ 1u, h.dp_strand, h.dp_win_start, h.dp_win_len, 0});
        }
        resolve_cigars(ctx, p, tasks, b.lens, rcig);
        tasks.clear();
        for (const mp_single_hit &h : singles) if (done[h.read_id >> 1] == 0) tasks.push_back(mp_dp_task{h.read_id, h.strand, h.win_start, h.win_len, 0});
        resolve_cigars(ctx, p, tasks, b.lens, scig);
    }
    // the default-DP stage writes every rescued pair of the batch before the unpaired output starts (alignment.cpp:225-330)
    for (size_t i = 0; i < rows.size();) {
        size_t j = i;
        res.clear();
        const uint32_t q = rows[i].pair;
        for (; j < rows.size() && rows[j].pair == q; j++) {
            const mp_rescue_hit &h = resc[rows[j].src];
            const int al = (int)(h.aligned_read & 1u), un = 1 - al;
            PairAln r;
            r.m[al].pos = h.aligned_pos; r.m[al].score = h.aligned_score; r.m[al].strand = (int)h.aligned_strand;
            r.m[un].pos = h.dp_pos; r.m[un].score = h.dp_score; r.m[un].strand = (int)h.dp_strand;
            if (bam) {
                int dis = 0;
                r.m[al].cigar = rcig[2 * j].c_str(); r.m[un].cigar = rcig[2 * j + 1].c_str();
                r.m[al].editdist = mpbam::edit_dist(r.m[al].cigar, (int)b.lens[h.aligned_read], h.aligned_score, nullptr);
                r.m[un].editdist = mpbam::edit_dist(r.m[un].cigar, (int)b.lens[h.aligned_read ^ 1u], h.dp_score, &dis);
                r.m[al].count = (int)h.aligned_count; r.m[un].count = (int)h.dp_count;
                // DV-DPfunctions.cpp:1553-1565
                if (h.dp_pos < h.aligned_pos) r.insert = (long long)(h.aligned_pos - h.dp_pos) + b.lens[h.aligned_read];
                else r.insert = (long long)(h.dp_pos - h.aligned_pos) + b.lens[h.aligned_read ^ 1u] + dis;
            }
            res.push_back(r);
        }
        emit_pair(q, mpbam::Writer::FILE_GOUT, bam ? bam->stage_rescue : 0);
        st.rescued++;
        i = j;
    }
    std::vector<Side> lst[2];
    std::vector<mpbam::Writer::RefText> ref_of[2];
    std::vector<uint8_t> tbuf[2];
    size_t si = 0, ci = 0;
    for (uint32_t q = 0; q < n_pairs; q++) {
        if (done[q]) { while (si < singles.size() && (singles[si].read_id >> 1) == q) si++; continue; }
        for (int k = 0; k < 2; k++) {
            lst[k].clear();
            const uint32_t r = 2 * q + (uint32_t)k;
            for (; si < singles.size() && singles[si].read_id == r; si++) {
                const mp_single_hit &h = singles[si];
                Side s;
                s.pos = h.algnmt; s.score = h.score; s.strand = (int)h.strand; s.count = (int)h.count;
                if (bam) { s.cigar = scig[ci++].c_str(); s.editdist = mpbam::edit_dist(s.cigar, (int)b.lens[r], h.score, nullptr); }
                lst[k].push_back(s);
            }
            w.single(b.recs[r], b.lens[r], lst[k]);
        }
        if (bam) {
            const mpbam::ReadView rd[2] = {view_of(b.recs[2 * q], b.lens[2 * q]), view_of(b.recs[2 * q + 1], b.lens[2 * q + 1])};
            const mpbam::Writer::RefText *rp[2] = {nullptr, nullptr};
            if (bam->w->wants_text())
                for (int k = 0; k < 2; k++) {
                    const size_t n = lst[k].size();
                    ref_of[k].assign(n, mpbam::Writer::RefText());
                    if (!n) continue;
                    std::vector<uint64_t> pos(n); std::vector<uint32_t> len(n); uint32_t stride = 16;
                    for (size_t i = 0; i < n; i++) {
                        const bool ok = lst[k][i].pos != NOT_ALIGNED;
                        pos[i] = ok ? lst[k][i].pos : 0; len[i] = ok ? mpbam::Writer::ref_span(lst[k][i].cigar) + 8 : 0; stride = std::max(stride, len[i]);
                    }
                    tbuf[k].assign(n * (size_t)stride, 0);
                    die(mp_index_fetch_text(ix, pos.data(), len.data(), (uint32_t)n, tbuf[k].data(), stride), "mp_index_fetch_text");
                    for (size_t i = 0; i < n; i++) { ref_of[k][i].pos = pos[i]; ref_of[k][i].codes = tbuf[k].data() + i * (size_t)stride; ref_of[k][i].len = len[i]; }
                    rp[k] = ref_of[k].data();
                }
            bam->w->unpaired(bam->stage_unpaired, rd, lst, rp);
        }
        st.unpaired++;
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
    int cli_device = -1, threads = 4, out_type = 2;                       // IniParam.cpp:627-648
    const char *prefix = argv[2], *fq1 = argv[3], *fq2 = argv[4];
    std::string out_prefix = fq1;                                         // IniParam.cpp:633
    mpbam::Options bopt;
    bool have_d = false, have_a = false, have_r = false;
    // updateUserInputText (IniParam.cpp:199-226): "\t" and "\n" typed on the command line become the characters
    auto unescape = [](const char *v) { std::string o; for (const char *c = v; *c; c++) { if (c[0] == '\\' && (c[1] == 't' || c[1] == 'n')) { o += c[1] == 't' ? '\t' : '\n'; c++; } else o += *c; } return o; };
    for (int i = 5; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
        // arities follow parseInputArgs (IniParam.cpp:651-896)
        if (a == "-e") (void)val();
        else if (a == "-o") out_prefix = val();
        else if (a == "-T") { threads = atoi(val()); if (threads <= 0) { fprintf(stderr, "Please specify a positive number of CPU threads after '-T'\n"); return 1; } }
        else if (a == "-D") { bopt.read_group = val(); have_d = true; }
        else if (a == "-A") { bopt.sample_name = unescape(val()); have_a = true; }
        else if (a == "-R") { bopt.rg_option = unescape(val()); have_r = true; }
        else if (a == "-h") { out_type = atoi(val()); if (out_type < 1 || out_type > 4) { fprintf(stderr, "The output option should be 1, 2, 3 or 4\n"); return 1; } }
        else if (a == "-s") { if (i + 1 < argc && isdigit((unsigned char)argv[i + 1][0])) i++; }   // IniParam.cpp:738-757
        else if (a == "-c") { cli_device = atoi(val()); if (cli_device < 0) { fprintf(stderr, "The GPU device ID should not be less than 0\n"); return 1; } }
        else if (a == "-p") bopt.print_mdnm = true;                                                   // NM / MD tags
        else if (a == "-I") {}                                                                       // flag without a value
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
    // IniParam.cpp:918-940: -R needs -D and -A; -A and -D come together
    if ((have_r && (!have_a || !have_d)) || (have_a != have_d)) { fprintf(stderr, "Options '-D' and '-A' have to be specified together ('-R' needs both).\n"); return 1; }
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
    if (p.want_bam) {
        if (out_type != 2) { fprintf(stderr, "[soap4-mi355x] -b is implemented for the default output option -h 2 (all best) only\n"); return 1; }
        if (ini_int(ini, "Score:BWALikeScore", 0) != 1) { fprintf(stderr, "[soap4-mi355x] -b needs BWALikeScore=1 (as in the shipped ini files)\n"); return 1; }
        bopt.max_mapq = ini_int(ini, "Score:MaxMAPQ", 40); bopt.min_mapq = ini_int(ini, "Score:MinMAPQ", 1);
        if (bopt.read_group.empty()) bopt.read_group = fq1;               // SOAP4.cpp:297-300
        bopt.threads = threads;
        if (const char *e = getenv("MP_BGZF_LEVEL")) bopt.level = atoi(e);
    }
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
    BamState bam_state, *bam = nullptr;
    if (p.want_bam) {
        bam_state.w.reset(new mpbam::Writer(ix, bopt));
        if (!bam_state.w->open(out_prefix, "@PG\tID:soap4-mi355x\tPN:soap4-mi355x\tVN:r1\n")) return 1;
        bam_state.stage_unpaired = p.skip_default_dp ? 2 : 3;
        bam = &bam_state;
    }
    for (int k = 0;; k ^= 1) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[k] == 1 || reader_done; }); if (ready[k] != 1) break; }
        align_batch(ix, ctx, p, bufs[k], first, w, bam, st);
        bool last = bufs[k].last;
        { std::lock_guard<std::mutex> lk(mu); ready[k] = 0; }
        cv.notify_all();
        if (last) break;
    }
    reader.join();
    w.flush();
    fflush(stdout);
    if (bam && !bam->w->close()) { fprintf(stderr, "[soap4-mi355x] error while writing the BAM files\n"); return 1; }
    fprintf(stderr, "[Main] %llu pairs: %llu aligned by deep DP, %llu by mate rescue, %llu unpaired; %llu DP instances\n", (unsigned long long)st.pairs,
            (unsigned long long)st.deep, (unsigned long long)st.rescued, (unsigned long long)st.unpaired, (unsigned long long)st.dp);
    mp_ctx_destroy(ctx);
    mp_index_close(ix);
    return 0;
}
