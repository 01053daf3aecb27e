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
#include <immintrin.h>
#include <stdarg.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <ctype.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>
#include <memory>
#include <functional>
#include "../../../include/megapath_hip.h"
#include "bam_out.h"
#include "fast_inflate.h"
#include "par_gzip.h"
#include <deque>

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
    int lsam = 0;                 // extension: 1 = print LSAM lines instead of annotated FASTQ (what `| fastq2lsam 1` yields), 2 = without sequences
    const struct Taxonomy *tax = nullptr;   // extension -lsamid: the lines are LSAM.id lines, what `| taxLookupAcc acc2tid nodes names -` makes of them
    double pass1_t = 40;          // extension -pass1: reassign's -t (cc/reassign.cpp:19)
    mp_mmp_params mmp = {30, 17, 6, 5, 27, 18, 0.85, 4, 0.5};       // defaults of ParseIniFile (IniParam.cpp:427-435)
    size_t batch_pairs = 1u << 20;                                   // SOAP4.cpp:206
};

// One read as the stages see it.  Name, comment, sequence and qualities are VIEWS into the batch's text -- the mapped input file,
// or the batch's own buffer of inflated bytes -- nothing is copied; only what cannot be used as it stands is owned (RecOwned).
struct RecOwned {
    std::string seq, qual;      // the decoded 2-bit read when the text has anything but upper-case ACGT (the reference echoes the decoded
                                // read: N -> G), and the joined lines of a wrapped record
    std::string comment;        // resident chunks: the comment the chunk before left in place of the record's own (Writer::carry)
};
struct Rec {
    const char *name = nullptr, *seq = nullptr, *qual = nullptr, *comment = nullptr;
    uint32_t name_len = 0, seq_len = 0, qual_len = 0, comment_len = 0;   // seq / qual after the -L cut (QueryParser.cpp:188)
    bool has_comment = false;   // the rest of the header line (kseq), unless -nc
    std::unique_ptr<RecOwned> own;
    std::string_view name_v() const { return std::string_view(name, name_len); }
    std::string_view seq_v() const { return std::string_view(seq, seq_len); }
    std::string_view qual_v() const { return std::string_view(qual, qual_len); }
    std::string_view comment_v() const { return std::string_view(comment, comment_len); }
    bool ignore() const { return has_comment && comment_len == 6 && memcmp(comment, "IGNORE", 6) == 0; }
    RecOwned &owned() { if (!own) own.reset(new RecOwned()); return *own; }
};

struct RescueRow { uint32_t pair; uint64_t a1, a2; int s1, s2; uint32_t src; };

// One batch on its way through the three stages: reader threads fill the inputs, the GPU stage the hit lists, the output
// stage prints.  The objects are recycled, so the vectors keep their capacity.
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// where a record's lines sit in its block of text (offsets from the block's base); sn/qn of a wrapped record span several lines
struct RawRec { uint64_t h; uint32_t hn; uint64_t s; uint32_t sn; uint64_t q; uint32_t qn; bool fasta, wrapped; };

// a batch's inflated text: grows by realloc (large blocks are remapped, not copied) and is never zero-filled -- a std::vector would
// write every byte once before the inflated bytes are written over it
struct TextBuf {
    char *p = nullptr;
    size_t cap = 0;
    TextBuf() = default;
    TextBuf(const TextBuf &) = delete;
    TextBuf &operator=(const TextBuf &) = delete;
    ~TextBuf() { free(p); }
    char *data() { return p; }
    const char *data() const { return p; }
    size_t size() const { return cap; }
    void resize(size_t n)
    {
        if (n <= cap) return;
        char *q = (char *)realloc(p, n);
        if (!q) { fprintf(stderr, "[soap4-mi355x] out of memory (%zu bytes of read text)\n", n); exit(1); }
        p = q; cap = n;
    }
};

struct Batch {
    // ---- stage 1: parsed and packed reads (2 per pair)
    std::vector<Rec> recs[2];               // per mate file, grown on demand
    std::vector<RawRec> raw[2];
    TextBuf text[2];                        // gz input: the inflated bytes of this batch's records (plain files are mapped instead)
    std::vector<uint32_t> nl[2];            // scratch of the splitter: where the lines of the block end
    std::vector<uint32_t> packed, lens;     // interleaved 2q / 2q+1 as the device wants them
    uint32_t W = 0, n_reads = 0;
    bool last = false;
    const Rec &rec(uint32_t r) const { return recs[r & 1u][r >> 1]; }
    Rec &rec_mut(uint32_t r) { return recs[r & 1u][r >> 1]; }
    // ---- stage 2: what the device found, per index chunk (one entry unless several chunks are resident, see main)
    struct ChunkHits {
        std::vector<mp_pair_hit> hits;          // deep-DP round
        std::vector<mp_single_hit> singles;     // rescue round: single-end hits of the failed pairs
        std::vector<mp_rescue_hit> resc;        //               mate rescues
        std::vector<RescueRow> rows;
        std::vector<char> done;                 // per pair: 1 deep, 2 rescued
        std::vector<std::string> cig, rcig, scig;   // -b: special CIGARs of hits / rows / singles, in list order
        bool lists_ready = false;               // hit lists already sorted and de-duplicated (done early when -b needs CIGARs)
        uint64_t dp_instances = 0;
        void clear() { hits.clear(); singles.clear(); resc.clear(); rows.clear(); cig.clear(); rcig.clear(); scig.clear(); lists_ready = false; dp_instances = 0; }
    };
    std::vector<ChunkHits> ch;
    // ---- stage 2, reduced on the device (no -b): one sorted list of (read, chunk, sequence, best score) keys for the whole batch and all
    // chunks (mp_reduce_finish), and where each read's keys start
    bool reduced = false;
    std::vector<uint64_t> red;
    std::vector<uint32_t> red_off;
    std::vector<uint8_t> red_class;         // per pair: how the LAST chunk aligned it (0 deep-DP round, 1 mate rescue, 2 neither): the sections of the output
    mp_reduce_stats red_last{};             // how the last chunk's pairs were aligned (the closing line counts that chunk, as the last run of a chain would)
    uint64_t red_dp_instances = 0;
    uint64_t red_floor_pairs = 0;           // (pair, chunk) combinations whose rescue work the -top rule left out
    uint64_t red_floor_mate_pairs = 0;      // ... of them by the bound on the rescued mate's windows
    uint64_t seq = 0;                       // position of the batch in the input: the printer writes batches in this order
};

unsigned char g_code[256], g_unclean[256];
void fill_char_map()
{
    memset(g_code, 0, sizeof g_code);                         // INDEXFillCharMap: everything is 'A' unless listed
    const char *b = "ACGT";
    for (int i = 0; i < 4; i++) { g_code[(int)b[i]] = (unsigned char)i; g_code[(int)b[i] + 32] = (unsigned char)i; }
    g_code['U'] = g_code['u'] = 3;
    g_code['N'] = g_code['n'] = 2;
    memset(g_unclean, 1, sizeof g_unclean);                   // the text can be echoed as it stands only where it is upper-case ACGT
    for (int i = 0; i < 4; i++) g_unclean[(int)b[i]] = 0;
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

// Errors end the process from whichever thread meets them.  exit() would run the atexit handlers and flush stdio while the
// other stages are still writing; _exit() after the message does not.
[[noreturn]] void fatal(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fflush(stderr);
    _exit(1);
}

// A few threads that live as long as their owner (one pool per read file for the splitter, one for the printer): run(n, body) calls
// body(0 .. n-1), the caller being one of the runners, and returns when all are done.  Thread creation per batch slice was a
// measurable part of the host pipeline.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> *job = nullptr;
    int n_jobs = 0, next = 0, running = 0;
    uint64_t gen = 0;
    bool stop = false;
    explicit Pool(int helpers)
    {
        for (int i = 0; i < helpers; i++)
            th.emplace_back([this]() {
                uint64_t seen = 0;
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    cv.wait(lk, [&] { return stop || (gen != seen && next < n_jobs); });
                    if (stop) return;
                    const uint64_t g = gen;
                    while (gen == g && next < n_jobs) {
                        const int t = next++;
                        running++;
                        const std::function<void(int)> *fn = job;
                        lk.unlock();
                        (*fn)(t);
                        lk.lock();
                        running--;
                    }
                    seen = g;
                    if (running == 0 && next >= n_jobs) cv_done.notify_all();
                }
            });
    }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
    void run(int n, const std::function<void(int)> &body)
    {
        if (n <= 0) return;
        if (n == 1 || th.empty()) { for (int t = 0; t < n; t++) body(t); return; }
        std::unique_lock<std::mutex> lk(mu);
        job = &body; n_jobs = n; next = 0; gen++;
        cv.notify_all();
        while (next < n_jobs) {
            const int t = next++;
            running++;
            lk.unlock();
            body(t);
            lk.lock();
            running--;
        }
        cv_done.wait(lk, [&] { return running == 0; });
        job = nullptr; n_jobs = 0;
    }
};

// runs body(t) for t = 0 .. n-1 on n threads (the caller is one of them): on the pool when there is one
template <typename F>
void fan_out(int n, F &&body, Pool *pool = nullptr)
{
    if (pool) { const std::function<void(int)> fn = [&body](int t) { body(t); }; pool->run(n, fn); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < n; t++) th.emplace_back([&body, t]() { body(t); });
    body(0);
    for (auto &x : th) x.join();
}

// ------------------------------------------------------------------------------------------------ FASTQ / FASTA input (plain or gz)
// The reference reads its inputs through kseq on ONE IO thread (aio_thread.cpp:247,665; QueryParser.cpp:184-262).  Here a read file is
// either MAPPED (plain text: no copy at all, the records of a batch are views into the mapping) or inflated by zlib into the
// batch's own buffer (one gzip stream inflates at a few hundred MB/s: that, not the aligner, is what a .gz input runs at).
// A batch's block of text is cut into lines by several threads, each over its own byte range (the line number of a byte follows
// from the newline counts of the ranges before it), then the same threads build the records and pack the reads.  That fast path
// takes the block to be strictly four (FASTQ) or two (FASTA) lines per record and CHECKS it record by record; anything else
// (blank lines, wrapped sequences, mixed files) goes through the careful sequential splitter below, which reads records the way
// kseq does: sequence lines up to the '+' line (or the next header), then quality lines until they are as long as the sequence.
// A gzip'ed read file is inflated by its own thread, ahead of the splitter (fast_inflate.h: two to three times zlib's rate), into
// 8 MiB chunks that each begin with the last 32 KiB of the chunk before (the history a match may reach into).  The consumer copies
// the bytes into the batch's buffer, computing each member's CRC-32 on the way and comparing it with the trailer's.
struct GzProducer {
    struct MemberEnd { size_t pos; uint32_t crc, isize; };
    struct Chunk { std::vector<uint8_t> buf; size_t begin = 0, end = 0; std::vector<MemberEnd> ends; bool eof = false, verified = false; std::string err;
                   mpz::ParGzip::Piece *piece = nullptr; };
    static constexpr size_t H = 32768, C = (size_t)8 << 20;
    mpz::Inflater z;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Chunk *> ready, free_;
    std::vector<std::unique_ptr<Chunk>> all;
    bool stop = false;
    // BGZF (htslib's blocked gzip -- bgzip, and what BBTools writes when bgzip is installed): every member carries its own compressed size in
    // a 'BC' extra field and refers to nothing outside itself, so groups of members are inflated side by side and handed over in file order
    struct Group { size_t a, b; };                  // compressed bytes [a, b) of consecutive members, <= C bytes inflated
    std::vector<Group> groups;
    size_t next_group = 0, deliver = 0;
    std::map<size_t, Chunk *> done;
    const uint8_t *data_ = nullptr;
    // an ordinary gzip file of some size: several threads on the one stream (par_gzip.h)
    std::unique_ptr<mpz::ParGzip> par;

    // the members of a BGZF file as groups; false when the file is not BGZF from its first to its last byte
    static bool scan_bgzf(const uint8_t *d, size_t len, std::vector<Group> &out)
    {
        size_t off = 0, ga = 0, inflated = 0;
        while (off < len) {
            if (len - off < 26 || d[off] != 0x1f || d[off + 1] != 0x8b || d[off + 2] != 8 || !(d[off + 3] & 4)) return false;
            const size_t xlen = d[off + 10] | (size_t)d[off + 11] << 8;
            if (off + 12 + xlen > len) return false;
            size_t bsize = 0;
            for (size_t x = off + 12; x + 4 <= off + 12 + xlen;) {
                const size_t sl = d[x + 2] | (size_t)d[x + 3] << 8;
                if (d[x] == 'B' && d[x + 1] == 'C' && sl == 2 && x + 6 <= off + 12 + xlen) bsize = (d[x + 4] | (size_t)d[x + 5] << 8) + 1;
                x += 4 + sl;
            }
            if (bsize < 26 || off + bsize > len) return false;
            const uint8_t *t = d + off + bsize - 4;
            const size_t isize = t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24;
            if (isize > 65536) return false;
            if (inflated + isize > C && off > ga) { out.push_back(Group{ga, off}); ga = off; inflated = 0; }
            inflated += isize;
            off += bsize;
        }
        if (off > ga) out.push_back(Group{ga, off});
        return !out.empty();
    }
    void start(const uint8_t *data, size_t len, int n_chunks, int bgzf_threads)
    {
        data_ = data;
        for (int i = 0; i < n_chunks; i++) { all.emplace_back(new Chunk()); free_.push_back(all.back().get()); }      // (a chunk's buffer is allocated when it is first used)
        if (bgzf_threads > 0 && scan_bgzf(data, len, groups)) {
            for (int w = 0; w < bgzf_threads; w++)
                th.emplace_back([this]() {
                    mpz::Inflater zz;
                    for (;;) {
                        Chunk *c = nullptr;
                        size_t g = 0;
                        {   // a group and the chunk for it are taken together, in file order: the group to deliver next always has its chunk
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return stop || next_group >= groups.size() || !free_.empty(); });
                            if (stop || next_group >= groups.size()) return;
                            g = next_group++; c = free_.front(); free_.pop_front();
                        }
                        c->ends.clear(); c->err.clear(); c->eof = g + 1 == groups.size(); c->verified = true;
                        if (c->buf.empty()) c->buf.resize(H + C + 512);
                        zz.init(data_ + groups[g].a, groups[g].b - groups[g].a);
                        size_t pos = 0, mstart = 0;
                        for (;;) {
                            const int rc = zz.run(c->buf.data(), pos, H + C);
                            if (rc == mpz::MEMBER_END) {
                                if (mpz::fast_crc32(0u, c->buf.data() + mstart, pos - mstart) != zz.member_crc() || (uint32_t)(pos - mstart) != zz.member_isize())
                                    { c->err = "incorrect data check (gzip CRC-32 or length)"; break; }
                                mstart = pos;
                                continue;
                            }
                            if (rc == mpz::ERR) c->err = zz.error();
                            else if (rc == mpz::NEED_OUTPUT) c->err = "BGZF member larger than its header says";
                            break;
                        }
                        c->begin = 0; c->end = pos;
                        { std::lock_guard<std::mutex> lk(mu); done[g] = c; }
                        cv.notify_all();
                    }
                });
            return;
        }
        groups.clear();
        {
            // one deflate stream, several threads: worth it from a few segments on (MP_PARGZ_MIN bytes of compressed input, segments of MP_PARGZ_SEG)
            const size_t min_len = getenv("MP_PARGZ_MIN") ? (size_t)atoll(getenv("MP_PARGZ_MIN")) : ((size_t)24 << 20);
            const size_t seg = getenv("MP_PARGZ_SEG") ? (size_t)atoll(getenv("MP_PARGZ_SEG")) : ((size_t)4 << 20);
            // (pieces of ~4 x the segment: as far ahead of the splitter as the one-thread reader runs, about one batch)
            if (bgzf_threads >= 2 && len >= min_len) { par.reset(new mpz::ParGzip(data, len, bgzf_threads, seg, bgzf_threads + 2, (int)std::max<size_t>(6, ((size_t)96 << 20) / seg))); return; }
        }
        z.init(data, len);
        th.emplace_back([this]() {
            std::vector<uint8_t> hist(H);
            size_t hist_n = 0;
            for (;;) {
                Chunk *c = nullptr;
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !free_.empty(); }); if (stop) return; c = free_.front(); free_.pop_front(); }
                c->ends.clear(); c->err.clear(); c->eof = false; c->verified = false;
                if (c->buf.empty()) c->buf.resize(H + C + 512);
                memcpy(c->buf.data(), hist.data(), hist_n);
                size_t pos = hist_n;
                c->begin = pos;
                int rc;
                for (;;) {
                    rc = z.run(c->buf.data(), pos, H + C);
                    if (rc != mpz::MEMBER_END) break;
                    c->ends.push_back(MemberEnd{pos, z.member_crc(), z.member_isize()});
                }
                c->end = pos;
                if (rc == mpz::ERR) c->err = z.error();
                if (rc == mpz::STREAM_END) c->eof = true;
                hist_n = std::min(pos, H);
                memcpy(hist.data(), c->buf.data() + pos - hist_n, hist_n);
                { std::lock_guard<std::mutex> lk(mu); ready.push_back(c); }
                cv.notify_all();
                if (rc != mpz::NEED_OUTPUT) return;
            }
        });
    }
    Chunk *pop()
    {
        if (par) {
            mpz::ParGzip::Piece *p = par->pop();
            Chunk *c = nullptr;
            { std::lock_guard<std::mutex> lk(mu); c = free_.front(); free_.pop_front(); }     // (the reader holds one chunk at a time: there is always a free one)
            c->buf.swap(p->bytes);
            c->begin = 0; c->end = p->n; c->ends.clear(); c->eof = p->eof; c->verified = true; c->err = p->err; c->piece = p;
            return c;
        }
        std::unique_lock<std::mutex> lk(mu);
        if (!groups.empty()) {
            cv.wait(lk, [&] { return done.count(deliver) != 0; });
            Chunk *c = done[deliver];
            done.erase(deliver++);
            return c;
        }
        cv.wait(lk, [&] { return !ready.empty(); });
        Chunk *c = ready.front(); ready.pop_front();
        return c;
    }
    void give_back(Chunk *c)
    {
        if (c->piece) { c->buf.swap(c->piece->bytes); par->recycle(c->piece); c->piece = nullptr; }
        { std::lock_guard<std::mutex> lk(mu); free_.push_back(c); }
        cv.notify_all();
    }
    ~GzProducer() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) if (t.joinable()) t.join(); }
};

struct Source {
    std::string path;
    bool gz = false;
    const char *map = nullptr;                // plain file
    size_t map_len = 0, pos = 0;
    gzFile f = nullptr;                       // gz file read through zlib (pipes, devices, MP_INFLATE=zlib)
    std::unique_ptr<GzProducer> fast;         // gz file inflated by its own thread (regular files: the compressed bytes are mapped)
    GzProducer::Chunk *cur = nullptr;
    size_t cur_pos = 0, cur_end_idx = 0;
    uint32_t m_crc = 0, m_size = 0;           // CRC-32 and length of the gzip member being consumed
    std::vector<char> carry;                  // inflated bytes behind the last batch's records
    bool eof = false;
    bool careful = false;                     // once the strict shape failed the file stays on the careful splitter
    double bytes_per_rec = 0;                 // estimate for the next window
    std::unique_ptr<Pool> pool;               // the threads that cut this file's blocks into lines and pack its reads
    std::thread toucher;                      // plain file: maps the pages of the NEXT block while this one is worked on
    void touch_ahead(size_t from, size_t bytes)
    {
        if (toucher.joinable()) toucher.join();
        static const bool on = getenv("MP_TOUCH_AHEAD") && atoi(getenv("MP_TOUCH_AHEAD")) != 0;      // off by default: measured slower (it competes with the splitters' own faults)
        if (!on || !map || from >= map_len) return;
        const size_t a = from & ~(size_t)4095, n = std::min(bytes + 4096, map_len - a);
        const char *base = map;
        toucher = std::thread([base, a, n]() {
#ifdef MADV_POPULATE_READ
            if (madvise((void *)(base + a), n, MADV_POPULATE_READ) == 0) return;
#endif
            volatile char sink = 0;
            for (size_t i = 0; i < n; i += 4096) sink = (char)(sink + base[a + i]);
            (void)sink;
        });
    }

    bool open(const char *file, int host_threads = 4)
    {
        path = file;
        const int fd = ::open(file, O_RDONLY);
        if (fd < 0) return false;
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return false; }
        gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        const char *inf = getenv("MP_INFLATE");
        if (gz && S_ISREG(st.st_mode) && st.st_size > 0 && !(inf && !strcmp(inf, "zlib"))) {
            map_len = (size_t)st.st_size;
            void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            close(fd);
            if (m == MAP_FAILED) return false;
            (void)madvise(m, map_len, MADV_SEQUENTIAL);
            map = (const char *)m;
            m_crc = (uint32_t)crc32(0L, Z_NULL, 0);
            fast.reset(new GzProducer());
            // BGZF: members inflate side by side; gzip: segments of the one stream (par_gzip.h).  Three eighths of -T per file, as for the splitters
            // (inflating and splitting alternate), at least 3 and at most 8
            const int bg = getenv("MP_INFLATE_THREADS") ? atoi(getenv("MP_INFLATE_THREADS")) : std::min(8, std::max(3, host_threads * 3 / 8));
            fast->start((const uint8_t *)map, map_len, 48, bg);             // up to 384 MiB inflated ahead of the splitter: about one batch
            return true;
        }
        if (gz || !S_ISREG(st.st_mode)) {
            close(fd);
            gz = true;                                            // (a pipe or device is read through zlib too: gzread passes plain bytes through)
            f = gzopen(file, "rb");
            if (f) gzbuffer(f, 1 << 20);
            return f != nullptr;
        }
        map_len = (size_t)st.st_size;
        if (map_len) {
            void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); return false; }
            (void)madvise(m, map_len, MADV_SEQUENTIAL);
            (void)madvise(m, map_len, MADV_WILLNEED);
            map = (const char *)m;
        }
        close(fd);
        return true;
    }
    ~Source() { if (toucher.joinable()) toucher.join(); fast.reset(); if (f) gzclose(f); if (map) munmap((void *)map, map_len); }

    // gz: appends up to `want` more inflated bytes to buf; false at the end of the stream
    bool inflate_more(TextBuf &buf, size_t &end, size_t want)
    {
        if (eof) return false;
        if (end + want > buf.size()) buf.resize(std::max(buf.size() * 2, end + want));
        if (fast) {
            bool got = false;
            while (want) {
                if (!cur) {
                    cur = fast->pop(); cur_pos = cur->begin; cur_end_idx = 0;
                    // a chunk whose members were checked by the thread that inflated it comes with its verdict: no byte of a chunk known to
                    // be corrupt is handed to the splitter (a batch cut from it could be aligned and printed before the end of the chunk is reached)
                    if (cur->verified && !cur->err.empty()) fatal("[soap4-mi355x] error while reading %s: %s\n", path.c_str(), cur->err.c_str());
                }
                const size_t lim = cur_end_idx < cur->ends.size() ? cur->ends[cur_end_idx].pos : cur->end;
                const size_t n = std::min(want, lim - cur_pos);
                if (n) {
                    memcpy(buf.data() + end, cur->buf.data() + cur_pos, n);
                    if (!cur->verified) m_crc = mpz::fast_crc32(m_crc, cur->buf.data() + cur_pos, n);      // (BGZF members are checked by the thread that inflates them)
                    m_size += (uint32_t)n; end += n; cur_pos += n; want -= n; got = true;
                }
                if (cur_pos < lim) continue;
                if (cur_end_idx < cur->ends.size()) {
                    // a gzip member ends here: what zlib checks at this point (a corrupt file must not pass for good data)
                    const GzProducer::MemberEnd &me = cur->ends[cur_end_idx++];
                    if (me.crc != m_crc || me.isize != m_size) fatal("[soap4-mi355x] error while reading %s: incorrect data check (gzip CRC-32 or length)\n", path.c_str());
                    m_crc = (uint32_t)crc32(0L, Z_NULL, 0); m_size = 0;
                    continue;
                }
                if (!cur->err.empty()) fatal("[soap4-mi355x] error while reading %s: %s\n", path.c_str(), cur->err.c_str());
                const bool last = cur->eof;
                fast->give_back(cur);
                cur = nullptr;
                if (last) { eof = true; break; }
            }
            return got;
        }
        const int got = gzread(f, buf.data() + end, (unsigned)std::min(want, (size_t)1 << 28));
        if (got < 0) {
            // a corrupt or truncated .gz must not pass for a shorter input
            int errnum = 0;
            const char *msg = gzerror(f, &errnum);
            fatal("[soap4-mi355x] error while reading %s: %s\n", path.c_str(), errnum == Z_ERRNO ? strerror(errno) : (msg ? msg : "zlib error"));
        }
        if (got == 0) {
            int errnum = 0;
            (void)gzerror(f, &errnum);
            if (errnum != Z_OK && errnum != Z_STREAM_END) fatal("[soap4-mi355x] error while reading %s: %s\n", path.c_str(), gzerror(f, &errnum));
            eof = true;
            return false;
        }
        end += (size_t)got;
        return true;
    }
};

// the careful splitter: records as kseq reads them, one after the other.  `avail(need)` makes sure text[0 .. need) is there if the
// input has that much (gz: inflates more) and returns the bytes now available; returns the bytes the records used.
template <typename Avail, typename Base>
size_t split_careful(Base &&base, Avail &&avail, size_t max_recs, std::vector<RawRec> &raw, const std::string &path)
{
    size_t pos = 0, have = avail(1);
    auto line = [&](size_t &off, size_t &n) -> bool {           // next line without its terminator
        for (;;) {
            const char *b = base();
            const char *nlp = have > pos ? (const char *)memchr(b + pos, '\n', have - pos) : nullptr;
            if (nlp) { off = pos; n = (size_t)(nlp - b) - pos; pos = (size_t)(nlp - b) + 1; break; }
            const size_t more = avail(have + (1u << 22));
            if (more == have) { if (pos >= have) return false; off = pos; n = have - pos; pos = have; break; }
            have = more;
        }
        const char *b = base();
        while (n && (b[off + n - 1] == '\r' || b[off + n - 1] == '\n')) n--;
        return true;
    };
    size_t pend_off = 0, pend_n = 0;
    bool pending = false;                                       // a header line already read while looking for the end of a FASTA record
    while (raw.size() < max_recs) {
        RawRec r{};
        size_t o, n;
        if (pending) { o = pend_off; n = pend_n; pending = false; }
        else do { if (!line(o, n)) return pos; } while (n == 0);
        const char c0 = base()[o];
        if (c0 != '@' && c0 != '>') fatal("[soap4-mi355x] %s: a record does not begin with '@' or '>' (\"%.20s\")\n", path.c_str(), std::string(base() + o, std::min(n, (size_t)20)).c_str());
        r.h = o; r.hn = (uint32_t)n; r.fasta = c0 == '>';
        // sequence lines up to '+', the next header or the end (kseq: ks_getuntil ... until c is '>', '+' or '@')
        size_t s0 = 0, s1 = 0, lines = 0;
        bool plus = false;
        for (;;) {
            const size_t before = pos;
            if (!line(o, n)) break;
            if (n == 0) continue;
            const char c = base()[o];
            if (c == '+') { plus = true; break; }
            if (c == '>' || c == '@') { pend_off = o; pend_n = n; pending = true; (void)before; break; }
            if (lines == 0) s0 = o;
            s1 = o + n; lines++;
        }
        if (lines == 0 && !plus && !pending && r.fasta == false) fatal("[soap4-mi355x] %s ends inside a record (header without sequence)\n", path.c_str());
        r.s = s0; r.sn = (uint32_t)(s1 - s0); r.wrapped = lines > 1;
        if (plus) {
            // quality lines until they hold as many characters as the sequence (kseq)
            size_t seq_chars = 0;
            if (r.wrapped) { const char *b = base(); for (size_t i = s0; i < s1; i++) seq_chars += b[i] != '\n' && b[i] != '\r'; } else seq_chars = r.sn;
            size_t q0 = 0, q1 = 0, ql = 0, got = 0;
            while (got < seq_chars) {
                if (!line(o, n)) break;
                if (ql == 0) q0 = o;
                q1 = o + n; ql++; got += n;
            }
            if (got < seq_chars && ql == 0) fatal("[soap4-mi355x] %s ends inside a FASTQ record\n", path.c_str());
            r.q = q0; r.qn = (uint32_t)(q1 - q0);
            if (ql > 1) r.wrapped = true;
            r.fasta = false;
        } else { r.q = 0; r.qn = 0; r.fasta = true; }
        raw.push_back(r);
    }
    return pending ? pend_off : pos;
}

// offsets (from `text`) of the newlines in [p, e), appended to v: 32 bytes per step where the CPU has AVX2 (the lines of a read file
// are ~80 bytes: a memchr call per line spends more on its start-up than on the bytes)
__attribute__((target("avx2"))) void newlines_avx2(const char *text, const char *p, const char *e, std::vector<uint32_t> &v)
{
    const __m256i nlv = _mm256_set1_epi8('\n');
    while (p + 32 <= e) {
        unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)p), nlv));
        while (m) { v.push_back((uint32_t)(p - text) + (uint32_t)__builtin_ctz(m)); m &= m - 1; }
        p += 32;
    }
    for (; p < e; p++) if (*p == '\n') v.push_back((uint32_t)(p - text));
}
void newlines_memchr(const char *text, const char *p, const char *e, std::vector<uint32_t> &v)
{
    while (p < e) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        v.push_back((uint32_t)(q - text));
        p = q + 1;
    }
}

// Fast path.  nl[] receives the offsets of the first `want_lines` newlines of text[0 .. len) (a missing newline at the very end of
// the input counts as one when at_eof), found by `threads` threads over byte ranges; returns how many were found (<= want_lines).
size_t find_newlines(const char *text, size_t len, bool at_eof, size_t want_lines, std::vector<uint32_t> &nl, int threads, Pool *pool)
{
    const int T = std::max(1, std::min(threads, (int)(len >> 22) + 1));
    std::vector<std::vector<uint32_t>> part((size_t)T);
    fan_out(T, [&](int t) {
        const size_t a = len * (size_t)t / (size_t)T, b = len * (size_t)(t + 1) / (size_t)T;
        std::vector<uint32_t> &v = part[(size_t)t];
        v.reserve((b - a) / 64 + 16);
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) newlines_avx2(text, text + a, text + b, v); else newlines_memchr(text, text + a, text + b, v);
    }, pool);
    size_t total = 0;
    for (auto &v : part) total += v.size();
    const bool virt = at_eof && len > 0 && text[len - 1] != '\n';
    const size_t n = std::min(want_lines, total + (virt ? 1 : 0));
    nl.resize(n);
    size_t o = 0;
    for (auto &v : part) { const size_t c = std::min(v.size(), n - o); if (c) memcpy(nl.data() + o, v.data(), c * 4); o += c; if (o == n) break; }
    if (o < n) nl[o++] = (uint32_t)len;
    return n;
}

// Reads the next block of at most max_recs records of mate file k: decides where the block's text is, cuts it into records
// (out.raw[k]) and returns the base pointer of the offsets in them.
const char *next_block(Source &src, Batch &out, int k, size_t max_recs, int threads, size_t &n_recs)
{
    std::vector<RawRec> &raw = out.raw[k];
    std::vector<uint32_t> &nl = out.nl[k];
    raw.clear();
    n_recs = 0;
    if (!src.gz && src.pos >= src.map_len) return src.map;
    TextBuf &buf = out.text[k];
    size_t end = 0;                                             // gz: bytes of buf in use
    if (src.gz) {
        if (buf.size() < (1u << 22)) buf.resize(1u << 22);
        if (!src.carry.empty()) { if (buf.size() < src.carry.size() + (1u << 22)) buf.resize(src.carry.size() + (1u << 22)); memcpy(buf.data(), src.carry.data(), src.carry.size()); end = src.carry.size(); src.carry.clear(); }
        if (end == 0 && !src.inflate_more(buf, end, 1u << 22) && end == 0) return buf.data();
    }
    auto text0 = [&]() -> const char * { return src.gz ? buf.data() : src.map + src.pos; };
    while (!src.careful) {
        const char first = *text0();
        if (first != '@' && first != '>') { src.careful = true; break; }          // blank lines in front, or not a read file: let the careful splitter say so
        const size_t lpr = first == '@' ? 4 : 2, want = lpr * max_recs;
        // how much text to look at: an estimate from the last block, grown until it holds the lines (gz: inflated until it does)
        size_t n_lines = 0, win = 0;
        double est = src.bytes_per_rec > 0 ? src.bytes_per_rec * 1.03 * (double)max_recs + 4096 : 512.0 * (double)max_recs + (1u << 20);
        for (;;) {
            if (src.gz) {
                if (buf.size() < (size_t)est + (1u << 24)) buf.resize((size_t)est + (1u << 25));      // once, instead of doubling on the way
                while (end < (size_t)est && src.inflate_more(buf, end, 1u << 24)) {}
                win = end;
            } else win = std::min(src.map_len - src.pos, (size_t)est);
            if (win >= ((size_t)1 << 32) - 16) fatal("[soap4-mi355x] %s: a batch of %zu records is larger than 4 GiB of text; lower MP_BATCH_PAIRS\n", src.path.c_str(), max_recs);
            const bool at_eof = src.gz ? src.eof : win == src.map_len - src.pos;
            n_lines = find_newlines(text0(), win, at_eof, want, nl, threads, src.pool.get());
            if (n_lines == want || at_eof) break;
            est *= 1.5;
        }
        const char *t = text0();
        if (n_lines % lpr) {
            // the text ends inside a record: either the file is cut short or this is not the strict shape
            src.careful = true;
            break;
        }
        const size_t n = n_lines / lpr;
        raw.resize(n);
        std::atomic<bool> ok{true};
        const int T = std::max(1, std::min(threads, (int)(n / 8192) + 1));
        fan_out(T, [&](int th) {
            const size_t q0 = n * (size_t)th / (size_t)T, q1 = n * (size_t)(th + 1) / (size_t)T;
            bool good = true;
            for (size_t q = q0; q < q1 && good; q++) {
                RawRec &r = raw[q];
                const size_t l0 = lpr * q;
                r.h = l0 ? (size_t)nl[l0 - 1] + 1 : 0;
                size_t he = nl[l0];
                r.s = he + 1;
                size_t se = nl[l0 + 1];
                r.fasta = lpr == 2; r.wrapped = false;
                if (t[r.h] != first) { good = false; break; }
                if (lpr == 4) {
                    const size_t pl = se + 1, pe = nl[l0 + 2];
                    if (pe <= pl || t[pl] != '+') { good = false; break; }
                    r.q = pe + 1;
                    size_t qe = nl[l0 + 3];
                    while (qe > r.q && t[qe - 1] == '\r') qe--;
                    r.qn = (uint32_t)(qe - r.q);
                } else { r.q = 0; r.qn = 0; }
                while (he > r.h && t[he - 1] == '\r') he--;
                while (se > r.s && t[se - 1] == '\r') se--;
                r.hn = (uint32_t)(he - r.h); r.sn = (uint32_t)(se - r.s);
                if (r.sn == 0 || r.hn == 0) { good = false; break; }                 // blank lines: not the strict shape
                const char c = t[r.s];
                if (c == '@' || c == '>' || c == '+') { good = false; break; }
            }
            if (!good) ok = false;
        }, src.pool.get());
        if (!ok) { src.careful = true; raw.clear(); break; }
        const size_t used = n_lines ? std::min((size_t)nl[n_lines - 1] + 1, win) : 0;
        if (n) src.bytes_per_rec = (double)used / (double)n;
        if (src.gz) { src.carry.assign(buf.data() + used, buf.data() + end); }
        else { src.pos += used; src.touch_ahead(src.pos, (size_t)(src.bytes_per_rec * 1.03 * (double)max_recs) + 4096); }
        n_recs = n;
        return t;
    }
    // ---- careful splitter (sequential)
    const char *t;
    size_t used;
    if (src.gz) {
        used = split_careful([&]() { return (const char *)buf.data(); }, [&](size_t need) { while (end < need && src.inflate_more(buf, end, 1u << 22)) {} return end; }, max_recs, raw, src.path);
        src.carry.assign(buf.data() + used, buf.data() + end);
        t = buf.data();
    } else {
        t = src.map + src.pos;
        const size_t left = src.map_len - src.pos;
        used = split_careful([&]() { return t; }, [&](size_t) { return left; }, max_recs, raw, src.path);
        src.pos += used;
    }
    n_recs = raw.size();
    return t;
}

// Builds the records of mate k from the block's lines and packs the reads as appendToQueryArrays does (QueryParser.cpp:184-234):
// length capped to maxReadLength-1, N/other -> G/A through the character map, 2 bits per base LSB-first, words contiguous per
// read, straight into the interleaved arrays the device takes (read 2q+k).  The name loses a trailing /1 /2
// (QueryParser.cpp:177); the comment is the rest of the header line (kseq).
void build_records(const char *base, const Params &p, Batch &out, int k, size_t n, int threads, Pool *pool)
{
    const uint32_t W = out.W;
    const size_t cap = (size_t)p.max_read_length - 1;
    std::vector<Rec> &recs = out.recs[k];
    const std::vector<RawRec> &raw = out.raw[k];
    if (recs.size() < n) recs.resize(n);
    auto work = [&](size_t q0, size_t q1) {
        std::string joined_s, joined_q;
        for (size_t q = q0; q < q1; q++) {
            const RawRec &rr = raw[q];
            const char *h = base + rr.h, *sq = base + rr.s, *ql = base + rr.q;
            size_t hn = rr.hn, sn = rr.sn, qn = rr.qn;
            Rec &r = recs[q];
            if (rr.wrapped) {
                // a wrapped record: the lines joined, as kseq hands them over
                joined_s.clear(); joined_q.clear();
                for (size_t i = 0; i < sn; i++) if (sq[i] != '\n' && sq[i] != '\r') joined_s += sq[i];
                for (size_t i = 0; i < qn; i++) if (ql[i] != '\n' && ql[i] != '\r') joined_q += ql[i];
                sn = joined_s.size(); qn = joined_q.size();
                sq = joined_s.data(); ql = joined_q.data();
            }
            size_t cut = 1;
            while (cut < hn && h[cut] != ' ' && h[cut] != '\t') cut++;
            size_t nl = cut - 1;
            if (nl > 2 && h[cut - 2] == '/' && h[cut - 1] >= '0' && h[cut - 1] <= '9') nl -= 2;
            r.name = h + 1; r.name_len = (uint32_t)nl;
            r.has_comment = !p.ignore_comments && cut + 1 < hn;
            r.comment = r.has_comment ? h + cut + 1 : nullptr; r.comment_len = r.has_comment ? (uint32_t)(hn - cut - 1) : 0u;
            const size_t keep = std::min(sn, cap);                                          // QueryParser.cpp:188
            uint32_t *dst = out.packed.data() + (2 * q + (size_t)k) * (size_t)W;
            unsigned unclean = 0;
            {
                // eight bases at a time: for upper-case ACGT the code is ((c >> 1) ^ (c >> 2)) & 3 (A 0, C 1, G 2, T 3); the letters are
                // rebuilt from the codes and compared, so anything else (N, lower case, IUPAC) sends the read to the table loop below
                uint64_t bad = 0;
                for (uint32_t w = 0; w < W; w++) {
                    const size_t lo = (size_t)w * 16;
                    uint32_t v = 0;
                    if (lo + 16 <= keep) {
                        for (int half = 0; half < 2; half++) {
                            uint64_t x;
                            memcpy(&x, sq + lo + 8 * (size_t)half, 8);
                            const uint64_t c = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
                            const uint64_t hi1 = (c >> 1) & 0x0101010101010101ull, both = c & hi1;
                            const uint64_t back = 0x4141414141414141ull + 2 * c + 2 * hi1 + 11 * both;
                            bad |= back ^ x;
                            uint64_t y = (c | (c >> 6)) & 0x000F000F000F000Full;
                            y = (y | (y >> 12)) & 0x000000FF000000FFull;
                            y = (y | (y >> 24)) & 0xFFFFull;
                            v |= (uint32_t)y << (16 * half);
                        }
                    } else {
                        const size_t hi = std::min(keep, lo + 16);
                        for (size_t i = lo; i < hi; i++) {
                            const unsigned char ch = (unsigned char)sq[i];
                            v |= (uint32_t)g_code[ch] << (2 * (i - lo));
                            unclean |= g_unclean[ch];
                        }
                    }
                    dst[w] = v;
                }
                if (bad) unclean = 1;
            }
            if (unclean)
                for (uint32_t w = 0; w < W; w++) {
                    uint32_t v = 0;
                    const size_t lo = (size_t)w * 16, hi = std::min(keep, lo + 16);
                    for (size_t i = lo; i < hi; i++) v |= (uint32_t)g_code[(unsigned char)sq[i]] << (2 * (i - lo));
                    dst[w] = v;
                }
            out.lens[2 * q + (size_t)k] = (uint32_t)keep;
            if (unclean || rr.wrapped) {
                std::string &o = r.owned().seq;
                o.resize(keep);
                for (size_t i = 0; i < keep; i++) o[i] = "ACGT"[g_code[(unsigned char)sq[i]]];
                r.seq = o.data();
            } else r.seq = sq;
            r.seq_len = (uint32_t)keep;
            const size_t qkeep = rr.fasta ? 0 : std::min(qn, keep);
            if (rr.wrapped && qkeep) { std::string &o = r.owned().qual; o.assign(ql, qkeep); r.qual = o.data(); } else r.qual = ql;
            r.qual_len = (uint32_t)qkeep;
        }
    };
    const int T = std::max(1, std::min(threads, (int)(n / 4096) + 1));
    fan_out(T, [&](int t) { work(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); }, pool);
}

// reads the next batch: the two files are cut into records side by side, then packed side by side
bool fill_batch(Source &a, Source &b, const Params &p, Batch &out, int threads)
{
    const size_t bp = p.batch_pairs;
    out.W = (uint32_t)(p.max_read_length + 15) / 16;
    size_t na = 0, nb = 0;
    const char *base[2] = {nullptr, nullptr};
    // -T is the budget of the whole host side: the two files' splitters take three eighths each, the printer's formatters half (they
    // are never all busy at once; measured best at -T 16 on a 16-CPU share; MP_READER_THREADS / MP_PRINTER_THREADS override)
    static const int reader_threads = getenv("MP_READER_THREADS") ? std::max(1, atoi(getenv("MP_READER_THREADS"))) : std::max(1, threads * 3 / 8);
    const int per_file = reader_threads;
    if (!a.pool) a.pool.reset(new Pool(per_file - 1));
    if (!b.pool) b.pool.reset(new Pool(per_file - 1));
    const double t0 = now_s();
    {
        std::thread second([&]() { base[1] = next_block(b, out, 1, bp, per_file, nb); });
        base[0] = next_block(a, out, 0, bp, per_file, na);
        second.join();
    }
    const double t1 = now_s();
    if (na != nb) fatal("Error: number of sequences of pair-end files not matched.\n");
    out.n_reads = (uint32_t)(2 * na);
    out.last = na < bp;
    out.packed.resize(2 * na * (size_t)out.W); out.lens.resize(2 * na);
    {
        std::thread second([&]() { build_records(base[1], p, out, 1, nb, per_file, b.pool.get()); });
        build_records(base[0], p, out, 0, na, per_file, a.pool.get());
        second.join();
    }
    for (auto &c : out.ch) c.clear();
    if (getenv("MP_TIMING")) fprintf(stderr, "[Main] reader: %zu pairs: lines found in %.3f s, records built and packed in %.3f s (%d threads per file)\n", na, t1 - t0, now_s() - t1, per_file);
    return na > 0;
}

// ------------------------------------------------------------------------------------------------ taxonomy (extension -lsamid / -pass1)
// What cc/taxLookupAcc does to an LSAM line (cc/taxLookupAcc.cpp:49-93): every listed accession becomes the taxid of its
// species (TaxDB::popUpToSpecies, cc/taxonomy.h:125-130), a species keeps the best of its accessions' scores, the hits are
// printed in ascending taxid order and the scientific names of the hits' super-kingdoms are appended.  Only accessions that
// name a sequence of the resident index chunks can ever be looked up, so only those rows of the accession2taxid file are kept.
struct Taxonomy {
    std::vector<int> parent;
    std::vector<uint8_t> rank;                    // 1 species, 2 superkingdom
    std::vector<std::string> sci_name;
    std::unordered_map<std::string, int> acc2tid;
    mutable std::mutex mu;
    mutable std::unordered_map<std::string, int> species_of_name;   // sequence name -> species taxid, -1 = accession unknown (memo)

    // removeVersion / getCorrectAcc (cc/taxonomy.h:14-24, cc/misc.h:29-45)
    static std::string remove_version(const std::string &a)
    {
        for (int i = (int)a.size() - 1; i >= 0; --i) {
            if (a[i] == '.') return a.substr(0, (size_t)i);
            if (!isdigit((unsigned char)a[i])) return a;
        }
        return a;
    }
    static std::string correct_acc(const std::string &h)
    {
        size_t s0 = h.find('|');
        if (s0 == std::string::npos || h.substr(0, s0) != "gi") return h;
        s0 = h.find('|', s0 + 1);
        s0 = h.find('|', s0 + 1);
        const size_t s1 = h.find('|', s0 + 1);
        return h.substr(s0 + 1, s1 - s0 - 1);
    }
    void ensure(int id) { if (id >= (int)parent.size()) { const size_t n = (size_t)((id + 1) * 1.5); parent.resize(n, 0); rank.resize(n, 0); sci_name.resize(n); } }
    static bool read_lines(const std::string &path, const std::function<void(char *)> &fn)
    {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) return false;
        gzbuffer(f, 1 << 20);
        std::vector<char> line(1 << 16);
        while (gzgets(f, line.data(), (int)line.size())) fn(line.data());
        gzclose(f);
        return true;
    }
    bool load(const std::string &acc_path, const std::string &nodes_path, const std::string &names_path, const std::unordered_set<std::string> &wanted)
    {
        parent.assign(2000000, 0); rank.assign(2000000, 0); sci_name.assign(2000000, std::string());        // TaxDB::kInitNumTax
        bool ok = read_lines(nodes_path, [&](char *l) {                                  // tid | parent | rank | ...   (TaxDB::readNode)
            char *sv = nullptr;
            const char *t0 = strtok_r(l, " \t\r\n", &sv); (void)strtok_r(nullptr, " \t\r\n", &sv);
            const char *t2 = strtok_r(nullptr, " \t\r\n", &sv); (void)strtok_r(nullptr, " \t\r\n", &sv);
            const char *t4 = strtok_r(nullptr, " \t\r\n", &sv);
            if (!t0 || !t2 || !t4) return;
            const int id = atoi(t0);
            if (id < 0) return;
            ensure(id);
            parent[(size_t)id] = atoi(t2);
            rank[(size_t)id] = !strcmp(t4, "species") ? 1 : !strcmp(t4, "superkingdom") ? 2 : 0;
        });
        ok = ok && read_lines(names_path, [&](char *l) {                                 // TaxDB::readName: the "scientific name" rows
            if (!strstr(l, "scientific name")) return;
            char *sv = nullptr;
            const char *t0 = strtok_r(l, " \t\r\n", &sv);
            if (!t0) return;
            const int id = atoi(t0);
            (void)strtok_r(nullptr, " \t\r\n", &sv);                                   // the '|' after the id
            std::string name;
            for (const char *w = strtok_r(nullptr, " \t\r\n", &sv); w && strcmp(w, "|") != 0; w = strtok_r(nullptr, " \t\r\n", &sv)) { if (!name.empty()) name += ' '; name += w; }
            if (id >= 0) { ensure(id); sci_name[(size_t)id] = name; }
        });
        ok = ok && read_lines(acc_path, [&](char *l) {                                   // accession  accession.version  taxid  gi  (TaxDB::initAccToTax)
            char *sv = nullptr;
            (void)strtok_r(l, " \t\r\n", &sv);
            const char *acc = strtok_r(nullptr, " \t\r\n", &sv), *tid = strtok_r(nullptr, " \t\r\n", &sv);
            if (!acc || !tid) return;
            const std::string key = remove_version(acc);
            if (wanted.empty() || wanted.count(key)) acc2tid[key] = atoi(tid);        // an empty set asks for every row
        });
        return ok;
    }
    int species_tid(int tid) const { while (tid != 1 && tid != 0 && !(tid < (int)rank.size() && rank[(size_t)tid] == 1)) tid = tid < (int)parent.size() ? parent[(size_t)tid] : 0; return tid; }
    int superkingdom(int tid) const { while (tid != 1 && tid != 0) { if (tid < (int)rank.size() && rank[(size_t)tid] == 2) return tid; tid = tid < (int)parent.size() ? parent[(size_t)tid] : 0; } return -1; }
    int species_of(const std::string &seq_name) const
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = species_of_name.find(seq_name);
        if (it != species_of_name.end()) return it->second;
        auto a = acc2tid.find(remove_version(correct_acc(seq_name)));
        int sp = -1;
        if (a != acc2tid.end()) sp = species_tid(a->second);
        else fprintf(stderr, "Error: Taxid not found for %s\n", correct_acc(seq_name).c_str());
        species_of_name.emplace(seq_name, sp);
        return sp;
    }
};

// Pass 1 of cc/reassign (cc/reassign.cpp:64-110) over the LSAM.id lines as they are produced: per species taxid the reads that
// list it, the reads that list only it, and per pair of taxids the reads that list both -- the counters the ranks of a
// multi-process run all-reduce (megapath_amd/taxcount.py).
struct Pass1 {
    std::map<int, long long> counts, uniq;
    std::map<std::pair<int, int>, long long> intersect;      // (larger taxid, smaller taxid): pairup(), cc/reassign.cpp:29-32
    void line(int best_score, const std::map<int, double> &hits, double t)
    {
        if (best_score < t || hits.empty()) return;
        std::vector<int> seen;
        for (const auto &h : hits) {
            counts[h.first]++;
            if (hits.size() == 1) uniq[h.first]++;
            else for (int s : seen) intersect[{std::max(s, h.first), std::min(s, h.first)}]++;
            seen.push_back(h.first);
        }
    }
    void add(const Pass1 &o)
    {
        for (const auto &x : o.counts) counts[x.first] += x.second;
        for (const auto &x : o.uniq) uniq[x.first] += x.second;
        for (const auto &x : o.intersect) intersect[x.first] += x.second;
    }
    bool write(const std::string &path, double t) const
    {
        FILE *f = fopen(path.c_str(), "w");
        if (!f) return false;
        fprintf(f, "#reassign-pass1\tt=%g\n", t);
        for (const auto &x : counts) fprintf(f, "counts\t%d\t%lld\n", x.first, x.second);
        for (const auto &x : uniq) fprintf(f, "uniq\t%d\t%lld\n", x.first, x.second);
        for (const auto &x : intersect) fprintf(f, "intersect\t%d\t%d\t%lld\n", x.first.first, x.first.second, x.second);
        return fclose(f) == 0;
    }
};

// ------------------------------------------------------------------------------------------------ output
struct ChrHit { int chr, neg; bool operator<(const ChrHit &o) const { return chr != o.chr ? chr < o.chr : neg < o.neg; } };
struct HdrMap { int score; std::string_view text; };

// atoi over a view (leading white space, optional sign, digits)
int view_atoi(const char *c, const char *e)
{
    while (c < e && (*c == ' ' || (*c >= '\t' && *c <= '\r'))) c++;
    bool neg = false;
    if (c < e && (*c == '-' || *c == '+')) { neg = *c == '-'; c++; }
    long v = 0;
    while (c < e && *c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); if (v > 0x7fffffffL) { v = 0x7fffffffL; } c++; }
    return (int)(neg ? -v : v);
}

// getMappingFromHeader (BGS-IO.cpp:1348-1371), quirks included (the record filter tests the header's best score); the texts of the
// kept entries are views into the comment
int mapping_from_header(const Rec &r, std::vector<HdrMap> &v, double top, double scoreT)
{
    v.clear();
    if (!r.has_comment || r.ignore()) return 0;
    const char *c = r.comment, *e = c + r.comment_len;
    if (r.comment_len < 6) return 0;
    int score = view_atoi(c + 6, e);
    if (score < scoreT) return score;
    else if (scoreT < score * top) scoreT = score * top;
    const char *p = (const char *)memchr(c + 6, ';', (size_t)(e - (c + 6)));
    while (p && p + 1 < e) {
        const char *s = p + 1;
        const char *q = (const char *)memchr(s, ';', (size_t)(e - s));
        if (!q) break;
        if (score >= scoreT) v.push_back(HdrMap{view_atoi(s, q), std::string_view(s, (size_t)(q - s))});
        p = q;
    }
    return score;
}

// "%d" without the formatting machinery (the printer writes millions of these per second)
inline void append_int(std::string &o, int v)
{
    char buf[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf[n++] = '-';
    while (n) o += buf[--n];
}

struct Writer {
    const mp_index *ix;
    const Params &p;
    std::string &out;           // the caller's buffer (kept between calls: its capacity is the steady state's)
    // carry = true: this is not the last resident index chunk.  Instead of printing a record, its new header comment
    // ("SCORE:..." or "IGNORE") replaces the record's own: exactly what the next run of the chained NT loop would read back
    // from this run's stdout (runMegaPath.sh:191-227, getMappingFromHeader BGS-IO.cpp:1348-1371), without the file round trip.
    bool carry;
    // mute = true: the record's text is made elsewhere (output_reduced, from the device's reduced list); pair() / single() are still called
    // for what they do to the RESULTS -- the BAM writer that runs next sees boundary-spanning mates un-aligned and the scores normalised
    bool mute = false;
    Writer(const mp_index *i, const Params &pp, std::string &out_, bool carry_ = false) : ix(i), p(pp), out(out_), carry(carry_) {}
    // -lsam: the line cc/fastq2lsam would print for a record (print_lsam_line, cc/fastq2lsam.cpp:28-78)
    void lsam_line(std::string_view name, const std::string &comm, std::string_view seq, std::string_view qual, int which_end)
    {
        char tmp[32];
        out += name;
        out += which_end == 1 ? "\t64\t" : which_end == 2 ? "\t128\t" : "\t0\t";
        const bool ignore = comm == "IGNORE";
        const int score = ignore ? -1 : (comm.size() > 6 ? atoi(comm.c_str() + 6) : 0);
        snprintf(tmp, sizeof tmp, "%d\t", score); out += tmp;
        if (p.lsam == 1) { out += seq; out += '\t'; out += qual; out += '\t'; } else out += "*\t*\t";
        if (p.tax) {
            // taxLookupAcc (cc/taxLookupAcc.cpp:59-90): species taxid of every listed accession, best score per species, ascending taxid
            std::map<int, double> by_tid;
            if (score > 0) {
                size_t i = comm.find(';');
                while (i != std::string::npos && i + 1 < comm.size()) {
                    size_t j = comm.find(';', i + 1);
                    if (j == std::string::npos) j = comm.size();
                    size_t c0 = comm.find(',', i + 1);
                    if (c0 != std::string::npos && c0 < j) {
                        const double sc = atof(comm.c_str() + i + 1);
                        for (size_t a = c0; a < j;) {
                            size_t b = comm.find(',', a + 1);
                            if (b == std::string::npos || b > j) b = j;
                            const int sp = p.tax->species_of(comm.substr(a + 1, b - a - 1));
                            if (sp >= 0) { double &v = by_tid[sp]; v = std::max(v, sc); }
                            a = b;
                        }
                    }
                    i = j;
                }
            }
            if (by_tid.empty()) out += '*';
            else {
                bool first = true;
                for (const auto &h : by_tid) { if (!first) out += ';'; first = false; snprintf(tmp, sizeof tmp, "%g,%d", h.second, h.first); out += tmp; }
            }
            if (ignore) out += "\tIGNORE";
            std::vector<int> sk;
            for (const auto &h : by_tid) { const int k = p.tax->superkingdom(h.first); if (k >= 0) sk.push_back(k); }
            std::sort(sk.begin(), sk.end());
            sk.erase(std::unique(sk.begin(), sk.end()), sk.end());
            for (int k : sk) { out += '\t'; out += p.tax->sci_name[(size_t)k]; }
            out += '\n';
            pass1.line(score, by_tid, p.pass1_t);
            return;
        }
        if (score <= 0) out += '*';
        else {
            bool first = true;
            size_t i = comm.find(';');                       // element 0 is the score itself
            while (i != std::string::npos && i + 1 < comm.size()) {
                size_t j = comm.find(';', i + 1);
                if (j == std::string::npos) j = comm.size();
                // one "score,acc[,acc...]" element
                size_t c0 = comm.find(',', i + 1);
                if (c0 == std::string::npos || c0 > j) c0 = j;
                for (size_t a = c0; a < j;) {
                    size_t b = comm.find(',', a + 1);
                    if (b == std::string::npos || b > j) b = j;
                    if (!first) out += ';';
                    first = false;
                    out.append(comm, i + 1, c0 - i - 1); out += ','; out.append(comm, a + 1, b - a - 1);
                    a = b;
                }
                i = j;
            }
        }
        if (ignore) out += "\tIGNORE";
        out += '\n';
    }
    // fastq2lsam's pairing of consecutive records with equal names (cc/fastq2lsam.cpp:92-112); every pair leaves this
    // writer as two consecutive records, so the state never spans the chunks of the output stage
    void lsam_record(const Rec &r, const std::string &comm)
    {
        if (held_) {
            if (held_->name_v() == r.name_v()) { lsam_line(held_->name_v(), held_comm_, held_->seq_v(), held_->qual_v(), 1); lsam_line(r.name_v(), comm, r.seq_v(), r.qual_v(), 2); held_ = nullptr; return; }
            lsam_line(held_->name_v(), held_comm_, held_->seq_v(), held_->qual_v(), 0);
        }
        held_ = &r; held_comm_ = comm;
    }
    void finish() { if (held_) { lsam_line(held_->name_v(), held_comm_, held_->seq_v(), held_->qual_v(), 0); held_ = nullptr; } }
    // one output record: the annotated FASTQ header + sequence + qualities, or its LSAM line
    void record(const Rec &r, int best, const std::vector<ChrHit> &hits, const std::vector<HdrMap> &v)
    {
        if (mute) return;
        const bool to_comment = p.lsam || carry;
        std::string &o = to_comment ? comm_ : out;
        if (to_comment) comm_.clear(); else { out += '@'; out.append(r.name, r.name_len); out += '\t'; }
        if (r.ignore()) o += "IGNORE";
        else {
            o += "SCORE:"; append_int(o, best); o += ';';
            if (best > 0) {
                for (size_t i = 0; i < hits.size(); i++) {
                    if (i > 0 && hits[i].chr == hits[i - 1].chr) continue;
                    if (-hits[i].neg > 0 && -hits[i].neg >= best * p.top) {
                        append_int(o, -hits[i].neg); o += ',';
                        o += mp_index_seq_name(ix, (uint32_t)hits[i].chr); o += ';';
                    }
                }
            }
            for (const HdrMap &m : v) if (m.score >= best * p.top) { o += m.text; o += ';'; }
        }
        if (carry) {
            // the record is this thread's alone (pairs are split between the formatting threads); v points into the old comment,
            // which is only let go of now
            Rec &w = const_cast<Rec &>(r);
            std::string &own = w.owned().comment;
            own.swap(comm_);
            w.comment = own.data(); w.comment_len = (uint32_t)own.size();
            w.has_comment = true;
            return;
        }
        if (p.lsam) { lsam_record(r, comm_); return; }
        out += '\n'; out.append(r.seq, r.seq_len); out += "\n+\n"; out.append(r.qual, r.qual_len); out += '\n';
    }
    // One output record from the device's reduced list (mp_reduce_finish): k[0..n) are the read's keys -- last chunk first, sequences
    // ascending, one per (chunk, sequence) with its best score.  What the chain of per-chunk merges leaves in the header
    // (pairDeepDPOutputFastqAPI / unproperlypairDPOutputFastqAPI + getMappingFromHeader, chunk after chunk): the best score over all
    // chunks and the record's own comment, then every entry that reaches top% of it.
    void reduced(const Rec &r, const uint64_t *k, size_t n, mp_index *const *chunks, uint32_t n_chunks)
    {
        int best = 0;
        for (size_t i = 0; i < n; i++) best = std::max(best, 1023 - (int)(k[i] & 1023u));
        const int prev = mapping_from_header(r, v1_, p.top, best * p.top);
        best = std::max(best, prev);
        const bool to_comment = p.lsam != 0;
        std::string &o = to_comment ? comm_ : out;
        if (to_comment) comm_.clear(); else { out += '@'; out.append(r.name, r.name_len); out += '\t'; }
        if (r.ignore()) o += "IGNORE";
        else {
            o += "SCORE:"; append_int(o, best); o += ';';
            if (best > 0)
                for (size_t i = 0; i < n; i++) {
                    const int sc = 1023 - (int)(k[i] & 1023u);
                    if (sc > 0 && sc >= best * p.top) {
                        const uint32_t chunk = n_chunks - 1 - (uint32_t)((k[i] >> 32) & 1023u), sq = (uint32_t)((k[i] >> 10) & 0x3FFFFFu);
                        append_int(o, sc); o += ',';
                        o += mp_index_seq_name(chunks[chunk], sq); o += ';';
                    }
                }
            for (const HdrMap &m : v1_) if (m.score >= best * p.top) { o += m.text; o += ';'; }
        }
        if (p.lsam) { lsam_record(r, comm_); return; }
        out += '\n'; out.append(r.seq, r.seq_len); out += "\n+\n"; out.append(r.qual, r.qual_len); out += '\n';
    }
    const Rec *held_ = nullptr;
    std::string held_comm_, comm_;
    Pass1 pass1;                  // counters of the lines this writer printed (-lsamid)
    // decideTargetChr (BGS-IO.cpp:1312-1341)
    int target_chr(uint64_t pos, uint32_t len) const
    {
        uint32_t c = 0;
        mp_index_target_chr(ix, pos, len, &c);
        return c ? (int)c : -1;
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
        int p1 = mapping_from_header(r1, v1_, p.top, best1 * p.top), p2 = mapping_from_header(r2, v2_, p.top, best2 * p.top);
        best1 = std::max(best1, p1); best2 = std::max(best2, p2);
        record(r1, best1, h1_, v1_);
        record(r2, best2, h2_, v2_);
    }
    // unproperlypairDPOutputFastqAPI (BGS-IO.cpp:1384-1446); in -F mode hits that span a sequence boundary are un-aligned in place
    void single(const Rec &r, uint32_t len, std::vector<Side> &hits)
    {
        if (r.ignore()) { h1_.clear(); v1_.clear(); record(r, 0, h1_, v1_); return; }
        int best = 0;
        h1_.clear();
        if (p.mode != 2)
            for (Side &hs : hits) {
                int c = target_chr(hs.pos, len);
                if (c < 0) { hs.pos = NOT_ALIGNED; hs.score = 0; } else h1_.push_back(ChrHit{c, -hs.score});
                best = std::max(best, hs.score);
            }
        std::sort(h1_.begin(), h1_.end());
        int prev = mapping_from_header(r, v1_, p.top, best * p.top);
        best = std::max(best, prev);
        record(r, best, h1_, v1_);
    }
    std::vector<ChrHit> h1_, h2_;
    std::vector<HdrMap> v1_, v2_;
};

void die(int rc, const char *what) { if (rc) fatal("[soap4-mi355x] %s failed: %s (%d)\n", what, mp_strerror(rc), rc); }

struct Stats {
    uint64_t pairs = 0, deep = 0, rescued = 0, unpaired = 0, dp = 0, floor_pairs = 0, floor_mate_pairs = 0;
    double t_read = 0, t_upload = 0, t_deep = 0, t_sort = 0, t_cigar = 0, t_emit_deep = 0, t_rescue = 0, t_emit_rest = 0;   // seconds (MP_TIMING=1 prints them)
};

// -b only: the special CIGAR of each listed DP instance.  The hits name their instance (read, strand, window); the same
// kernels run it again with the op strings switched on (mp_sgdp_batch), in bounded chunks.
void resolve_cigars(mp_ctx *ctx, const Params &p, int threads, const std::vector<mp_dp_task> &tasks, const std::vector<uint32_t> &lens, std::vector<std::string> &out)
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
        auto convert = [&](size_t t0, size_t t1) {
            for (size_t t = t0; t < t1; t++) {
                if (res[t].score <= 0 || res[t].n_ops == 0) fatal("[soap4-mi355x] internal error: a reported hit did not align again\n");
                // CigarStringEncoder (DV-DPfunctions.h:377-427): runs collected in traceback order, printed from the read's start
                const uint8_t *o = ops.data() + t * (size_t)stride;
                std::string &cg = out[off + t];
                char num[16];
                for (int i = (int)res[t].n_ops - 1; i >= 0;) {
                    int j = i;
                    while (j >= 0 && o[j] == o[i]) j--;
                    cg.append(num, (size_t)snprintf(num, sizeof num, "%d", i - j)); cg += (char)o[i];
                    i = j;
                }
            }
        };
        const size_t nt = (size_t)std::max(1, std::min(threads, (int)(cnt / 4096) + 1));
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; t++) pool.emplace_back(convert, cnt * t / nt, cnt * (t + 1) / nt);
        convert(0, cnt / nt);
        for (auto &th : pool) th.join();
    }
}

struct BamState {
    std::unique_ptr<mpbam::Writer> w;
    bool want_cov = false;                                   // -spike: keep what bamtobed would list of the .dpout / .gout records
    std::vector<mpbam::CovRec> cov;
    int stage_rescue = 2, stage_unpaired = 3;                // hspaux->dpStageId (alignment.cpp:93,228,308)
};

// (a read without qualities -- FASTA input -- gets blanks: the BAM record then carries 0xff, "no quality")
const std::string g_blank_qual(1024, ' ');
mpbam::ReadView view_of(const Rec &r, uint32_t len) { return mpbam::ReadView{r.name, r.name_len, r.seq, r.qual_len >= len || len > g_blank_qual.size() ? r.qual : g_blank_qual.data(), len}; }

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

// OutputBuffer::ready for the three hit lists of a batch: sorted by (positions, scores), exact duplicates dropped
// (DV-DPfunctions.h:165-196; :853-866 for the single-end hits; ready(1) also drops failed rescues, which never get here)
void prepare_lists(Batch &bt, Batch::ChunkHits &b, int threads)
{
    if (b.lists_ready) return;
    auto &hits = b.hits;
    auto hit_less = [](const mp_pair_hit &x, const mp_pair_hit &y) {
        if (x.pair_id != y.pair_id) return x.pair_id < y.pair_id;
        if (x.algnmt_1 != y.algnmt_1) return x.algnmt_1 < y.algnmt_1;
        if (x.algnmt_2 != y.algnmt_2) return x.algnmt_2 < y.algnmt_2;
        if (x.score_1 != y.score_1) return x.score_1 < y.score_1;
        return x.score_2 < y.score_2;
    };
    // order the few hits of each pair, slices of the list side by side (a list that is not grouped by pair is bucketed first)
    const uint32_t n_pairs = bt.n_reads / 2;
    bool grouped_by_pair = true;                    // the library hands the list over grouped by pair, pairs ascending (megapath_hip.h)
    for (size_t i = 1; i < hits.size() && grouped_by_pair; i++) grouped_by_pair = hits[i].pair_id >= hits[i - 1].pair_id;
    if (!grouped_by_pair) {
        std::vector<uint32_t> start((size_t)n_pairs + 2, 0u);
        for (const mp_pair_hit &h : hits) start[(size_t)h.pair_id + 2]++;
        for (size_t q = 2; q < start.size(); q++) start[q] += start[q - 1];          // start[q+1] = first slot of pair q
        std::vector<mp_pair_hit> bucketed(hits.size());
        for (const mp_pair_hit &h : hits) bucketed[start[(size_t)h.pair_id + 1]++] = h;
        hits.swap(bucketed);
    }
    auto sort_groups = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi;) {
            size_t j = i + 1;
            while (j < hi && hits[j].pair_id == hits[i].pair_id) j++;
            if (j - i > 1) std::sort(hits.begin() + (long)i, hits.begin() + (long)j, hit_less);
            i = j;
        }
    };
    {
        const size_t nt = (size_t)std::max(1, std::min(threads, (int)(hits.size() / 65536) + 1));
        std::vector<size_t> cut(nt + 1, hits.size());
        cut[0] = 0;
        for (size_t t = 1; t < nt; t++) {
            size_t c = hits.size() * t / nt;
            while (c < hits.size() && c > 0 && hits[c].pair_id == hits[c - 1].pair_id) c++;
            cut[t] = std::max(c, cut[t - 1]);
        }
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; t++) pool.emplace_back(sort_groups, cut[t], cut[t + 1]);
        sort_groups(cut[0], cut[1]);
        for (auto &th : pool) th.join();
    }
    hits.erase(std::unique(hits.begin(), hits.end(), [](const mp_pair_hit &x, const mp_pair_hit &y) {
        return x.pair_id == y.pair_id && x.algnmt_1 == y.algnmt_1 && x.algnmt_2 == y.algnmt_2 && x.score_1 == y.score_1 && x.score_2 == y.score_2; }), hits.end());
    auto &rows = b.rows;
    rows.resize(b.resc.size());
    for (size_t i = 0; i < b.resc.size(); i++) {
        const mp_rescue_hit &h = b.resc[i];
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
    for (const RescueRow &r : rows) b.done[r.pair] = 2;
    auto &singles = b.singles;
    auto single_less = [](const mp_single_hit &x, const mp_single_hit &y) {
        if (x.read_id != y.read_id) return x.read_id < y.read_id;
        if (x.algnmt != y.algnmt) return x.algnmt < y.algnmt;
        return x.score < y.score;
    };
    // single-end hits arrive in task order, i.e. grouped by read: only each read's few hits need ordering
    bool grouped = true;
    for (size_t i = 1; i < singles.size() && grouped; i++) grouped = singles[i].read_id >= singles[i - 1].read_id;
    if (!grouped) std::sort(singles.begin(), singles.end(), single_less);
    else
        for (size_t i = 0; i < singles.size();) {
            size_t j = i + 1;
            while (j < singles.size() && singles[j].read_id == singles[i].read_id) j++;
            if (j - i > 1) std::sort(singles.begin() + (long)i, singles.begin() + (long)j, single_less);
            i = j;
        }
    singles.erase(std::unique(singles.begin(), singles.end(), [](const mp_single_hit &x, const mp_single_hit &y) {
        return x.read_id == y.read_id && x.algnmt == y.algnmt && x.score == y.score; }), singles.end());
    b.lists_ready = true;
}

// Stage 2: everything that touches the device.  Per batch and index chunk: deep-DP round, and for the pairs it left without
// a hit the rescue round (alignment.cpp:29-355); with -b also the CIGARs of every hit that will be written.  The reads of the
// batch are uploaded once (upload_reads); the context is bound to one resident chunk after the other.
void upload_reads(mp_ctx *ctx, Batch &bt, Stats &st)
{
    const double t0 = now_s();
    die(mp_set_reads(ctx, bt.packed.data(), bt.W, bt.lens.data(), bt.n_reads), "mp_set_reads");
    st.t_upload += now_s() - t0;
}

void gpu_stage(mp_ctx *ctx, const Params &p, Batch &bt, Batch::ChunkHits &b, bool bam, int threads, Stats &st,
               int reduce_chunk = -1 /* >= 0: the batch's per-reference scores are being reduced on the device (mp_reduce_*) and this is the LAST chunk, whose hit lists -b needs on the host as well */)
{
    const uint32_t n_reads = bt.n_reads, n_pairs = n_reads / 2;
    double t0 = now_s();
    mp_align_params ap;
    ap.insert_low = p.insert_low; ap.insert_high = p.insert_high; ap.clip_left = p.clip_left; ap.clip_right = p.clip_right; ap.mmp = p.mmp;
    ap.want_tie_counts = bam ? 1 : 0; ap.reserved = 0;          // num_sameScore only feeds the BAM MAPQ / X0 / X1
    uint64_t nh = 0;
    mp_round_stats rs;
    die(mp_deep_dp_round(ctx, &ap, &nh, &rs), "mp_deep_dp_round");
    const mp_pair_hit *view = nullptr;
    die(mp_get_pair_hits_view(ctx, &view, &nh), "mp_get_pair_hits_view");
    b.hits.assign(view, view + nh);
    b.dp_instances = 2 * rs.n_candidates;
    b.done.assign(n_pairs, 0);
    uint32_t n_done = 0;
    for (const mp_pair_hit &h : b.hits) if (!b.done[h.pair_id]) { b.done[h.pair_id] = 1; n_done++; }
    st.t_deep += now_s() - t0; t0 = now_s();
    if (n_done < n_pairs) {
        // rescue passes (alignment.cpp:137-352): single-end DP + mate rescue on the device
        mp_rescue_stats rst;
        die(mp_rescue_round(ctx, &ap, p.skip_default_dp ? 1 : 0, &rst), "mp_rescue_round");
        b.dp_instances += rst.n_single_tasks + rst.n_rescue_tasks;
        uint64_t ns = 0, nr = 0;
        const mp_single_hit *sv = nullptr;
        const mp_rescue_hit *rv = nullptr;
        die(mp_get_single_hits_view(ctx, &sv, &ns), "mp_get_single_hits_view");
        b.singles.assign(sv, sv + ns);
        die(mp_get_rescue_hits_view(ctx, &rv, &nr), "mp_get_rescue_hits_view");
        b.resc.assign(rv, rv + nr);
    }
    st.t_rescue += now_s() - t0; t0 = now_s();
    if (reduce_chunk >= 0) {
        // (before the CIGAR batches below reuse the context's DP buffers: the hit lists are still on the device)
        mp_reduce_stats red{};
        bt.red_class.resize(n_pairs + 1);
        die(mp_reduce_chunk(ctx, p.mode, (uint32_t)reduce_chunk, &red, bt.red_class.data()), "mp_reduce_chunk");
        bt.red_last = red;
        bt.red_dp_instances += b.dp_instances;
    }
    if (getenv("MP_TIMING"))
        fprintf(stderr, "[Main] batch %llu: %llu seed positions, %llu candidates, %llu pair hits, %u of %u pairs done, %zu single hits, %zu rescues (insert %d..%d)\n",
                (unsigned long long)bt.seq, (unsigned long long)rs.seed.n_seedpos, (unsigned long long)rs.n_candidates, (unsigned long long)nh, n_done, n_pairs,
                b.singles.size(), b.resc.size(), ap.insert_low, ap.insert_high);
    // ordering and de-duplicating the hit lists is done here, by the worker that owns the batch (several workers side by side), not by
    // the one printer thread every batch has to pass through in input order
    prepare_lists(bt, b, std::max(1, threads / 4));         // (several workers do this side by side)
    st.t_sort += now_s() - t0; t0 = now_s();
    if (bam) {
        std::vector<mp_dp_task> tasks;
        tasks.reserve(2 * b.hits.size());
        for (const mp_pair_hit &h : b.hits) {
            const uint32_t s1 = h.left_is_mate ? 2u : 1u;       // the left leg is aligned as is, the right leg reverse-complemented
            tasks.push_back(mp_dp_task{2 * h.pair_id, s1, h.start_1, h.dplen_1, 0});
            tasks.push_back(mp_dp_task{2 * h.pair_id + 1, 3u - s1, h.start_2, h.dplen_2, 0});
        }
        resolve_cigars(ctx, p, threads, tasks, bt.lens, b.cig);
        tasks.clear();
        for (const RescueRow &r : b.rows) {
            const mp_rescue_hit &h = b.resc[r.src];
            tasks.push_back(mp_dp_task{h.aligned_read, h.aligned_strand, h.aligned_win_start, h.aligned_win_len, 0});
            tasks.push_back(mp_dp_task{h.aligned_read ^ 1u, h.dp_strand, h.dp_win_start, h.dp_win_len, 0});
        }
        resolve_cigars(ctx, p, threads, tasks, bt.lens, b.rcig);
        tasks.clear();
        for (const mp_single_hit &h : b.singles) if (b.done[h.read_id >> 1] == 0) tasks.push_back(mp_dp_task{h.read_id, h.strand, h.win_start, h.win_len, 0});
        resolve_cigars(ctx, p, threads, tasks, bt.lens, b.scig);
        st.t_cigar += now_s() - t0;
    }
}

// Stage 3: per-pair bookkeeping and printing (OutputDPResult.cpp:65-404, BGS-IO.cpp writers).  Order within a batch as in
// the reference: pairs of the deep-DP round, then every rescued pair, then the unpaired rest (alignment.cpp:105-330).
// Pairs are independent, so each section is cut into chunks that worker threads format into their own buffers; the
// buffers go out in chunk order.
struct Sink { std::string text, bam[3]; uint64_t n = 0; Pass1 pass1; std::vector<mpbam::CovRec> cov; };

// the printer's threads and their buffers: both live as long as the run (the buffers keep the capacity of the steady state --
// fresh strings cost a page fault per 4 KiB of output)
struct Formatters {
    Pool pool;
    std::vector<Sink> sinks;
    explicit Formatters(int threads) : pool(std::max(1, threads) - 1), sinks((size_t)std::max(1, threads)) {}
};

template <typename F>
void for_chunks(size_t n_items, Formatters &fm, BamState *bam, Pass1 *pass1, F &&body)
{
    const size_t chunk = 4096;
    const size_t n_chunks = (n_items + chunk - 1) / chunk;
    const size_t wave = fm.sinks.size();
    std::vector<Sink> &sinks = fm.sinks;
    for (size_t c0 = 0; c0 < n_chunks; c0 += wave) {
        const size_t nc = std::min(wave, n_chunks - c0);
        const std::function<void(int)> run = [&](int t) {
            Sink &s = sinks[(size_t)t];
            s.text.clear(); s.pass1 = Pass1(); s.cov.clear();
            for (auto &x : s.bam) x.clear();
            body((c0 + (size_t)t) * chunk, std::min(n_items, (c0 + (size_t)t + 1) * chunk), s);
        };
        fm.pool.run((int)nc, run);
        for (size_t t = 0; t < nc; t++) {
            fwrite(sinks[t].text.data(), 1, sinks[t].text.size(), stdout);
            if (bam) for (int k = 0; k < 3; k++) bam->w->append(k, sinks[t].bam[k]);
            if (bam && bam->want_cov) bam->cov.insert(bam->cov.end(), sinks[t].cov.begin(), sinks[t].cov.end());
            if (pass1) pass1->add(sinks[t].pass1);
        }
    }
}

// `carry`: not the last resident chunk -- the records keep their new header comments for the next chunk instead of being printed.
void output_stage(mp_index *ix, const Params &p, Batch &bt, Batch::ChunkHits &b, BamState *bam, int threads, Formatters &fm, Stats &st, bool carry, Pass1 *pass1,
                  bool keep_unchanged = false /* carry mode, second chunk on: a read without a hit in this chunk keeps the comment the chunk before left */,
                  bool bam_only = false /* the batch's text has been printed from the device's reduced list (output_reduced): only the BAM records of this chunk are made here */)
{
    const uint32_t n_pairs = bt.n_reads / 2;
    double t0 = now_s();
    prepare_lists(bt, b, threads);
    st.t_sort += now_s() - t0; t0 = now_s();
    if (bam_only) { carry = true; pass1 = nullptr; }        // (carry: the counters below belong to the pass that printed the text)
    else st.dp += b.dp_instances;
    if (!carry) st.pairs += n_pairs;
    const auto &hits = b.hits;
    auto best_of = [](const std::vector<PairAln> &v) {           // first maximum of score_1 + score_2 (OutputDPResult.cpp:156-232)
        size_t best = 0;
        for (size_t i = 1; i < v.size(); i++) if (v[i].m[0].score + v[i].m[1].score > v[best].m[0].score + v[best].m[1].score) best = i;
        return best;
    };
    auto emit_pair = [&](Writer &w, Sink &sink, PairText &text, std::vector<PairAln> &res, uint32_t q, int file, int stage) {
        const Rec &r1 = bt.rec(2 * q), &r2 = bt.rec(2 * q + 1);
        const size_t best = best_of(res);
        if (p.mode) w.pair(r1, r2, bt.lens[2 * q], bt.lens[2 * q + 1], res);     // `-b` alone: no annotated FASTQ, and the BAM writer sees the results as they are (BGS-IO.cpp:2103)
        if (!bam) return;
        const mpbam::ReadView rd[2] = {view_of(r1, bt.lens[2 * q]), view_of(r2, bt.lens[2 * q + 1])};
        if (bam->w->wants_text()) { const Side *bs[2] = {&res[best].m[0], &res[best].m[1]}; text.fetch(ix, bs); }
        bam->w->pair(sink.bam[file], stage, rd, res, best, text.ref, bam->want_cov ? &sink.cov : nullptr);
    };
    // ---- pairs of the deep-DP round: groups of equal pair_id in the sorted hit list
    std::vector<size_t> grp;
    for (size_t i = 0; i < hits.size(); i++) if (i == 0 || hits[i].pair_id != hits[i - 1].pair_id) grp.push_back(i);
    grp.push_back(hits.size());
    for_chunks(grp.size() - 1, fm, bam, pass1, [&](size_t g0, size_t g1, Sink &sink) {
        Writer w(ix, p, sink.text, carry && !bam_only);
        w.mute = bam_only;
        PairText text;
        std::vector<PairAln> res;
        for (size_t g = g0; g < g1; g++) {
            res.clear();
            const uint32_t q = hits[grp[g]].pair_id;
            for (size_t j = grp[g]; j < grp[g + 1]; j++) {
                const mp_pair_hit &h = hits[j];
                PairAln r;
                r.m[0].pos = h.algnmt_1; r.m[1].pos = h.algnmt_2; r.m[0].score = h.score_1; r.m[1].score = h.score_2;
                r.m[0].strand = h.left_is_mate ? 2 : 1; r.m[1].strand = 3 - r.m[0].strand;
                if (bam) {
                    // the reference computes both edit distances with the right leg's read length (DV-DPfunctions.cpp:3466, :2999)
                    const int right_len = (int)bt.lens[2 * q + (h.left_is_mate ? 0 : 1)], right_side = h.left_is_mate ? 0 : 1;
                    int dis[2];
                    for (int k = 0; k < 2; k++) {
                        r.m[k].cigar = b.cig[2 * j + k].c_str();
                        r.m[k].editdist = mpbam::edit_dist(r.m[k].cigar, right_len, r.m[k].score, &dis[k]);
                    }
                    r.m[0].count = h.count_1; r.m[1].count = h.count_2;
                    const uint64_t d = h.algnmt_1 < h.algnmt_2 ? h.algnmt_2 - h.algnmt_1 : h.algnmt_1 - h.algnmt_2;
                    r.insert = (long long)d + right_len + dis[right_side];                              // :3520-3525
                }
                res.push_back(r);
            }
            emit_pair(w, sink, text, res, q, mpbam::Writer::FILE_DPOUT, 1);
        }
        w.finish();
        sink.pass1.add(w.pass1);
    });
    if (!carry) st.deep += grp.size() - 1;
    st.t_emit_deep += now_s() - t0; t0 = now_s();
    // ---- rescued pairs
    const auto &rows = b.rows;
    grp.clear();
    for (size_t i = 0; i < rows.size(); i++) if (i == 0 || rows[i].pair != rows[i - 1].pair) grp.push_back(i);
    grp.push_back(rows.size());
    for_chunks(grp.size() - 1, fm, bam, pass1, [&](size_t g0, size_t g1, Sink &sink) {
        Writer w(ix, p, sink.text, carry && !bam_only);
        w.mute = bam_only;
        PairText text;
        std::vector<PairAln> res;
        for (size_t g = g0; g < g1; g++) {
            res.clear();
            const uint32_t q = rows[grp[g]].pair;
            for (size_t j = grp[g]; j < grp[g + 1]; j++) {
                const mp_rescue_hit &h = b.resc[rows[j].src];
                const int al = (int)(h.aligned_read & 1u), un = 1 - al;
                PairAln r;
                r.m[al].pos = h.aligned_pos; r.m[al].score = h.aligned_score; r.m[al].strand = (int)h.aligned_strand;
                r.m[un].pos = h.dp_pos; r.m[un].score = h.dp_score; r.m[un].strand = (int)h.dp_strand;
                if (bam) {
                    int dis = 0;
                    r.m[al].cigar = b.rcig[2 * j].c_str(); r.m[un].cigar = b.rcig[2 * j + 1].c_str();
                    r.m[al].editdist = mpbam::edit_dist(r.m[al].cigar, (int)bt.lens[h.aligned_read], h.aligned_score, nullptr);
                    r.m[un].editdist = mpbam::edit_dist(r.m[un].cigar, (int)bt.lens[h.aligned_read ^ 1u], h.dp_score, &dis);
                    r.m[al].count = (int)h.aligned_count; r.m[un].count = (int)h.dp_count;
                    // DV-DPfunctions.cpp:1553-1565
                    if (h.dp_pos < h.aligned_pos) r.insert = (long long)(h.aligned_pos - h.dp_pos) + bt.lens[h.aligned_read];
                    else r.insert = (long long)(h.dp_pos - h.aligned_pos) + bt.lens[h.aligned_read ^ 1u] + dis;
                }
                res.push_back(r);
            }
            emit_pair(w, sink, text, res, q, mpbam::Writer::FILE_GOUT, bam ? bam->stage_rescue : 0);
        }
        w.finish();
        sink.pass1.add(w.pass1);
    });
    if (!carry) st.rescued += grp.size() - 1;
    // ---- the unpaired rest: per pair where its single-end hits (and their CIGARs) start
    const auto &singles = b.singles;
    struct Rest { uint32_t q; size_t si, ci; };
    std::vector<Rest> rest;
    {
        size_t si = 0, ci = 0;
        for (uint32_t q = 0; q < n_pairs; q++) {
            size_t e = si;
            while (e < singles.size() && (singles[e].read_id >> 1) == q) e++;
            if (!b.done[q]) { rest.push_back(Rest{q, si, ci}); ci += e - si; }
            si = e;
        }
    }
    for_chunks(rest.size(), fm, bam, pass1, [&](size_t i0, size_t i1, Sink &sink) {
        Writer w(ix, p, sink.text, carry && !bam_only);
        w.mute = bam_only;
        std::vector<Side> lst[2];
        std::vector<mpbam::Writer::RefText> ref_of[2];
        std::vector<uint8_t> tbuf[2];
        for (size_t i = i0; i < i1; i++) {
            const uint32_t q = rest[i].q;
            size_t si = rest[i].si, ci = rest[i].ci;
            for (int k = 0; k < 2; k++) {
                lst[k].clear();
                const uint32_t r = 2 * q + (uint32_t)k;
                for (; si < singles.size() && singles[si].read_id == r; si++) {
                    const mp_single_hit &h = singles[si];
                    Side s;
                    s.pos = h.algnmt; s.score = h.score; s.strand = (int)h.strand; s.count = (int)h.count;
                    if (bam) { s.cigar = b.scig[ci++].c_str(); s.editdist = mpbam::edit_dist(s.cigar, (int)bt.lens[r], h.score, nullptr); }
                    lst[k].push_back(s);
                }
                // Resident chunks: the comment this run's previous chunk wrote is a fixed point of the merge when the chunk adds nothing (its
                // entries already passed the threshold of the best score it carries), so it is left alone -- with many chunks most reads hit few
                if (keep_unchanged && bt.rec(r).has_comment && (lst[k].empty() || p.mode == 2)) continue;
                if (p.mode) w.single(bt.rec(r), bt.lens[r], lst[k]);
            }
            if (!bam) continue;
            const mpbam::ReadView rd[2] = {view_of(bt.rec(2 * q), bt.lens[2 * q]), view_of(bt.rec(2 * q + 1), bt.lens[2 * q + 1])};
            const mpbam::Writer::RefText *rp[2] = {nullptr, nullptr};
            if (bam->w->wants_text())
                for (int k = 0; k < 2; k++) {
                    const size_t n = lst[k].size();
                    ref_of[k].assign(n, mpbam::Writer::RefText());
                    if (!n) continue;
                    std::vector<uint64_t> pos(n); std::vector<uint32_t> len(n); uint32_t stride = 16;
                    for (size_t x = 0; x < n; x++) {
                        const bool ok = lst[k][x].pos != NOT_ALIGNED;
                        pos[x] = ok ? lst[k][x].pos : 0; len[x] = ok ? mpbam::Writer::ref_span(lst[k][x].cigar) + 8 : 0; stride = std::max(stride, len[x]);
                    }
                    tbuf[k].assign(n * (size_t)stride, 0);
                    die(mp_index_fetch_text(ix, pos.data(), len.data(), (uint32_t)n, tbuf[k].data(), stride), "mp_index_fetch_text");
                    for (size_t x = 0; x < n; x++) { ref_of[k][x].pos = pos[x]; ref_of[k][x].codes = tbuf[k].data() + x * (size_t)stride; ref_of[k][x].len = len[x]; }
                    rp[k] = ref_of[k].data();
                }
            bam->w->unpaired(sink.bam[mpbam::Writer::FILE_UNPAIR], bam->stage_unpaired, rd, lst, rp);
        }
        w.finish();
        sink.pass1.add(w.pass1);
    });
    if (!carry) st.unpaired += rest.size();
    st.t_emit_rest += now_s() - t0;
}

// ---- the same two stages when nothing but the annotated FASTQ (or LSAM) is wanted: the device reduces the hit lists of every chunk to
// one (sequence, best score) entry per mate, chunk and reference, and the printer makes one pass over the batch whatever the number of
// chunks.  (-b needs the alignments themselves: it keeps the hit-list path above.)
mp_align_params align_params_of(const Params &p, bool bam)
{
    mp_align_params ap;
    ap.insert_low = p.insert_low; ap.insert_high = p.insert_high; ap.clip_left = p.clip_left; ap.clip_right = p.clip_right; ap.mmp = p.mmp;
    ap.want_tie_counts = bam ? 1 : 0; ap.reserved = 0;          // num_sameScore only feeds the BAM MAPQ / X0 / X1
    return ap;
}

void gpu_stage_reduced(mp_ctx *ctx, const Params &p, Batch &bt, uint32_t chunk, bool last, Stats &st)
{
    double t0 = now_s();
    const mp_align_params ap = align_params_of(p, false);
    uint64_t nh = 0;
    mp_round_stats rs;
    die(mp_deep_dp_round(ctx, &ap, &nh, &rs), "mp_deep_dp_round");
    st.t_deep += now_s() - t0; t0 = now_s();
    mp_rescue_stats rst{};
    die(mp_rescue_round(ctx, &ap, p.skip_default_dp ? 1 : 0, &rst), "mp_rescue_round");
    st.t_rescue += now_s() - t0; t0 = now_s();
    mp_reduce_stats red{};
    if (last) bt.red_class.resize(bt.n_reads / 2 + 1);
    die(mp_reduce_chunk(ctx, p.mode, chunk, &red, last ? bt.red_class.data() : nullptr), "mp_reduce_chunk");
    bt.red_dp_instances += 2 * rs.n_candidates + rst.n_single_tasks + rst.n_rescue_tasks;
    if (last) bt.red_last = red;
    if (getenv("MP_TIMING"))
        fprintf(stderr, "[Main] batch %llu chunk %u: %llu seed positions, %llu candidates, %llu pair hits, %llu + %llu + %llu pairs deep / rescued / left, %llu entries (insert %d..%d)\n",
                (unsigned long long)bt.seq, chunk, (unsigned long long)rs.seed.n_seedpos, (unsigned long long)rs.n_candidates, (unsigned long long)nh,
                (unsigned long long)red.n_deep_pairs, (unsigned long long)red.n_rescued_pairs, (unsigned long long)red.n_unpaired_pairs, (unsigned long long)red.n_entries, ap.insert_low, ap.insert_high);
    st.t_sort += now_s() - t0;
}

// The two passes of the -top rule across resident chunks (mp_chunk_stash, kernels_floor.hip): pass 1 = the deep-DP round of a chunk and its pair hits into
// the reduction (which raises the best score known per read); pass 2 = the chunk's rescue round, without the pairs that cannot put anything into a header.
// false: the stash is out of memory -- the chunk's deep-DP round is done and reduced, its rescue round has to follow at once (finish_unstashed)
bool gpu_stage_pass1(mp_ctx *ctx, const Params &p, Batch &bt, uint32_t chunk, Stats &st)
{
    double t0 = now_s();
    const mp_align_params ap = align_params_of(p, false);
    uint64_t nh = 0;
    mp_round_stats rs;
    die(mp_deep_dp_round(ctx, &ap, &nh, &rs), "mp_deep_dp_round");
    st.t_deep += now_s() - t0; t0 = now_s();
    mp_reduce_stats red{};
    die(mp_reduce_chunk_part(ctx, p.mode, chunk, 1, &red, nullptr), "mp_reduce_chunk_part");
    bt.red_dp_instances += 2 * rs.n_candidates;
    const int rc = mp_chunk_stash(ctx, chunk);
    if (rc != 0 && rc != MP_ENOMEM) die(rc, "mp_chunk_stash");
    st.t_sort += now_s() - t0;
    return rc == 0;
}

void finish_unstashed(mp_ctx *ctx, const Params &p, Batch &bt, uint32_t chunk, Stats &st)
{
    double t0 = now_s();
    const mp_align_params ap = align_params_of(p, false);
    mp_rescue_stats rst{};
    die(mp_rescue_round(ctx, &ap, p.skip_default_dp ? 1 : 0, &rst), "mp_rescue_round");
    st.t_rescue += now_s() - t0; t0 = now_s();
    mp_reduce_stats red{};
    die(mp_reduce_chunk_part(ctx, p.mode, chunk, 2, &red, nullptr), "mp_reduce_chunk_part");
    bt.red_dp_instances += rst.n_single_tasks + rst.n_rescue_tasks;
    st.t_sort += now_s() - t0;
}

void gpu_stage_pass2(mp_ctx *ctx, const Params &p, Batch &bt, uint32_t chunk, int top_percent, Stats &st)
{
    double t0 = now_s();
    const mp_align_params ap = align_params_of(p, false);
    die(mp_chunk_unstash(ctx, chunk), "mp_chunk_unstash");
    die(mp_ctx_set_score_floor(ctx, top_percent), "mp_ctx_set_score_floor");
    mp_rescue_stats rst{};
    die(mp_rescue_round(ctx, &ap, p.skip_default_dp ? 1 : 0, &rst), "mp_rescue_round");
    die(mp_ctx_set_score_floor(ctx, 0), "mp_ctx_set_score_floor");
    st.t_rescue += now_s() - t0; t0 = now_s();
    mp_reduce_stats red{};
    die(mp_reduce_chunk_part(ctx, p.mode, chunk, 2, &red, nullptr), "mp_reduce_chunk_part");
    bt.red_dp_instances += rst.n_single_tasks + rst.n_rescue_tasks;
    bt.red_floor_pairs += rst.n_floor_pairs;
    bt.red_floor_mate_pairs += rst.n_floor_mate_pairs;
    if (getenv("MP_TIMING"))
        fprintf(stderr, "[Main] batch %llu chunk %u, rescue pass: %llu failed pairs, %llu left out by the -top rule (%llu instances), %llu instances run, %llu + %llu hits\n",
                (unsigned long long)bt.seq, chunk, (unsigned long long)rst.n_failed_pairs, (unsigned long long)rst.n_floor_pairs, (unsigned long long)rst.n_floor_tasks,
                (unsigned long long)rst.n_single_tasks, (unsigned long long)rst.n_single_hits, (unsigned long long)rst.n_rescue_hits);
    st.t_sort += now_s() - t0;
}

void finish_reduced(mp_ctx *ctx, Batch &bt, Stats &st)
{
    const double t0 = now_s();
    const uint64_t *keys = nullptr;
    uint64_t n = 0;
    die(mp_reduce_finish(ctx, &keys, &n), "mp_reduce_finish");
    bt.red.assign(keys, keys + n);
    // where each read's keys start (the keys are sorted by read)
    bt.red_off.assign((size_t)bt.n_reads + 1, 0u);
    for (uint64_t i = 0; i < n; i++) bt.red_off[(size_t)(keys[i] >> 42) + 1]++;
    for (size_t r = 0; r < bt.n_reads; r++) bt.red_off[r + 1] += bt.red_off[r];
    bt.reduced = true;
    st.t_sort += now_s() - t0;
}

void output_reduced(mp_index *const *chunks, uint32_t n_chunks, const Params &p, Batch &bt, Formatters &fm, Stats &st, Pass1 *pass1)
{
    const double t0 = now_s();
    const uint32_t n_pairs = bt.n_reads / 2;
    st.pairs += n_pairs; st.dp += bt.red_dp_instances; st.floor_pairs += bt.red_floor_pairs; st.floor_mate_pairs += bt.red_floor_mate_pairs;
    st.deep += bt.red_last.n_deep_pairs; st.rescued += bt.red_last.n_rescued_pairs; st.unpaired += bt.red_last.n_unpaired_pairs;
    // the batch in the reference's order (alignment.cpp:105-330): the pairs of the deep-DP round, the rescued pairs, the rest -- as the last chunk saw them
    std::vector<uint32_t> order[3];
    for (uint32_t q = 0; q < n_pairs; q++) order[bt.red_class[q] < 3 ? bt.red_class[q] : 2].push_back(q);
    for (int sec = 0; sec < 3; sec++) {
        const std::vector<uint32_t> &ord = order[sec];
        for_chunks(ord.size(), fm, nullptr, pass1, [&](size_t i0, size_t i1, Sink &sink) {
            Writer w(chunks[n_chunks - 1], p, sink.text, false);
            for (size_t i = i0; i < i1; i++)
                for (uint32_t k = 0; k < 2; k++) {
                    const uint32_t r = 2 * ord[i] + k;
                    w.reduced(bt.rec(r), bt.red.data() + bt.red_off[r], bt.red_off[r + 1] - bt.red_off[r], chunks, n_chunks);
                }
            w.finish();
            sink.pass1.add(w.pass1);
        });
    }
    st.t_emit_deep += now_s() - t0;
}

// hand-over of batches between the stages
struct BatchQueue {
    std::mutex mu; std::condition_variable cv;
    std::vector<Batch *> q;
    bool closed = false;
    void push(Batch *b) { { std::lock_guard<std::mutex> lk(mu); q.push_back(b); } cv.notify_all(); }
    void close() { { std::lock_guard<std::mutex> lk(mu); closed = true; } cv.notify_all(); }
    Batch *pop()                                                  // nullptr once closed and drained
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        Batch *b = q.front(); q.erase(q.begin());
        return b;
    }
    // the batch with sequence number `seq`, whenever the workers get it done (they finish out of order)
    Batch *pop_seq(uint64_t seq)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            for (size_t i = 0; i < q.size(); i++) if (q[i]->seq == seq) { Batch *b = q[i]; q.erase(q.begin() + (long)i); return b; }
            if (closed) return nullptr;
            cv.wait(lk);
        }
    }
};

std::vector<std::string> split_commas(const std::string &v)
{
    std::vector<std::string> out;
    size_t a = 0;
    while (a <= v.size()) { size_t b = v.find(',', a); if (b == std::string::npos) b = v.size(); if (b > a) out.push_back(v.substr(a, b - a)); a = b + 1; }
    return out;
}

// One GPU of the node: every index chunk resident in its HBM (a replica per GPU), and the contexts of its workers.
struct Device {
    int id = 0;
    std::vector<mp_index *> chunks;
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 5 || strcmp(argv[1], "pair") != 0) {
        fprintf(stderr, "usage: %s pair <indexPrefix[,indexPrefix...]> <r1.fq[.gz]> <r2.fq[.gz]> -o <outPrefix> -C <ini> -L <maxReadLen> [-T n] [-u maxInsert] [-v minInsert] -F|-P [-nc] [-top N]\n"
                        "          [-b [-p] [-D rg -A sample [-R opt]]]   BAM side files; implemented for output option -h 2 (the default) with BWALikeScore=1 (both shipped ini files)\n"
                        "          [-c gpu[,gpu...]]                      GPUs of this node to use (default 0, or $MP_GPUS): pair blocks are spread over them, output stays in input order\n"
                        "       several comma-separated index prefixes = the chunks of one database (the NT loop of runMegaPath.sh:191-227 in one process): all are kept in HBM,\n"
                        "       every batch is aligned against each, and the per-reference hits are merged as the chained runs would merge them through the SCORE: comments\n", argv[0]);
        return 1;
    }
    Params p;
    std::string ini_path;
    int threads = 4, out_type = 2;                                        // IniParam.cpp:627-648
    std::vector<int> gpu_ids;
    const char *fq1 = argv[3], *fq2 = argv[4];
    const std::vector<std::string> prefixes = split_commas(argv[2]);
    if (prefixes.empty()) { fprintf(stderr, "[soap4-mi355x] no index prefix given\n"); return 1; }
    std::string out_prefix = fq1;                                         // IniParam.cpp:633
    mpbam::Options bopt;
    std::string tax_files[3], pass1_path;
    double spike_k = 60, spike_overlap = 0.5;                  // SPIKE_STDEV / SPIKE_OVERLAP (runMegaPath.sh:52-53)
    bool want_spike = false;
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
        else if (a == "-c") {
            // -c <GPU device id> (IniParam.cpp:822-838); a comma-separated list names several GPUs of the node
            for (const std::string &g : split_commas(val())) { const int id = atoi(g.c_str()); if (id < 0 || !isdigit((unsigned char)g[0])) { fprintf(stderr, "The GPU device ID should not be less than 0\n"); return 1; } gpu_ids.push_back(id); }
        }
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
        else if (a == "-lsam") p.lsam = 1;                     // not in the reference: stdout = LSAM, saves the fastq2lsam pipe stage
        else if (a == "-lsam-noseq") p.lsam = 2;
        else if (a == "-lsamid") { tax_files[0] = val(); tax_files[1] = val(); tax_files[2] = val(); }   // not in the reference: stdout = LSAM.id (fastq2lsam | taxLookupAcc in-process)
        else if (a == "-spike") { spike_k = atof(val()); spike_overlap = atof(val()); want_spike = true; }         // not in the reference: the spike filter of runMegaPath.sh:211-218 in-process
        else if (a == "-pass1") pass1_path = val();                                                        // not in the reference: counters of reassign's pass 1 over those lines
        else if (a == "-pass1-t") p.pass1_t = atof(val());
    }
    if (p.insert_low > p.insert_high) { fprintf(stderr, "The minimum value of insert size should not be greater than the maximum value of insert size.\n"); return 1; }
    // IniParam.cpp:918-940: -R needs -D and -A; -A and -D come together
    if ((have_r && (!have_a || !have_d)) || (have_a != have_d)) { fprintf(stderr, "Options '-D' and '-A' have to be specified together ('-R' needs both).\n"); return 1; }
    // Output modes that work in the reference: -F / -P (with or without -b) and -b alone (BAM side files only, nothing on stdout).
    // With none of them the reference dies with SIGSEGV after seeding (this is what runSOAP3dp.pl:44-50 would
    // start): here that invocation is refused before any work is done.
    if (p.mode == 0 && !p.want_bam) { fprintf(stderr, "[soap4-mi355x] no output mode: give -F or -P (annotated FASTQ on stdout) and/or -b (BAM side files); the reference crashes on this invocation\n"); return 1; }
    if (p.mode == 0 && (p.lsam || !tax_files[0].empty() || !pass1_path.empty())) { fprintf(stderr, "[soap4-mi355x] -lsam / -lsamid / -pass1 rewrite the stdout of -F / -P: give one of them\n"); return 1; }
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
    if (gpu_ids.empty()) {
        if (const char *e = getenv("MP_GPUS")) for (const std::string &g : split_commas(e)) gpu_ids.push_back(atoi(g.c_str()));
        else if (const char *e1 = getenv("MP_DEVICE")) gpu_ids.push_back(atoi(e1));
        else if (const char *e2 = getenv("LOCAL_RANK")) gpu_ids.push_back(atoi(e2));
        else gpu_ids.push_back(0);
    }
    // Test aid for the host plumbing (reader -> workers -> ordered output) on a machine without a GPU: no index is opened, nothing is
    // aligned, every pair is printed as unaligned.  Never a substitute for the device path: it says so on stderr.
    const bool plumbing_only = getenv("MP_PLUMBING_TEST") != nullptr;
    if (plumbing_only) fprintf(stderr, "[soap4-mi355x] MP_PLUMBING_TEST: NO ALIGNMENT IS DONE -- host plumbing test only, every pair is reported unaligned\n");
    int workers_per_gpu = 3;                   // three contexts per GPU: the memory-bound seeding of one batch overlaps with the VALU-bound DP of the others (bench.py: +4 % over two)
    fill_char_map();

    const size_t n_chunks = prefixes.size();
    if (n_chunks >= 2) workers_per_gpu = 4;    // several resident chunks (the NT stage): a batch stays on the device for seconds, four side by side measured +3 % over three (bench.py --contexts)
    if (const char *e = getenv("MP_WORKERS_PER_GPU")) workers_per_gpu = std::max(1, atoi(e));
    std::vector<Device> devices(gpu_ids.size());
    if (!plumbing_only && n_chunks >= 2) {
        // A database of several chunks gets ONE layout.  Left to itself the library decides index by index from the HBM that is free when the index is opened
        // (INTEGRATION.md, section 4): the first chunks of the NT database would take every optional table and the rest would not fit.  So: unless the chunks
        // with all their optional tables (~38 bytes per base) fit into 60 % of the card, every chunk is opened in the lean layout (0.92 B/base + the 13-mer
        // table), and with the 13-mer tables as 8-byte entries that hold a seed's first extension step (MP_LKT14, +256 MiB per chunk) when that still leaves
        // 36 GiB for the batches.  Switches the user has set stay as they are.
        uint64_t bases = 0, mfree = 0, mtotal = 0;
        for (const std::string &pre : prefixes) { struct stat sb; if (stat((pre + ".pac").c_str(), &sb) == 0) bases += 4ull * (uint64_t)sb.st_size; }
        if (const char *e = getenv("MP_LAYOUT_HBM_BYTES")) mtotal = strtoull(e, nullptr, 10);           // test aid: decide as if the card had this much
        else if (mp_device_memory(gpu_ids[0], &mfree, &mtotal) != 0) mtotal = 0;
        if (bases && mtotal) {
            const double lean = 0.92 * (double)bases + (double)n_chunks * (double)(1ull << 28), rich = 38.0 * (double)bases;
            if (rich > 0.6 * (double)mtotal) {
                const bool lkt14 = lean + (double)n_chunks * (double)(1ull << 28) + (double)(36ull << 30) <= (double)mtotal;
                for (const char *k : {"MP_KMER_FILTER", "MP_KMER15", "MP_LKT16", "MP_SA_DENSE", "MP_LKT16U"}) setenv(k, "0", 0);
                if (lkt14) setenv("MP_LKT14", "1", 0);
                fprintf(stderr, "[Main] %zu index chunks, %.1f Gbp: lean layout for all of them (%.1f GB)%s\n", n_chunks, bases / 1e9,
                        (lean + (lkt14 ? (double)n_chunks * (double)(1ull << 28) : 0.0)) / 1e9, lkt14 ? ", 13-mer tables with the first extension step in them (MP_LKT14)" : "");
            }
        }
    }
    if (!plumbing_only) {
        // every GPU loads its own replica of every chunk, the GPUs side by side
        std::vector<std::thread> loaders;
        for (size_t d = 0; d < devices.size(); d++) {
            devices[d].id = gpu_ids[d];
            devices[d].chunks.assign(n_chunks, nullptr);
            loaders.emplace_back([&, d]() {
                for (size_t c = 0; c < n_chunks; c++) {
                    die(mp_index_open(prefixes[c].c_str(), devices[d].id, &devices[d].chunks[c]), "mp_index_open");
                    mp_index_info info;
                    mp_index_get_info(devices[d].chunks[c], &info);
                    fprintf(stderr, "[Main] GPU %d: index %s in HBM: %.2f GB, %u sequences\n", devices[d].id, prefixes[c].c_str(), info.device_bytes / 1e9, info.num_seq);
                }
            });
        }
        for (auto &t : loaders) t.join();
    } else
        for (size_t d = 0; d < devices.size(); d++) { devices[d].id = gpu_ids[d]; devices[d].chunks.assign(n_chunks, nullptr); }

    Taxonomy tax;
    if (!tax_files[0].empty()) {
        if (plumbing_only) { fprintf(stderr, "[soap4-mi355x] -lsamid needs the index (sequence names)\n"); return 1; }
        // accessions that can turn up: the names of the resident sequences -- and, when header comments of earlier runs are merged
        // (no -nc), anything those list: then every row of the table is kept
        std::unordered_set<std::string> wanted;
        if (p.ignore_comments)
            for (mp_index *ix : devices[0].chunks) {
                mp_index_info info;
                mp_index_get_info(ix, &info);
                for (uint32_t q = 1; q <= info.num_seq; q++) wanted.insert(Taxonomy::remove_version(Taxonomy::correct_acc(mp_index_seq_name(ix, q))));
            }
        const double t0 = now_s();
        if (!tax.load(tax_files[0], tax_files[1], tax_files[2], wanted)) { fprintf(stderr, "[soap4-mi355x] cannot read the taxonomy files given to -lsamid\n"); return 1; }
        fprintf(stderr, "[Main] taxonomy: %zu accessions kept, loaded in %.1f s\n", tax.acc2tid.size(), now_s() - t0);
        p.tax = &tax;
        if (p.lsam == 0) p.lsam = 1;
    } else if (!pass1_path.empty()) { fprintf(stderr, "[soap4-mi355x] -pass1 counts taxids: it needs -lsamid\n"); return 1; }
    Pass1 pass1_total;
    Source ra, rb;
    if (!ra.open(fq1, threads) || !rb.open(fq2, threads)) { fprintf(stderr, "[soap4-mi355x] cannot open read files\n"); return 1; }
    BamState bam_state, *bam = nullptr;
    if (p.want_bam && !plumbing_only) {
        // as in the chained runs of runMegaPath.sh:191-227 -- every run writes the same -o prefix -- the files that remain are the last chunk's
        bam_state.w.reset(new mpbam::Writer(devices[0].chunks[n_chunks - 1], bopt));
        if (!bam_state.w->open(out_prefix, "@PG\tID:soap4-mi355x\tPN:soap4-mi355x\tVN:r2\n")) return 1;
        bam_state.stage_unpaired = p.skip_default_dp ? 2 : 3;
        bam_state.want_cov = want_spike;
        bam = &bam_state;
    }
    if (want_spike && !bam) { fprintf(stderr, "[soap4-mi355x] -spike works on the alignments of the BAM side files: it needs -b\n"); return 1; }
    // Stages: the reader threads parse and pack batches (the reference's IO thread, aio_thread.cpp:247); every worker of every
    // GPU takes the next batch -- a contiguous block of pairs --, aligns it against all resident chunks and hands it on; the
    // printer writes the batches in input order.  Batch objects are recycled.
    const size_t n_workers = devices.size() * (size_t)workers_per_gpu;
    std::vector<Batch> pool(n_workers + 2);
    for (Batch &b : pool) b.ch.resize(n_chunks);
    BatchQueue free_q, filled_q, aligned_q;
    for (Batch &b : pool) free_q.push(&b);
    const double t_start = now_s();
    Stats st_reader, st_printer;
    std::vector<Stats> st_worker(n_workers);
    std::thread reader([&]() {
        uint64_t seq = 0;
        for (;;) {
            Batch *b = free_q.pop();
            if (!b) break;
            const double tr = now_s();
            const bool got = fill_batch(ra, rb, p, *b, threads);
            st_reader.t_read += now_s() - tr;
            if (got && seq == 0) {
                // SOAP4.cpp:465-474 / QueryParser.cpp:2253-2277: the minimum insert size is at least the read lengths detected in the
                // first batch.  Settled here, before any worker sees a batch (the queue's mutex orders the write before their reads).
                const uint32_t n_pairs = b->n_reads / 2;
                uint32_t d1 = 0, d2 = 0;
                for (uint32_t q = 0; q < n_pairs && q < 999999; q++) { d1 = std::max(d1, b->lens[2 * q]); d2 = std::max(d2, b->lens[2 * q + 1]); }
                p.insert_low = std::max(p.insert_low, (int)std::max(d1, d2));
            }
            if (got) { b->seq = seq++; filled_q.push(b); }
            if (!got || b->last) break;
        }
        filled_q.close();
    });
    // Without -b only the per-reference best scores are needed: they are reduced on the device over all chunks (mp_reduce_*), the printer
    // makes one pass per batch.  MP_HOST_MERGE=1 (and -b, and databases beyond the key's limits) keeps the hit lists and the per-chunk merge
    // of the header comments on the host: the reference's chain, step by step.
    // With -b the reduction still does the comments: the chained runs all write the same -o prefix (runMegaPath.sh:199), so the BAM files that
    // remain are the LAST chunk's -- only that chunk's hit lists travel to the host (MP_BAM_HOST_MERGE=1: the round-3 path, every chunk's).
    bool reduce_on_device = !plumbing_only && (bam == nullptr || !getenv("MP_BAM_HOST_MERGE")) && p.mode != 0 && !getenv("MP_HOST_MERGE") && n_chunks <= 1024;
    if (reduce_on_device)
        for (mp_index *ix : devices[0].chunks) { mp_index_info info; mp_index_get_info(ix, &info); if (info.num_seq >= (1u << 22)) reduce_on_device = false; }
    if (p.max_read_length > 512 || p.batch_pairs > (1u << 21)) reduce_on_device = false;       // scores must stay below 1024, read ids below 2^22
    // The -top rule across resident chunks (kernels_floor.hip): on whenever the reduction is and there is more than one chunk; MP_TOP_FLOOR=0 switches it off
    // (one pass over the chunks, every rescue round does all its work: the test of the rule)
    const int top_percent = (int)(p.top * 100.0 + 0.5);
    const bool top_floor = reduce_on_device && n_chunks >= 2 && top_percent >= 1 && top_percent <= 100 && !(getenv("MP_TOP_FLOOR") && atoi(getenv("MP_TOP_FLOOR")) == 0);
    std::atomic<bool> stash_warned{false};
    if (top_floor && !getenv("MP_BATCH_PAIRS")) {
        // the rule keeps the seed positions of every chunk but the last between its two passes: ~0.6 KB per pair and chunk with 17-base seeds (67 located
        // hits per pair and 2.95 Gbp chunk, 8 bytes each: stash_pack_kernel), for every worker's batch.  Batches are sized so that this fits into half of what the indexes left free.
        uint64_t mfree = 0, mtotal = 0;
        if (mp_device_memory(devices[0].id, &mfree, &mtotal) == 0) {
            const double per_pair = ((getenv("MP_STASH_PLAIN") && atoi(getenv("MP_STASH_PLAIN"))) ? 1200.0 : 620.0) * (double)(n_chunks - 1);
            const size_t fit = (size_t)(0.5 * (double)mfree / (per_pair * (double)workers_per_gpu));
            const size_t want = std::max<size_t>(16384, fit / 4096 * 4096);
            if (want < p.batch_pairs) {
                p.batch_pairs = want;
                fprintf(stderr, "[Main] %zu index chunks resident, %.1f GB of HBM free: batches of %zu pairs (the -top rule keeps every chunk's seed positions between its two passes)\n",
                        n_chunks, mfree / 1e9, p.batch_pairs);
            }
        }
    }
    const bool carry_rewrite = getenv("MP_CARRY_REWRITE") != nullptr;     // test switch: merge every comment again in every chunk, also where the chunk adds nothing     // test switch: merge every comment again in every chunk, also where the chunk adds nothing
    std::thread printer([&]() {
        Formatters fm(getenv("MP_PRINTER_THREADS") ? std::max(1, atoi(getenv("MP_PRINTER_THREADS"))) : std::max(1, threads / 2));
        for (uint64_t seq = 0;; seq++) {
            Batch *b = aligned_q.pop_seq(seq);
            if (!b) break;
            if (b->reduced) {
                output_reduced(devices[0].chunks.data(), (uint32_t)n_chunks, p, *b, fm, st_printer, p.tax ? &pass1_total : nullptr);
                // -b: the BAM records of the last chunk, from its hit lists, in the same three sections
                if (bam) output_stage(devices[0].chunks[n_chunks - 1], p, *b, b->ch[n_chunks - 1], bam, threads, fm, st_printer, false, nullptr, false, true);
                b->reduced = false;
                free_q.push(b);
                continue;
            }
            // the chunks in database order: each one's per-reference hits join the header comments the earlier ones left
            for (size_t c = 0; c < n_chunks; c++) {
                const bool last = c + 1 == n_chunks;
                output_stage(devices[0].chunks[c], p, *b, b->ch[c], last ? bam : nullptr, threads, fm, st_printer, !last, p.tax ? &pass1_total : nullptr, !last && c > 0 && !carry_rewrite);
            }
            free_q.push(b);
        }
    });
    std::atomic<uint64_t> pairs_seen{0};
    std::vector<std::thread> workers;
    for (size_t w = 0; w < n_workers; w++) {
        workers.emplace_back([&, w]() {
            Device &dev = devices[w / (size_t)workers_per_gpu];
            Stats &st = st_worker[w];
            mp_ctx *ctx = nullptr;
            if (!plumbing_only) die(mp_ctx_create(dev.chunks[0], &ctx), "mp_ctx_create");
            if (ctx && reduce_on_device && !bam) die(mp_ctx_set_hit_copies(ctx, 0), "mp_ctx_set_hit_copies");
            for (;;) {
                Batch *b = filled_q.pop();
                if (!b) break;
                const double tb = now_s();
                pairs_seen += b->n_reads / 2;
                if (!plumbing_only && reduce_on_device) {
                    upload_reads(ctx, *b, st);
                    die(mp_reduce_begin(ctx, (uint32_t)n_chunks), "mp_reduce_begin");
                    b->red_dp_instances = 0; b->red_floor_pairs = 0; b->red_floor_mate_pairs = 0;
                    auto whole_chunk = [&](size_t c) {
                        if (n_chunks > 1 || c > 0) die(mp_ctx_set_index(ctx, dev.chunks[c]), "mp_ctx_set_index");
                        const bool last = c + 1 == n_chunks;
                        if (bam) die(mp_ctx_set_hit_copies(ctx, last ? 1 : 0), "mp_ctx_set_hit_copies");
                        if (bam && last) gpu_stage(ctx, p, *b, b->ch[c], true, threads, st, (int)c);
                        else gpu_stage_reduced(ctx, p, *b, (uint32_t)c, last, st);
                    };
                    if (!top_floor) for (size_t c = 0; c < n_chunks; c++) whole_chunk(c);
                    else {
                        // two passes (the -top rule across chunks): the deep-DP round of every chunk first -- it fixes the best score each read is known
                        // to reach --, the last chunk as ever (its classes are the sections of the output, its lists feed -b), then the rescue rounds.
                        // Should the stash run out of memory at chunk c (a batch with far more seeds than the sizing assumed), that chunk and the ones
                        // after it are done in one go as without the rule; the chunks already stashed still get their second pass.
                        if (bam) die(mp_ctx_set_hit_copies(ctx, 0), "mp_ctx_set_hit_copies");
                        size_t n_stashed = 0;
                        for (size_t c = 0; c + 1 < n_chunks; c++) {
                            die(mp_ctx_set_index(ctx, dev.chunks[c]), "mp_ctx_set_index");
                            if (n_stashed == c && gpu_stage_pass1(ctx, p, *b, (uint32_t)c, st)) n_stashed = c + 1;
                            else {
                                if (n_stashed == c) { finish_unstashed(ctx, p, *b, (uint32_t)c, st); if (!stash_warned.exchange(true)) fprintf(stderr, "[soap4-mi355x] the -top rule's stash does not fit at chunk %zu of %zu (batch of %u pairs): the remaining chunks are aligned in one pass; a smaller MP_BATCH_PAIRS gives it room\n", c, n_chunks, b->n_reads / 2); }
                                else gpu_stage_reduced(ctx, p, *b, (uint32_t)c, false, st);
                            }
                        }
                        whole_chunk(n_chunks - 1);
                        if (bam) die(mp_ctx_set_hit_copies(ctx, 0), "mp_ctx_set_hit_copies");
                        for (size_t c = 0; c < n_stashed; c++) { die(mp_ctx_set_index(ctx, dev.chunks[c]), "mp_ctx_set_index"); gpu_stage_pass2(ctx, p, *b, (uint32_t)c, top_percent, st); }
                    }
                    finish_reduced(ctx, *b, st);
                } else if (!plumbing_only) {
                    upload_reads(ctx, *b, st);
                    for (size_t c = 0; c < n_chunks; c++) {
                        if (n_chunks > 1 || c > 0) die(mp_ctx_set_index(ctx, dev.chunks[c]), "mp_ctx_set_index");
                        gpu_stage(ctx, p, *b, b->ch[c], bam != nullptr && c + 1 == n_chunks, threads, st);
                    }
                } else {
                    for (auto &c : b->ch) c.done.assign(b->n_reads / 2, 0);
                    if (const char *e = getenv("MP_PLUMBING_JITTER_US")) usleep((useconds_t)((b->seq * 7919u + w * 104729u) % (uint64_t)std::max(1, atoi(e))));
                }
                if (getenv("MP_TIMING")) fprintf(stderr, "[Main] GPU %d: device stage of batch %llu (%u pairs): %.3f s\n", dev.id, (unsigned long long)b->seq, b->n_reads / 2, now_s() - tb);
                aligned_q.push(b);
            }
            if (ctx) mp_ctx_destroy(ctx);
        });
    }
    for (auto &t : workers) t.join();
    aligned_q.close();
    printer.join();
    free_q.close();
    reader.join();
    const double t_loop = now_s() - t_start;
    fflush(stdout);
    if (bam && !bam->w->close()) { fprintf(stderr, "[soap4-mi355x] error while writing the BAM files\n"); return 1; }
    if (want_spike) {
        // coverage of the last chunk's .dpout / .gout records (the files `samtools merge *[0-9]` picks up), spikes, reads to drop
        mp_index *ix = devices[0].chunks[n_chunks - 1];
        mp_coverage *cov = nullptr;
        die(mp_coverage_create(ix, &cov), "mp_coverage_create");
        const size_t n = bam->cov.size();
        std::vector<uint32_t> seq(n), span(n), covered(n);
        std::vector<uint64_t> start(n);
        for (size_t i = 0; i < n; i++) { seq[i] = bam->cov[i].seq_id; start[i] = bam->cov[i].start0; span[i] = bam->cov[i].span; }
        die(mp_coverage_add(cov, seq.data(), start.data(), span.data(), n), "mp_coverage_add");
        die(mp_coverage_finish(cov, spike_k), "mp_coverage_finish");
        uint64_t n_iv = 0;
        die(mp_coverage_spike_intervals(cov, nullptr, nullptr, nullptr, 0, &n_iv), "mp_coverage_spike_intervals");
        std::vector<uint32_t> iv_seq(n_iv + 1); std::vector<uint64_t> iv_a(n_iv + 1), iv_b(n_iv + 1);
        die(mp_coverage_spike_intervals(cov, iv_seq.data(), iv_a.data(), iv_b.data(), n_iv, &n_iv), "mp_coverage_spike_intervals");
        FILE *f = fopen((out_prefix + ".genomecovFilter").c_str(), "w");
        if (!f) { fprintf(stderr, "[soap4-mi355x] cannot write %s.genomecovFilter\n", out_prefix.c_str()); return 1; }
        for (uint64_t i = 0; i < n_iv; i++) fprintf(f, "%s\t%llu\t%llu\n", mp_index_seq_name(ix, iv_seq[i]), (unsigned long long)iv_a[i], (unsigned long long)iv_b[i]);
        fclose(f);
        die(mp_coverage_spike_bases(cov, seq.data(), start.data(), span.data(), n, covered.data()), "mp_coverage_spike_bases");
        f = fopen((out_prefix + ".readFilter").c_str(), "w");
        if (!f) { fprintf(stderr, "[soap4-mi355x] cannot write %s.readFilter\n", out_prefix.c_str()); return 1; }
        uint64_t n_drop = 0;
        for (size_t i = 0; i < n; i++) {
            // bedtools annotate prints the fraction with six decimals and awk compares that number with the threshold (runMegaPath.sh:218)
            char buf[32];
            snprintf(buf, sizeof buf, "%f", span[i] ? (double)covered[i] / (double)span[i] : 0.0);
            if (atof(buf) >= spike_overlap) { fprintf(f, "%s/%d\n", bam->cov[i].name.c_str(), (int)bam->cov[i].mate); n_drop++; }
        }
        fclose(f);
        fprintf(stderr, "[Main] spike filter: %zu alignments, %llu spike intervals (depth > mean + %g stdev), %llu alignments >= %g on spikes\n", n, (unsigned long long)n_iv, spike_k,
                (unsigned long long)n_drop, spike_overlap);
        mp_coverage_destroy(cov);
    }
    if (!pass1_path.empty() && !pass1_total.write(pass1_path, p.pass1_t)) { fprintf(stderr, "[soap4-mi355x] cannot write %s\n", pass1_path.c_str()); return 1; }
    // every stage kept its own counters: summed only now that the threads are gone
    Stats st = st_printer;
    st.t_read = st_reader.t_read;
    for (const Stats &w : st_worker) { st.t_upload += w.t_upload; st.t_deep += w.t_deep; st.t_rescue += w.t_rescue; st.t_sort += w.t_sort; st.t_cigar += w.t_cigar; }
    fprintf(stderr, "[Main] %llu pairs: %llu aligned by deep DP, %llu by mate rescue, %llu unpaired; %llu DP instances\n", (unsigned long long)st.pairs,
            (unsigned long long)st.deep, (unsigned long long)st.rescued, (unsigned long long)st.unpaired, (unsigned long long)st.dp);
    fprintf(stderr, "[Main] alignment loop %.2f s (%.0f pairs/s, reading and packing overlapped; %zu GPU(s) x %d worker(s), %zu index chunk(s)%s%s)\n", t_loop,
            st.pairs / std::max(t_loop, 1e-9), devices.size(), workers_per_gpu, n_chunks, reduce_on_device ? ", per-reference scores reduced on the device" : "",
            top_floor ? ", rescue rounds after the deep-DP rounds of all chunks (-top rule)" : "");
    if (top_floor) fprintf(stderr, "[Main] -top rule: %llu (pair, chunk) rescue rounds left out, %llu of them by the bound on the rescued mate\n", (unsigned long long)st.floor_pairs,
                           (unsigned long long)st.floor_mate_pairs);
    if (getenv("MP_TIMING"))
        fprintf(stderr, "[Main] seconds: reading and packing %.2f (own threads), upload %.2f, deep-DP round %.2f, sort %.2f, BAM CIGARs %.2f, output of deep pairs %.2f, rescue round %.2f, output of the rest %.2f\n",
                st.t_read, st.t_upload, st.t_deep, st.t_sort, st.t_cigar, st.t_emit_deep, st.t_rescue, st.t_emit_rest);
    for (Device &d : devices) for (mp_index *ix : d.chunks) if (ix) mp_index_close(ix);
    return 0;
}
