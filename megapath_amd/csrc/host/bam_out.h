// megapath_amd/csrc/host/bam_out.h -- product code: the BAM side files of `soap4 -b`.
//
//   <prefix>.dpout.1   pairs aligned by the deep-DP round          (DV-DPfunctions.cpp:3541-3609 -> samOutputDPFilePtr)
//   <prefix>.gout.N    pairs placed by mate rescue, N = 1..-T      (DV-DPfunctions.cpp:1625-1747 -> currSamOutputFilePtr[thread])
//   <prefix>.unpair    every remaining pair                        (OutputDPResult.cpp:284-404)
// Record content follows the reference's writers:
//   pairDeepDPOutputSAMAPI        BGS-IO.cpp:2093-2788     unproperlypairDPOutputSAMAPI  BGS-IO.cpp:1448-1915
//   initializeSAMAlgnmt[2]        BGS-IO.cpp:433-675       SAMOutputHeaderConstruct      SAM.cpp:83-132
//   BoundaryCheckDP               BGS-IO.cpp:219-386       convertToCigarStr             PE.cpp:381-445
//   getMisInfoForDP               PE.cpp:459-620           bwaLike*QualScore             BGS-IO.cpp:709-730, 826-876
// The container is written here (BGZF blocks + BAM records, SAM spec section 4) instead of through the bundled samtools.
// Only the reference's own output type ALL_BEST (-h 2, the default and what MegaPath uses) and BWALikeScore=1 scoring
// (both shipped ini files) are implemented; the driver refuses anything else when -b is given.
#ifndef MP_BAM_OUT_H
#define MP_BAM_OUT_H
#include <zlib.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/megapath_hip.h"

namespace mpbam {

const uint64_t NOT_ALIGNED = ~0ull;

// ------------------------------------------------------------------------------------------------ BGZF container
// Blocks are independent deflate streams, so a group of them is compressed by several threads at once and written in
// order (the reference compresses on its single output thread).
class Bgzf {
public:
    bool open(const std::string &path, int level, int threads)
    {
        f_ = fopen(path.c_str(), "wb");
        level_ = level; threads_ = threads < 1 ? 1 : threads;
        return f_ != nullptr;
    }
    void write(const void *data, size_t n)
    {
        const uint8_t *p = (const uint8_t *)data;
        raw_.insert(raw_.end(), p, p + n);
        if (raw_.size() >= kBlock * kGroup * (size_t)threads_) flush_blocks(false);
    }
    bool close()
    {
        if (!f_) return true;
        flush_blocks(true);
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool ok = fwrite(eof, 1, sizeof eof, f_) == sizeof eof && !failed_;
        ok = fclose(f_) == 0 && ok;
        f_ = nullptr;
        return ok;
    }
    ~Bgzf() { close(); }

private:
    static constexpr size_t kBlock = 0xff00, kGroup = 16, kOut = 0x10000 + 64;   // constexpr: std::min takes them by reference (ODR use)
    // one BGZF block: 18-byte header with the BC extra field, raw deflate data, CRC32 and input size
    static size_t deflate_block(const uint8_t *in, size_t n, uint8_t *out, int level)
    {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
        z.next_in = const_cast<uint8_t *>(in); z.avail_in = (uInt)n;
        z.next_out = out + 18; z.avail_out = (uInt)(kOut - 18 - 8);
        const int rc = deflate(&z, Z_FINISH);
        deflateEnd(&z);
        if (rc != Z_STREAM_END) return 0;                        // cannot happen: 0xff00 bytes always fit
        const size_t clen = z.total_out, total = clen + 26;
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(out, hdr, 16);
        out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n), isz = (uint32_t)n;
        memcpy(out + 18 + clen, &crc, 4); memcpy(out + 22 + clen, &isz, 4);
        return total;
    }
    void flush_blocks(bool all)
    {
        const size_t n_full = raw_.size() / kBlock, n_blocks = all ? (raw_.size() + kBlock - 1) / kBlock : n_full;
        if (n_blocks == 0) return;
        out_.resize(n_blocks * kOut);
        len_.assign(n_blocks, 0);
        auto work = [&](size_t t, size_t stride) {
            for (size_t i = t; i < n_blocks; i += stride) {
                const size_t off = i * kBlock, n = std::min(kBlock, raw_.size() - off);
                len_[i] = deflate_block(raw_.data() + off, n, out_.data() + i * kOut, level_);
            }
        };
        const size_t nt = std::min((size_t)threads_, n_blocks);
        if (nt <= 1) work(0, 1);
        else {
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nt; t++) pool.emplace_back(work, t, nt);
            work(0, nt);
            for (auto &th : pool) th.join();
        }
        for (size_t i = 0; i < n_blocks; i++) {
            if (len_[i] == 0 || fwrite(out_.data() + i * kOut, 1, len_[i], f_) != len_[i]) failed_ = true;
        }
        const size_t used = std::min(raw_.size(), n_blocks * kBlock);
        raw_.erase(raw_.begin(), raw_.begin() + (long)used);
    }
    FILE *f_ = nullptr;
    int level_ = Z_DEFAULT_COMPRESSION, threads_ = 1;
    bool failed_ = false;
    std::vector<uint8_t> raw_, out_;
    std::vector<size_t> len_;
};

// ------------------------------------------------------------------------------------------------ alignment records
struct Side {                      // one mate of one alignment
    uint64_t pos = NOT_ALIGNED;    // packed text position
    int score = 0, strand = 1;     // strand 1 '+', 2 '-'
    int count = 0;                 // num_sameScore
    int editdist = 0;
    const char *cigar = "";        // special CIGAR (M match run, m mismatch run, I D S)
};
struct PairAln { Side m[2]; long long insert = 0; };

// what bedtools bamtobed would print for a mapped record of the merged BAM (runMegaPath.sh:212-215): the spike filter's input
struct CovRec { uint32_t seq_id; uint64_t start0; uint32_t span; uint8_t mate; std::string name; };

struct ReadView { const char *name; uint32_t name_len; const char *seq; const char *qual; uint32_t len; };   // views, not NUL-terminated; seq = "ACGT" text, qual = raw ASCII

struct Options {
    std::string read_group, sample_name = "default", rg_option;   // -D / -A / -R (SOAP4.cpp:297-306)
    bool print_mdnm = false;                                      // -p
    int max_mapq = 60, min_mapq = 1;                              // ini [Score]
    int threads = 1;                                              // -T: number of .gout files
    int level = Z_DEFAULT_COMPRESSION;
};

// edit distance from the score identity (DV-DPfunctions.cpp:733-737, 1545-1548, 3466-3469); `len` is the read length the
// reference uses there.  *dis receives D - I - S.
inline int edit_dist(const char *sp, int len, int score, int *dis)
{
    int nI = 0, nD = 0, nS = 0, gap = 0, cur = 0;
    for (const char *p = sp; *p; p++) {
        if (*p >= '0' && *p <= '9') { cur = cur * 10 + (*p - '0'); continue; }
        if (cur > 0) {
            if (*p == 'I') { nI += cur; gap -= cur + 2; }
            else if (*p == 'D') { nD += cur; gap -= cur + 2; }
            else if (*p == 'S') nS += cur;
        }
        cur = 0;
    }
    if (dis) *dis = nD - nI - nS;
    return nI + nD + (len - nI - nS + gap - score) / 3;
}

class Writer {
public:
    Writer(const mp_index *ix, const Options &o) : ix_(ix), o_(o)
    {
        for (int i = 1; i < 256; i++) g_log_n_[i] = (int)(4.343 * log((double)i) + 0.5);       // bwase_initialize, CPUfunctions.h:104-108
        g_log_n_[0] = 0;
    }
    bool open(const std::string &prefix, const char *program_line)
    {
        mp_index_info info;
        if (mp_index_get_info(ix_, &info)) return false;
        std::string text = "@HD\tVN:1.3\tSO:unsorted\n@RG\tID:" + o_.read_group + "\tSM:" + o_.sample_name + "\t" + o_.rg_option + "\n";
        std::string refs;
        for (uint32_t s = 1; s <= info.num_seq; s++) {
            uint64_t actual = 0;
            mp_index_seq_info(ix_, s, nullptr, nullptr, &actual);
            const char *nm = mp_index_seq_name(ix_, s);
            char ln[64];
            snprintf(ln, sizeof ln, "\tLN:%u\n", (unsigned)actual);
            text += std::string("@SQ\tSN:") + nm + ln;
            const int32_t l_name = (int32_t)strlen(nm) + 1, l_ref = (int32_t)(uint32_t)actual;
            refs.append((const char *)&l_name, 4); refs.append(nm, (size_t)l_name); refs.append((const char *)&l_ref, 4);
        }
        text += program_line;
        std::string hdr = "BAM\1";
        const int32_t l_text = (int32_t)text.size(), n_ref = (int32_t)info.num_seq;
        hdr.append((const char *)&l_text, 4); hdr += text; hdr.append((const char *)&n_ref, 4); hdr += refs;
        for (size_t k = 0; k < 2 + (size_t)o_.threads; k++) {
            std::string path = prefix + (k == 0 ? ".dpout.1" : k == 1 ? ".unpair" : ".gout." + std::to_string(k - 1));
            files_.emplace_back(new Bgzf());
            if (!files_[k]->open(path, o_.level, o_.threads)) { fprintf(stderr, "[soap4-mi355x] cannot write %s\n", path.c_str()); return false; }
            files_[k]->write(hdr.data(), hdr.size());
        }
        return true;
    }
    bool close() { bool ok = true; for (auto &f : files_) ok = f->close() && ok; return ok; }

    enum { FILE_DPOUT = 0, FILE_UNPAIR = 1, FILE_GOUT = 2 };
    // records built by pair() / unpaired() go to the file in the order the caller appends them
    void append(int file, const std::string &records) { if (!records.empty()) files_[(size_t)file]->write(records.data(), records.size()); }

    // text bases for MD: the caller resolves [pos, pos+len) of the best alignments in bulk (mp_index_fetch_text)
    struct RefText { uint64_t pos = 0; const uint8_t *codes = nullptr; uint32_t len = 0; };

    // Properly paired output.  `res` already went through the annotated-FASTQ writer (the reference mutates the
    // results there first, BGS-IO.cpp:2103-2111); `best` was chosen before that (OutputDPResult.cpp:156-232).
    void pair(std::string &sink, int stage, const ReadView rd[2], const std::vector<PairAln> &res, size_t best, const RefText ref[2], std::vector<CovRec> *cov = nullptr) const
    {
        const PairAln &b = res[best];
        const bool a[2] = {b.m[0].pos != NOT_ALIGNED, b.m[1].pos != NOT_ALIGNED};
        if (!a[0] && !a[1]) return;
        Rec r[2];
        uint64_t tp[2] = {0, 0}; uint32_t chr[2] = {0, 0};
        long long trim[2] = {0, 0}; int rlen[2] = {(int)rd[0].len, (int)rd[1].len}, avgq[2] = {20, 20};
        const long long best_insert = (a[0] && a[1]) ? b.insert : 0;
        for (int k = 0; k < 2; k++) {
            r[k].strand = a[k] ? b.m[k].strand : 1;
            if (!a[k]) continue;
            std::string nc;
            trim[k] = chr_pos_boundary(rd[k].len, b.m[k].pos, b.m[k].cigar, tp[k], chr[k], nc);
            const char *use = trim[k] ? nc.c_str() : b.m[k].cigar;
            convert_cigar(use, r[k].cigar, nullptr);
            mis_info(rd[k], b.m[k].pos, use, trim[k], ref[k], r[k], avgq[k]);
            rlen[k] = read_len_with_cigar(b.m[k].cigar);
        }
        int best_pair_num = 0, best_pair_score = 0, sec_pair_score = 0;
        if (a[0] && a[1]) {
            best_pair_num = 1; best_pair_score = b.m[0].score + b.m[1].score;
            for (size_t i = 0; i < res.size(); i++) {
                if (i == best) continue;
                const int sum = res[i].m[0].score + res[i].m[1].score;
                if (sum == best_pair_score) best_pair_num++;
                else if (sum > sec_pair_score) sec_pair_score = sum;
            }
        }
        for (int k = 0; k < 2; k++) hit_counts(res, best, k, r[k].x0, r[k].x1);
        int mapq[2];
        bwa_pair(r[0].x0, r[0].x1, r[1].x0, r[1].x1, best_pair_score, best_pair_num, sec_pair_score, (int)res.size() - best_pair_num,
                 (int)rd[0].len, (int)rd[1].len, mapq[0], mapq[1]);
        for (int k = 0; k < 2; k++) if (trim[k]) mapq[k] = 0;
        for (int k = 0; k < 2; k++) {
            const int o = 1 - k;
            r[k].xa.clear();
            if (a[k] && res.size() > 1)
                for (size_t i = 0; i < res.size(); i++) {
                    if (i == best || res[i].m[0].score + res[i].m[1].score < best_pair_score || res[i].m[k].pos == NOT_ALIGNED) continue;
                    xa_entry(r[k].xa, res[i].m[k]);
                }
            r[k].mapped = a[k];
            if (a[k]) {
                if (a[o]) r[k].mapq = mapq[k];
                else { r[k].mapq = mapq_for_dp(r[k].x0, b.m[k].score, (int)rd[k].len, avgq[k]); if (trim[k]) r[k].mapq = 0; }
            }
            int flag = 1 | (k ? 128 : 64);
            if (a[0] && a[1]) flag |= 2;
            if (!a[k]) flag |= 4;
            if (!a[o]) flag |= 8;
            if (a[k] && b.m[k].strand == 2) flag |= 16;
            if (a[o] && b.m[o].strand == 2) flag |= 32;
            long long isize = 0;
            if (best_insert > 0) {
                if (tp[k] > tp[o]) isize = -(long long)(tp[k] + (uint64_t)rlen[k] - tp[o]);
                else isize = (long long)(tp[o] + (uint64_t)rlen[o] - tp[k]);
            }
            emit(sink, stage, rd[k], flag, chr[k], tp[k], chr[o], tp[o], isize, r[k]);
            if (cov && r[k].mapped && chr[k]) {
                uint32_t num = 0, span = 0;                     // reference span of the BAM CIGAR: M and D
                for (char ch : r[k].cigar) { if (ch >= '0' && ch <= '9') { num = num * 10 + (uint32_t)(ch - '0'); continue; } if (ch == 'M' || ch == 'D') span += num; num = 0; }
                cov->push_back(CovRec{chr[k], tp[k] - 1, span, (uint8_t)(k + 1), std::string(rd[k].name, rd[k].name_len)});
            }
        }
    }

    // Unpaired output.  `lst` already went through the annotated-FASTQ writer (BGS-IO.cpp:1406-1411 un-aligns hits that
    // span a sequence boundary, in -F mode only).
    void unpaired(std::string &sink, int stage, const ReadView rd[2], const std::vector<Side> lst[2], const RefText *ref_of_hit[2]) const
    {
        int best[2] = {-1, -1}, best_score[2] = {0, 0}, best_num[2] = {0, 0}, x1[2] = {0, 0}, deleted_end[2] = {0, 0};
        uint64_t tp[2] = {0, 0}; uint32_t chr[2] = {0, 0};
        Rec r[2];
        for (int k = 0; k < 2; k++) {
            const std::vector<Side> &l = lst[k];
            if (!l.empty()) {
                best[k] = 0; best_score[k] = l[0].score; best_num[k] = 1;
                for (size_t i = 1; i < l.size(); i++) {
                    if (l[i].score > best_score[k]) { best[k] = (int)i; best_score[k] = l[i].score; best_num[k] = 1; }
                    else if (l[i].score == best_score[k]) best_num[k]++;
                }
            }
            for (const Side &s : l) if (s.score < best_score[k]) x1[k]++;
            r[k].strand = 1;
            if (best[k] >= 0 && l[best[k]].pos != NOT_ALIGNED) {
                const Side &b = l[best[k]];
                std::string nc; int avgq;
                const long long trim = chr_pos_boundary(rd[k].len, b.pos, b.cigar, tp[k], chr[k], nc);
                if (trim) convert_cigar(nc.c_str(), r[k].cigar, nullptr); else convert_cigar(b.cigar, r[k].cigar, &deleted_end[k]);
                static const RefText none;
                mis_info(rd[k], b.pos, trim ? nc.c_str() : b.cigar, trim, ref_of_hit[k] ? ref_of_hit[k][best[k]] : none, r[k], avgq);
                r[k].mapq = bwa_single(best_num[k], x1[k]);
                if (r[k].mapq < o_.min_mapq) r[k].mapq = o_.min_mapq;
                if (trim) r[k].mapq = 0;
                r[k].mapped = true; r[k].strand = b.strand; r[k].x0 = best_num[k]; r[k].x1 = x1[k];
            } else best[k] = -1;
        }
        for (int k = 0; k < 2; k++) {
            const int o = 1 - k;
            r[k].xa.clear();
            for (size_t i = 0; i < lst[k].size(); i++) {
                const Side &s = lst[k][i];
                if ((int)i == best[k] || s.score < best_score[k] || s.pos == NOT_ALIGNED) continue;
                xa_entry(r[k].xa, s);
            }
            int flag = 1 | (k ? 128 : 64);
            if (best[k] < 0) flag |= 4;
            if (best[o] < 0) flag |= 8;
            if (best[k] >= 0 && r[k].strand == 2) flag |= 16;
            if (best[o] >= 0 && r[o].strand == 2) flag |= 32;
            long long isize = 0;
            if (chr[0] > 0 && chr[0] == chr[1]) {
                if (tp[o] > tp[k]) isize = (long long)(tp[o] - (uint64_t)deleted_end[o] + rd[o].len - tp[k]);
                else isize = -(long long)(tp[k] - (uint64_t)deleted_end[k] + rd[k].len - tp[o]);
            }
            emit(sink, stage, rd[k], flag, chr[k], tp[k], chr[o], tp[o], isize, r[k]);
        }
    }

    // reference span needed by MD for a special CIGAR: M + m + D columns
    static uint32_t ref_span(const char *sp)
    {
        uint32_t cur = 0, n = 0;
        for (const char *p = sp; *p; p++) {
            if (*p >= '0' && *p <= '9') { cur = cur * 10 + (uint32_t)(*p - '0'); continue; }
            if (*p == 'M' || *p == 'm' || *p == 'D') n += cur;
            cur = 0;
        }
        return n;
    }
    bool wants_text() const { return o_.print_mdnm; }

private:
    struct Rec {
        bool mapped = false;
        int strand = 1, mapq = 0, n_mis = 0, gap_open = 0, gap_ext = 0, x0 = 0, x1 = 0;
        std::string cigar, md, xa;
    };

    // convertToCigarStr (PE.cpp:381-445); a deletion in front leaves the running number untouched, as there
    static void convert_cigar(const char *sp, std::string &out, int *deleted_end)
    {
        int cur = 0, cur_m = 0;
        const size_t L = strlen(sp);
        out.clear();
        if (deleted_end) *deleted_end = 0;
        for (size_t i = 0; i < L; i++) {
            const char c = sp[i];
            if (c >= '0' && c <= '9') { cur = cur * 10 + (c - '0'); continue; }
            if (c == 'M' || c == 'm') { cur_m += cur; cur = 0; continue; }
            if (c == 'D' && ((out.empty() && cur_m == 0) || i == L - 1)) { if (i == L - 1 && deleted_end) *deleted_end = cur_m; continue; }
            if (c == 'D' || c == 'I' || c == 'S') {
                if (cur_m > 0) { out += std::to_string(cur_m); out += 'M'; cur_m = 0; }
                out += std::to_string(cur); out += c;
                cur = 0;
            }
        }
        if (cur_m > 0) { out += std::to_string(cur_m); out += 'M'; }
    }
    static int read_len_with_cigar(const char *p)          // readLengthWithCigar, BGS-IO.cpp:1917-1943
    {
        char op = ' '; int x = 0, len = 0;
        while (*p) {
            x = 0;
            while (*p && *p <= '9') { x = x * 10 + (*p - '0'); p++; }
            op = *p; if (*p) p++;
            if (op == 'M' || op == 'm' || op == 'D') len += x;
        }
        if (op == 'D') len -= x;
        return len;
    }
    // BoundaryCheckDP (BGS-IO.cpp:219-386) behind getChrAndPosWithBoundaryCheckDP (:412-431):
    // > 0 trimmed on the left, < 0 on the right, 0 none
    long long chr_pos_boundary(uint32_t read_len, uint64_t amb, const char *cigar, uint64_t &tp, uint32_t &chr, std::string &new_cigar) const
    {
        uint64_t seg_end = 0, start = 0, plen = 0;
        mp_index_chr_pos_segment(ix_, amb, &tp, &chr, &seg_end);
        mp_index_seq_info(ix_, chr, &start, &plen, nullptr);
        const uint64_t chr_end = start + plen - 1;
        const long long rl = (long long)read_len;
        new_cigar.clear();
        if (amb + (uint64_t)rl * 2 <= chr_end + 1 && amb + (uint64_t)rl * 2 <= seg_end + 1) return 0;
        if (chr_end < seg_end) seg_end = chr_end;
        std::string left, right;
        long long left_len = 0, right_len = 0, right_off = 0, left_s = 0, right_s = 0;
        uint64_t ref = amb;
        auto item = [](long long n, char op) { return std::to_string((int)n) + op; };
        for (const char *p = cigar; *p;) {
            long long num = 0;
            while (*p && *p <= '9') { num = num * 10 + (*p - '0'); p++; }
            const char op = *p; if (*p) p++;
            if (op == 'S') {
                if (ref <= seg_end) { left += item(num, 'S'); left_s += num; } else { right += item(num, 'S'); right_s += num; }
            } else if (op == 'M' || op == 'm') {
                if (ref > seg_end) { right_len += num; right += item(num, op); }
                else if (ref + (uint64_t)num <= seg_end + 1) { left_len += num; left += item(num, op); }
                else {
                    const long long l = (long long)(seg_end - ref + 1), rr = num - l;
                    left_len += l; left += item(l, op); right_len += rr; right += item(rr, op);
                }
                ref += (uint64_t)num;
            } else if (op == 'D') {
                if (ref > seg_end) { if (right.empty()) right_off = num; else right += item(num, 'D'); }
                else if (ref + (uint64_t)num <= seg_end + 1) left += item(num, 'D');
                else {
                    const long long l = (long long)(seg_end - ref + 1), rr = num - l;
                    left += item(l, 'D');
                    if (right.empty()) right_off = rr; else right += item(rr, 'D');
                }
                ref += (uint64_t)num;
            } else if (op == 'I') {
                if (ref == chr_end + 1) { /* exactly at the boundary: dropped on both sides */ }
                else if (ref <= seg_end) { left_len += num; left += item(num, 'I'); }
                else { right_len += num; right += item(num, 'I'); }
            }
        }
        if (!left_len || !right_len) return 0;
        long long ret; uint64_t corrected;
        if (left_len >= right_len) { new_cigar = left + item(rl - left_len - left_s, 'S'); corrected = amb; ret = -(rl - left_len - left_s); }
        else { new_cigar = item(rl - right_len - right_s, 'S') + right; corrected = seg_end + 1 + (uint64_t)right_off; ret = rl - right_len - right_s; }
        if (ret && corrected > amb) { uint64_t se; mp_index_chr_pos_segment(ix_, corrected, &tp, &chr, &se); }
        return ret;
    }
    // getMisInfoForDP (PE.cpp:459-620): XM / XO / XG counts always, MD text and the mismatch-quality average when needed
    void mis_info(const ReadView &rd, uint64_t pos, const char *sp, long long trim, const RefText &ref, Rec &r, int &avg_q) const
    {
        if (trim > 0) pos += (uint64_t)trim;
        r.n_mis = r.gap_open = r.gap_ext = 0; r.md.clear();
        double sum_q = 0.0;
        int cur = 0, cur_match = 0, q = 0;
        uint64_t t = pos;
        const size_t L = strlen(sp);
        // The reference text under the alignment was fetched for the alignment as the DP gave it (its span + 8).  An alignment trimmed on the LEFT at a sequence
        // boundary is read from pos + trim, where trim counts READ bases -- soft-clipped ones and inserted ones included (PE.cpp:459-620) --, so it may run
        // past that window: such a base is fetched on its own (rare: an alignment across a boundary whose left part is clipped or holds an insertion).
        auto base = [&](uint64_t p) -> char {
            if (ref.codes && p >= ref.pos && p - ref.pos < ref.len) return "ACGT"[ref.codes[p - ref.pos] & 3];
            uint8_t code = 0; const uint32_t one = 1;
            if (mp_index_fetch_text(ix_, &p, &one, 1, &code, 1) != 0) return 'N';
            return "ACGT"[code & 3];
        };
        const bool md = o_.print_mdnm;
        for (size_t i = 0; i < L; i++) {
            const char c = sp[i];
            if (c >= '0' && c <= '9') { cur = cur * 10 + (c - '0'); continue; }
            switch (c) {
            case 'M': cur_match += cur; q += cur; t += (uint64_t)cur; cur = 0; break;
            case 'm':
                if (md) { r.md += std::to_string(cur_match); r.md += base(t); }
                sum_q += (unsigned char)rd.qual[q];               // the reference never reverses or shifts the quality pointer
                for (int j = 1; j < cur; j++) { if (md) { r.md += '0'; r.md += base(t + (uint64_t)j); } sum_q += (unsigned char)rd.qual[q + j]; }
                q += cur; t += (uint64_t)cur; r.n_mis += cur; cur_match = 0; cur = 0; break;
            case 'I': q += cur; r.gap_open++; r.gap_ext += cur; cur = 0; break;
            case 'D':
                if (i == L - 1) break;
                if (md) { r.md += std::to_string(cur_match); r.md += '^'; for (int j = 0; j < cur; j++) r.md += base(t + (uint64_t)j); }
                t += (uint64_t)cur; r.gap_open++; r.gap_ext += cur; cur_match = 0; cur = 0; break;
            case 'S': q += cur; cur = 0; break;
            }
        }
        if (md) r.md += std::to_string(cur_match);
        avg_q = r.n_mis > 0 ? (int)(sum_q / r.n_mis) : 20;     // DEFAULT_QUAL_VALUE, PE.h:27
    }
    int bwa_single(int x0, int x1) const                        // bwaLikeSingleQualScore, BGS-IO.cpp:709-730
    {
        if (x0 > 1) return 0;
        if (x1 == 0) return 37;
        if (x1 > 255) x1 = 255;
        const int n = g_log_n_[x1];
        return 23 < n ? 0 : 23 - n;
    }
    void bwa_pair(int x0_0, int x1_0, int x0_1, int x1_1, int op_score, int op_num, int subop_score, int subop_num, int len0, int len1,
                  int &out0, int &out1) const                   // bwaLikePairQualScore, BGS-IO.cpp:826-876
    {
        int m0 = bwa_single(x0_0, x1_0), m1 = bwa_single(x0_1, x1_1), mp = 0;
        op_score *= 10; subop_score *= 10;
        if (m0 > 0 && m1 > 0) { mp = m0 + m1; if (mp > 60) mp = 60; m0 = m1 = mp; }
        else {
            if (op_num == 1) {
                if (subop_num == 0) mp = 29;
                else if (op_score - subop_score > 0.3 * ((len0 + len1) / 2)) mp = 23;
                else {
                    if (subop_num > 255) subop_num = 255;
                    mp = (op_score - subop_score) / 2 - g_log_n_[subop_num];
                    if (mp < 0) mp = 0;
                }
            }
            if (m0 == 0) m0 = mp + 7 < m1 ? mp + 7 : m1;
            if (m1 == 0) m1 = mp + 7 < m0 ? mp + 7 : m0;
        }
        out0 = m0; out1 = m1;
    }
    int mapq_for_dp(int n, int dp_score, int max_dp_score, int avg_q) const     // getMapQualScoreForDP, BGS-IO.cpp:903-935
    {
        static const double tab[6][2] = {{1.0, 1.0}, {0.875, 0.85}, {0.75, 0.7}, {0.625, 0.55}, {0.475, 0.4}, {0.325, 0.25}};   // :37
        if (n != 1) return o_.min_mapq;
        int d = 0;
        if (dp_score < max_dp_score) d = (int)((1.0 - (double)dp_score / max_dp_score) * 100.0 - 1.0) / 5 + 1;
        if (d > 5) d = 5;
        int a = (avg_q - 1) / 20;
        if (a > 1) a = 1; else if (a < 0) a = 0;
        const int m = (int)(o_.max_mapq * tab[d][a]);
        return m < o_.min_mapq ? o_.min_mapq : m;
    }
    // bestHitNum / secBestHitNum of one mate over the results of the pair (BGS-IO.cpp:2290-2344, 2363-2417)
    static void hit_counts(const std::vector<PairAln> &res, size_t best, int k, int &best_num, int &sec_num)
    {
        int bn = 0, sn = 0, bs = 0, ss = 0;
        uint64_t bp = NOT_ALIGNED, sp = NOT_ALIGNED;
        const Side &b = res[best].m[k];
        if (b.pos != NOT_ALIGNED) { bs = b.score; bp = b.pos; bn = 1; }
        if (b.pos != NOT_ALIGNED && res.size() > 1)
            for (size_t i = 0; i < res.size(); i++) {
                if (i == best) continue;
                const Side &s = res[i].m[k];
                if (s.score >= bs) {
                    if (s.score == bs) { if (s.pos != bp) bn += s.count; }
                    else { ss = bs; sn = bn; sp = bp; bs = s.score; bn = s.count; bp = s.pos; }
                } else if (s.score >= ss) {
                    if (s.score == ss) { if (s.pos != sp) sn += s.count; }
                    else { ss = s.score; sp = s.pos; sn = s.count; }
                }
            }
        best_num = bn; sec_num = sn;
    }
    void xa_entry(std::string &xa, const Side &s) const
    {
        uint64_t tp; uint32_t chr;
        mp_index_chr_and_pos(ix_, s.pos, &tp, &chr);
        std::string cg;
        convert_cigar(s.cigar, cg, nullptr);
        xa += mp_index_seq_name(ix_, chr); xa += ','; xa += s.strand == 2 ? '-' : '+'; xa += std::to_string((unsigned long long)tp); xa += ',';
        xa += cg; xa += ','; xa += std::to_string(s.editdist); xa += ';';
    }
    static void put32(std::string &b, uint32_t v) { b.append((const char *)&v, 4); }
    static void tag_i(std::string &b, const char *t, int v) { b.append(t, 2); b += 'i'; b.append((const char *)&v, 4); }
    static void tag_z(std::string &b, const char *t, const std::string &v) { b.append(t, 2); b += 'Z'; b += v; b += '\0'; }

    // initializeSAMAlgnmt / initializeSAMAlgnmt2 (BGS-IO.cpp:433-675) + the core fields the callers fill in
    void emit(std::string &rec_, int stage, const ReadView &rd, int flag, uint32_t c_self, uint64_t t_self, uint32_t c_oth, uint64_t t_oth, long long isize, const Rec &r) const
    {
        const int32_t tid = c_self == 0 ? (c_oth == 0 ? -1 : (int32_t)c_oth - 1) : (int32_t)c_self - 1;
        const int32_t pos = t_self == 0 ? (t_oth == 0 ? -1 : (int32_t)(t_oth - 1)) : (int32_t)(t_self - 1);
        const int32_t mtid = c_oth == 0 ? (c_self == 0 ? -1 : (int32_t)c_self - 1) : (int32_t)c_oth - 1;
        const int32_t mpos = t_oth == 0 ? (t_self == 0 ? -1 : (int32_t)(t_self - 1)) : (int32_t)(t_oth - 1);
        std::string cig;                                        // AssignCigarStrToSAMIU (BGS-IO.cpp:138-161): zero-length operations are dropped
        uint32_t n_cig = 0;
        if (r.mapped) {
            uint32_t num = 0;
            for (char ch : r.cigar) {
                if (ch >= '0' && ch <= '9') { num = num * 10 + (uint32_t)(ch - '0'); continue; }
                if (num > 0) { const uint32_t op = ch == 'M' ? 0 : ch == 'I' ? 1 : ch == 'D' ? 2 : 4; put32(cig, num << 4 | op); n_cig++; num = 0; }
            }
        }
        const uint32_t l_name = rd.name_len + 1, L = rd.len;
        const uint32_t mapq = r.mapped ? (uint32_t)(r.mapq & 0xFF) : 0u;
        const size_t at = rec_.size();                          // records are appended to the caller's buffer (one per file kind)
        put32(rec_, 0);                                         // block_size, patched below
        put32(rec_, (uint32_t)tid); put32(rec_, (uint32_t)pos);
        put32(rec_, 0u << 16 | mapq << 8 | l_name);             // bin = bam_reg2bin(0, 0) = 0 for every record (BGS-IO.cpp:438)
        put32(rec_, (uint32_t)flag << 16 | n_cig);
        put32(rec_, L); put32(rec_, (uint32_t)mtid); put32(rec_, (uint32_t)mpos); put32(rec_, (uint32_t)(int32_t)isize);
        rec_.append(rd.name, rd.name_len); rec_ += '\0';
        rec_ += cig;
        const bool neg = r.strand == 2;
        auto nt16 = [&](uint32_t i) -> uint8_t {                // bam_nt16_table of A C G T
            char c = neg ? rd.seq[L - 1 - i] : rd.seq[i];
            if (neg) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : 8;
        };
        for (uint32_t i = 0; i < L; i += 2) rec_ += (char)(nt16(i) << 4 | (i + 1 < L ? nt16(i + 1) : 0));
        for (uint32_t i = 0; i < L; i++) rec_ += (char)((unsigned char)rd.qual[neg ? L - 1 - i : i] - 33);
        tag_z(rec_, "RG", o_.read_group);
        if (r.mapped) {
            if (o_.print_mdnm && r.n_mis + r.gap_ext >= 0) tag_i(rec_, "NM", r.n_mis + r.gap_ext);
            if (r.x0 >= 0) tag_i(rec_, "X0", r.x0);
            if (r.x1 >= 0) tag_i(rec_, "X1", r.x1);
            if (r.n_mis >= 0) tag_i(rec_, "XM", r.n_mis);
            if (r.gap_open >= 0) tag_i(rec_, "XO", r.gap_open);
            if (r.gap_ext >= 0) tag_i(rec_, "XG", r.gap_ext);
            if (o_.print_mdnm && !r.md.empty()) tag_z(rec_, "MD", r.md);
            if (!r.xa.empty()) tag_z(rec_, "XA", r.xa);
            if (stage != -1) tag_i(rec_, "PH", stage);
        } else if (!r.xa.empty()) tag_z(rec_, "XA", r.xa);
        const uint32_t bs = (uint32_t)(rec_.size() - at) - 4;
        memcpy(&rec_[at], &bs, 4);
    }

    const mp_index *ix_;
    Options o_;
    int g_log_n_[256];
    std::vector<std::unique_ptr<Bgzf>> files_;
};

}  // namespace mpbam
#endif
