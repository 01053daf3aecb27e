// megapath_amd/csrc/host/fast_inflate.h -- product code: gzip (RFC 1952) / deflate (RFC 1951) decoder of the executable's reader.
//
// runMegaPath.sh hands the aligner gzip'ed FASTQ (runMegaPath.sh:136: ${PREFIX}.bbduk_1.fq.gz); the reference reads it through
// zlib's gzread inside kseq (QueryParser.cpp, aio_thread.cpp:247).  zlib inflates FASTQ at 350-400 MB/s per stream, i.e. ~1 M pairs/s
// for the two files together: a thirtieth of what the device stage takes.  This decoder works on the whole mapped file with a
// 64-bit bit buffer, 11-bit (literal/length) and 8-bit (distance) first-level tables with second-level tables behind them and
// 8-byte match copies, resumable at any symbol when the output chunk is full.  It checks what zlib checks: block types, code
// tables, distances against the history, stored-block length complements, CRC-32 and ISIZE of every member (the CRC by the
// consumer thread, see soap4_main.cpp), and concatenated members (bgzip, `cat a.gz b.gz`).
// Output goes into chunk buffers that begin with the last 32 KiB of the chunk before, so a match never leaves its buffer.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

namespace mpz {

enum { OK = 0, NEED_OUTPUT = 1, MEMBER_END = 2, STREAM_END = 3, AT_BLOCK = 4, ERR = -1 };

class Inflater {
public:
    // the whole compressed file
    void init(const uint8_t *data, size_t len)
    {
        in_ = data; in_end_ = data + len; p_ = data; bb_ = 0; bc_ = 0; virt_ = 0; state_ = S_HEADER; err_.clear(); resumed_ = false;
        member_out_ = 0; member_known_ = false;
    }
    // the same, but decoding starts at a deflate block header `bitpos` bits into the data (the parallel reader: par_gzip in soap4_main.cpp)
    void init_at_block(const uint8_t *data, size_t len, uint64_t bitpos)
    {
        init(data, len);
        p_ = data + (size_t)(bitpos >> 3);
        refill();
        drop((unsigned)(bitpos & 7u));
        state_ = S_BLOCK;
    }
    // run() returns AT_BLOCK every time it is about to read a block header (bit_position() says where); calling it again goes on
    void stop_at_blocks(bool on) { block_stops_ = on; }
    uint64_t bit_position() const { return (uint64_t)(p_ - in_) * 8u + (uint64_t)virt_ * 8u - bc_; }
    const std::string &error() const { return err_; }
    uint32_t member_crc() const { return crc_; }        // valid after MEMBER_END: the trailer's CRC-32 and length (mod 2^32)
    uint32_t member_isize() const { return isize_; }

    // Decodes into out[pos ..): out[0 .. pos) must hold the output so far (at least the last 32 KiB of it).  Returns
    //   NEED_OUTPUT  fewer than 320 bytes of room left before `cap` (call again with a fresh chunk that starts with the history)
    //   MEMBER_END   a gzip member ended (check its CRC, then call again: another member or the end of the file may follow)
    //   STREAM_END   the input is used up     ERR  malformed input (error())
    // Out = uint8_t: bytes.  Out = uint16_t: the same symbols one per 16-bit word, for a decoder that does not know the 32 KiB in front
    // of its first block: the caller fills out[0 .. 32768) with markers 0x8000 | i and starts at pos = 32768, so that a match reaching
    // back before the start copies markers, to be replaced once the window is known.
    template <typename Out>
    int run(Out *out, size_t &pos, size_t cap)
    {
        ms_ = (int64_t)pos - (int64_t)member_out_;                           // where the current member's output began, in this call's coordinates
        const int rc = run_states(out, pos, cap);
        member_out_ = (uint64_t)((int64_t)pos - ms_);
        return rc;
    }

private:
    template <typename Out>
    int run_states(Out *out, size_t &pos, size_t cap)
    {
        for (;;) {
            switch (state_) {
            case S_HEADER: {
                if (!header()) return err_.empty() ? STREAM_END : ERR;
                ms_ = (int64_t)pos; member_known_ = true;
                state_ = S_BLOCK;
                break;
            }
            case S_BLOCK: {
                if (block_stops_ && !resumed_) { resumed_ = true; return AT_BLOCK; }
                resumed_ = false;
                refill();
                if (virt_ > 8) return fail("truncated deflate stream (block header)");
                last_ = (bb_ & 1u) != 0;
                const unsigned type = (unsigned)(bb_ >> 1) & 3u;
                drop(3);
                if (type == 0) {
                    drop(bc_ & 7u);                                            // to the byte boundary
                    refill();
                    if (virt_ && (int)(bc_ >> 3) - (int)virt_ < 4) return fail("truncated stored block");
                    const unsigned len = (unsigned)(bb_ & 0xFFFFu), nlen = (unsigned)((bb_ >> 16) & 0xFFFFu);
                    drop(32);
                    if ((len ^ 0xFFFFu) != nlen) return fail("stored block length check failed");
                    stored_left_ = len;
                    state_ = S_STORED;
                } else if (type == 1) { fixed_tables(); state_ = S_HUFF; }
                else if (type == 2) { if (!dynamic_tables()) return ERR; state_ = S_HUFF; }
                else return fail("invalid deflate block type");
                break;
            }
            case S_STORED: {
                if (virt_ && (long)stored_left_ > (long)(bc_ >> 3) - (long)virt_ + (long)(in_end_ - p_)) return fail("truncated stored block");
                while (stored_left_) {
                    if (pos + 320 > cap) return NEED_OUTPUT;
                    // bytes still in the bit buffer first (it is byte-aligned here), then straight from the input
                    if (bc_ >= 8) { out[pos++] = (Out)(uint8_t)bb_; drop(8); stored_left_--; continue; }
                    bb_ = 0; bc_ = 0;                                          // (bits of the next byte that refill() had OR-ed in without counting them)
                    const size_t room = cap - 320 - pos + 1, avail = (size_t)(in_end_ - p_);
                    size_t n = stored_left_ < room ? stored_left_ : room;
                    if (n > avail) { if (avail == 0) return fail("truncated stored block"); n = avail; }
                    if (sizeof(Out) == 1) memcpy(out + pos, p_, n);
                    else for (size_t i = 0; i < n; i++) out[pos + i] = (Out)p_[i];
                    p_ += n; pos += n; stored_left_ -= (unsigned)n;
                }
                state_ = last_ ? S_TRAILER : S_BLOCK;
                break;
            }
            case S_HUFF: {
                const int rc = huffman(out, pos, cap);
                if (rc != OK) return rc;
                state_ = last_ ? S_TRAILER : S_BLOCK;
                break;
            }
            case S_TRAILER: {
                drop(bc_ & 7u);
                refill();
                if (virt_ && (int)(bc_ >> 3) - (int)virt_ < 8) return fail("truncated gzip trailer");
                if (bc_ < 64) {                                               // fewer than 8 bytes left: read what is there
                    uint8_t t[8];
                    for (int i = 0; i < 8; i++) { refill(); if (bc_ < 8) return fail("truncated gzip trailer"); t[i] = (uint8_t)bb_; drop(8); }
                    crc_ = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                    isize_ = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
                } else { crc_ = (uint32_t)bb_; isize_ = (uint32_t)(bb_ >> 32); bb_ = 0; bc_ -= 64; }         // (a shift by 64 is undefined)
                state_ = S_HEADER;
                return MEMBER_END;
            }
            case S_DONE: return err_.empty() ? STREAM_END : ERR;
            }
        }
    }

    enum { S_HEADER, S_BLOCK, S_STORED, S_HUFF, S_TRAILER, S_DONE };
    enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_LIT2 = 5 };     // K_LIT2: two literals in one first-level entry
    static constexpr unsigned LBITS = 11, DBITS = 8;
    const uint8_t *in_ = nullptr, *in_end_ = nullptr, *p_ = nullptr;
    uint64_t bb_ = 0;
    unsigned bc_ = 0;
    int state_ = S_HEADER;
    bool last_ = false, block_stops_ = false, resumed_ = false;
    unsigned stored_left_ = 0;
    unsigned virt_ = 0;          // zero bytes appended past the end of the input (refill)
    // Bytes of the CURRENT member written before the run() call in progress: a match may reach back that far plus what this call has written,
    // not into an earlier member's output (zlib: "invalid distance too far back"), even though that output lies in front of it in the same
    // buffer (concatenated members, the groups of a BGZF file, the 32 KiB of history a chunked caller keeps).  A decoder started in the
    // middle of a stream (init_at_block) does not know where its member began: no limit until it reads a member header itself.
    uint64_t member_out_ = 0;
    int64_t ms_ = 0;             // run() in progress: out[] index at which the current member's output began (negative: before the buffer)
    bool member_known_ = false;
    uint32_t crc_ = 0, isize_ = 0;
    std::string err_;
    // entry: value << 16 | extra (or second-level bits) << 8 | kind << 5 | code length
    uint32_t lit_[(1u << LBITS) + 2048], dist_[(1u << DBITS) + 1024], fast_[1u << LBITS];

    int fail(const char *m) { err_ = m; state_ = S_DONE; return ERR; }
    void drop(unsigned n) { bb_ >>= n; bc_ -= n; }
    void refill()
    {
        if (p_ + 8 <= in_end_) {
            uint64_t w;
            memcpy(&w, p_, 8);
            bb_ |= w << bc_;
            p_ += (63 - bc_) >> 3;
            bc_ |= 56;
        } else
            // the last bytes one at a time; past the end the buffer is topped up with zero bits that are counted: a well-formed stream
            // never consumes them (a trailer of eight bytes follows its last code), a truncated one is caught by the counter
            while (bc_ <= 56) { if (p_ < in_end_) bb_ |= (uint64_t)*p_++ << bc_; else virt_++; bc_ += 8; }
    }

    // gzip member header (RFC 1952 2.3); at the end of the input: S_DONE
    bool header()
    {
        // the bit buffer may hold whole bytes read ahead: give them back
        if (virt_) { if ((bc_ >> 3) < virt_) { fail("truncated gzip stream"); return false; } p_ -= (bc_ >> 3) - virt_; virt_ = 0; }
        else p_ -= bc_ >> 3;
        bb_ = 0; bc_ = 0;
        while (p_ < in_end_ && *p_ == 0) p_++;                              // zero padding between or after members (as gzip -d tolerates)
        if (p_ == in_end_) { state_ = S_DONE; return false; }
        if (in_end_ - p_ < 18) { fail("truncated gzip header"); return false; }
        if (p_[0] != 0x1f || p_[1] != 0x8b) { fail("not a gzip member (bad magic)"); return false; }
        if (p_[2] != 8) { fail("unknown gzip compression method"); return false; }
        const unsigned flg = p_[3];
        if (flg & 0xE0u) { fail("reserved gzip header flags set"); return false; }
        p_ += 10;
        if (flg & 4u) { if (in_end_ - p_ < 2) { fail("truncated gzip header"); return false; } const size_t x = p_[0] | (size_t)p_[1] << 8; p_ += 2; if ((size_t)(in_end_ - p_) < x) { fail("truncated gzip header"); return false; } p_ += x; }
        for (unsigned bit = 8; bit <= 16; bit <<= 1)                          // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) { while (p_ < in_end_ && *p_) p_++; if (p_ == in_end_) { fail("truncated gzip header"); return false; } p_++; }
        if (flg & 2u) { if (in_end_ - p_ < 2) { fail("truncated gzip header"); return false; } p_ += 2; }
        return true;
    }

    static unsigned rev(unsigned code, unsigned len)
    {
        unsigned r = 0;
        for (unsigned i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
        return r;
    }
    // canonical Huffman decoding table: first level of `pbits`, second-level tables appended; entry(sym, len) makes the payload
    template <typename F>
    bool build(const uint8_t *lens, unsigned n, uint32_t *table, unsigned pbits, unsigned table_cap, F &&entry)
    {
        unsigned count[16] = {0}, next[16];
        for (unsigned s = 0; s < n; s++) count[lens[s]]++;
        count[0] = 0;
        unsigned code = 0, maxlen = 0;
        long left = 1;
        for (unsigned l = 1; l <= 15; l++) {
            left = left * 2 - (long)count[l];
            if (left < 0) { err_ = "over-subscribed Huffman code"; return false; }
            code = (code + count[l - 1]) << 1;
            next[l] = code;
            if (count[l]) maxlen = l;
        }
        const uint32_t bad = (uint32_t)K_BAD << 5 | 1u;
        for (unsigned i = 0; i < (1u << pbits); i++) table[i] = bad;
        if (maxlen == 0) return true;                                       // no code at all (a block without distances)
        // second-level sizes: for every first-level prefix the longest code that starts with it
        unsigned sub_bits[1u << 11];
        if (maxlen > pbits) {
            memset(sub_bits, 0, sizeof(unsigned) << pbits);
            unsigned c[16];
            memcpy(c, next, sizeof c);
            for (unsigned s = 0; s < n; s++) {
                const unsigned l = lens[s];
                if (l <= pbits) { if (l) c[l]++; continue; }
                const unsigned r = rev(c[l]++, l) & ((1u << pbits) - 1u);
                if (l - pbits > sub_bits[r]) sub_bits[r] = l - pbits;
            }
        }
        unsigned used = 1u << pbits;
        for (unsigned s = 0; s < n; s++) {
            const unsigned l = lens[s];
            if (!l) continue;
            const unsigned r = rev(next[l]++, l);
            if (l <= pbits) {
                const uint32_t e = entry(s) | l;
                for (unsigned i = r; i < (1u << pbits); i += 1u << l) table[i] = e;
            } else {
                const unsigned pre = r & ((1u << pbits) - 1u), sb = sub_bits[pre];
                if ((table[pre] >> 5 & 7u) != K_SUB) {
                    if (used + (1u << sb) > table_cap) { err_ = "Huffman table overflow"; return false; }
                    table[pre] = (uint32_t)used << 16 | sb << 8 | (uint32_t)K_SUB << 5 | pbits;
                    for (unsigned i = 0; i < (1u << sb); i++) table[used + i] = bad;
                    used += 1u << sb;
                }
                const unsigned start = table[pre] >> 16, rem = l - pbits;
                const uint32_t e = entry(s) | rem;
                for (unsigned i = r >> pbits; i < (1u << sb); i += 1u << rem) table[start + i] = e;
            }
        }
        return true;
    }
    static uint32_t lit_entry(unsigned s)
    {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return (uint32_t)s << 16 | (uint32_t)K_LIT << 5;
        if (s == 256) return (uint32_t)K_EOB << 5;
        if (s > 285) return (uint32_t)K_BAD << 5;
        return (uint32_t)base[s - 257] << 16 | (uint32_t)extra[s - 257] << 8 | (uint32_t)K_LEN << 5;
    }
    static uint32_t dist_entry(unsigned s)
    {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s > 29) return (uint32_t)K_BAD << 5;
        return (uint32_t)base[s] << 16 | (uint32_t)extra[s] << 8 | (uint32_t)K_LEN << 5;
    }
    void fixed_tables()
    {
        uint8_t l[288 + 32];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        (void)build(l, 288, lit_, LBITS, sizeof lit_ / 4, lit_entry);
        for (int i = 0; i < 32; i++) l[i] = 5;
        (void)build(l, 32, dist_, DBITS, sizeof dist_ / 4, dist_entry);
        pair_literals();
    }
    bool dynamic_tables()
    {
        refill();
        if (bc_ < 14) { fail("truncated dynamic block header"); return false; }
        const unsigned hlit = (unsigned)(bb_ & 31u) + 257, hdist = (unsigned)((bb_ >> 5) & 31u) + 1, hclen = (unsigned)((bb_ >> 10) & 15u) + 4;
        drop(14);
        if (hlit > 286 || hdist > 30) { fail("too many length or distance codes"); return false; }
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (unsigned i = 0; i < hclen; i++) { refill(); if (bc_ < 3) { fail("truncated dynamic block header"); return false; } cl[order[i]] = (uint8_t)(bb_ & 7u); drop(3); }
        uint32_t ct[(1u << 7) + 8];
        if (!build(cl, 19, ct, 7, sizeof ct / 4, [](unsigned s) { return (uint32_t)s << 16 | (uint32_t)K_LIT << 5; })) { state_ = S_DONE; return false; }
        uint8_t lens[286 + 30 + 138];
        unsigned n = 0;
        while (n < hlit + hdist) {
            refill();
            const uint32_t e = ct[bb_ & 127u];
            const unsigned l = e & 31u;
            if ((e >> 5 & 7u) != K_LIT || bc_ < l + 7) { fail("bad code-length code"); return false; }
            drop(l);
            const unsigned s = e >> 16;
            if (s < 16) { lens[n++] = (uint8_t)s; continue; }
            unsigned rep, val = 0;
            if (s == 16) { if (!n) { fail("repeat without a previous length"); return false; } val = lens[n - 1]; rep = 3 + (unsigned)(bb_ & 3u); drop(2); }
            else if (s == 17) { rep = 3 + (unsigned)(bb_ & 7u); drop(3); }
            else { rep = 11 + (unsigned)(bb_ & 127u); drop(7); }
            if (n + rep > hlit + hdist) { fail("code lengths overrun"); return false; }
            while (rep--) lens[n++] = (uint8_t)val;
        }
        if (lens[256] == 0) { fail("no end-of-block code"); return false; }
        if (!build(lens, hlit, lit_, LBITS, sizeof lit_ / 4, lit_entry) || !build(lens + hlit, hdist, dist_, DBITS, sizeof dist_ / 4, dist_entry)) { state_ = S_DONE; return false; }
        pair_literals();
        return true;
    }
    // FASTQ is mostly literals with short codes (bases: 2-3 bits, qualities: 3-5).  fast_[i] is what the first-level index i starts with:
    //   0                                   not a literal: look at lit_[i]
    //   b1 | len << 16 | 1 << 24            one literal
    //   b1 | b2 << 8 | len << 16 | 2 << 24  two literals (the second code is what lit_ holds for i >> l1, valid when it is no longer than
    //                                       the LBITS - l1 bits that are left of the index)
    void pair_literals()
    {
        for (unsigned i = 0; i < (1u << LBITS); i++) {
            const uint32_t e1 = lit_[i];
            if ((e1 >> 5 & 7u) != K_LIT) { fast_[i] = 0; continue; }
            const unsigned l1 = e1 & 31u;
            const uint32_t e2 = lit_[i >> l1];
            const unsigned l2 = e2 & 31u;
            if ((e2 >> 5 & 7u) == K_LIT && l1 + l2 <= LBITS) fast_[i] = (e1 >> 16 & 0xFFu) | (e2 >> 16 & 0xFFu) << 8 | (uint32_t)(l1 + l2) << 16 | 2u << 24;
            else fast_[i] = (e1 >> 16 & 0xFFu) | (uint32_t)l1 << 16 | 1u << 24;
        }
    }

    template <typename Out>
    int huffman(Out *out, size_t &pos_io, size_t cap)
    {
        size_t pos = pos_io;
        const uint32_t *lit = lit_, *dist = dist_;
        for (;;) {
            if (pos + 320 > cap) { pos_io = pos; return NEED_OUTPUT; }
            refill();
            if (virt_ > 8) { pos_io = pos; return fail("truncated deflate stream"); }
            // literal run: up to four first-level entries (<= 44 bits of the >= 56 in the buffer), one or two literals each
            {
                uint32_t f = fast_[bb_ & ((1u << LBITS) - 1u)];
                int n = 0;
                while (f && n < 4) {
                    out[pos] = (Out)(uint8_t)f; out[pos + 1] = (Out)(uint8_t)(f >> 8);
                    pos += f >> 24;
                    drop(f >> 16 & 31u);
                    f = fast_[bb_ & ((1u << LBITS) - 1u)];
                    n++;
                }
                if (n == 4) continue;
                if (n) refill();                                            // (the low bits stay: the look-up below sees what fast_ saw)
            }
            uint32_t e = lit[bb_ & ((1u << LBITS) - 1u)];
            unsigned kind = e >> 5 & 7u;
            if (kind == K_SUB) {
                drop(e & 31u);
                e = lit[(e >> 16) + (bb_ & ((1u << (e >> 8 & 31u)) - 1u))];
                kind = e >> 5 & 7u;
                if (kind == K_LIT) { drop(e & 31u); out[pos++] = (Out)(uint8_t)(e >> 16); continue; }
            }
            if (bc_ < (e & 31u)) { pos_io = pos; return fail("truncated deflate stream"); }
            drop(e & 31u);
            if (kind == K_EOB) { pos_io = pos; return OK; }
            if (kind != K_LEN) { pos_io = pos; return fail("invalid literal/length code"); }
            const unsigned xb = e >> 8 & 31u;
            unsigned len = (e >> 16) + (unsigned)(bb_ & ((1u << xb) - 1u));
            drop(xb);
            refill();
            uint32_t d = dist[bb_ & ((1u << DBITS) - 1u)];
            if ((d >> 5 & 7u) == K_SUB) { drop(d & 31u); d = dist[(d >> 16) + (bb_ & ((1u << (d >> 8 & 31u)) - 1u))]; }
            if ((d >> 5 & 7u) != K_LEN) { pos_io = pos; return fail("invalid distance code"); }
            drop(d & 31u);
            const unsigned db = d >> 8 & 31u;
            if (bc_ < db) { pos_io = pos; return fail("truncated deflate stream"); }
            const size_t distance = (d >> 16) + (size_t)(bb_ & ((1u << db) - 1u));
            drop(db);
            if (distance > pos) { pos_io = pos; return fail("distance reaches before the start of the output"); }
            if (member_known_ && (int64_t)distance > (int64_t)pos - ms_) { pos_io = pos; return fail("invalid distance too far back"); }
            Out *dst = out + pos;
            const Out *src = dst - distance;
            pos += len;
            constexpr unsigned PER = 8 / sizeof(Out);                       // symbols per 8-byte copy
            if (distance >= PER) {
                // 8 bytes at a time; may write up to 7 bytes past the match (room is kept: 320 symbols of slack)
                do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += PER; dst += PER; len = len > PER ? len - PER : 0; } while (len);
            } else if (distance == 1) { const Out v = *src; for (unsigned i = 0; i < len; i++) dst[i] = v; }
            else while (len--) *dst++ = *src++;
        }
    }
};

}  // namespace mpz
