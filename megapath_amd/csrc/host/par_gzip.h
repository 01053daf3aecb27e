// megapath_amd/csrc/host/par_gzip.h -- product code: ONE gzip stream inflated by several threads (the reader of soap4_main.cpp).
//
// runMegaPath.sh:136 hands the aligner plain `gzip` files; a deflate stream has no index, and every block may refer to the 32 KiB in front
// of it, so the stream is decoded front to back by one thread (fast_inflate.h: 0.85 GB/s, 2 M pairs/s for the whole executable against
// 22 M with plain input).  What is done here instead (the scheme of pugz / rapidgzip, restated from the idea, for text input):
//
//   * the compressed file is cut into SEGMENTS of a few MiB.  For every segment but the first a thread looks for the first bit position
//     at or after the segment's first byte that holds the header of a dynamic-Huffman block which is not the last of its member: both
//     code tables complete (zlib's always are), an end-of-block code, and -- decoding this block and the next -- no invalid code, no
//     invalid distance, nothing but text among the literals.  (A position that passes all that by chance is caught below.)
//   * from there the segment is decoded WITHOUT its 32 KiB window into 16-bit symbols: a byte, or 0x8000 | i for "byte i of the
//     unknown window" (the output buffer simply begins with 32 768 such markers, so a match that reaches back before the segment copies
//     markers).  The segment ends where the next segment's start was found: the decoder stops when it stands EXACTLY on that bit.
//     A start that the decoder of the true chain passes without standing on it was a false one: the chain jumps over that segment (its
//     work is thrown away) to the segment whose start the decoder did stop on.
//   * in file order, the window of each segment (the last 32 KiB of everything before it, known once the segment before is resolved) turns
//     the markers into bytes: first the segment's own last 32 KiB (so the next segment can start), then the rest, side by side with the
//     other segments; the CRC-32 of every piece is taken there and combined in order (crc32_combine), and compared with each member's
//     trailer, as zlib would.
//
// Nothing is trusted that the sequential decoder would not accept: every block is decoded by the same code (mpz::Inflater), the chain of
// segments starts at the true beginning of the stream and every link is a position where the decoder itself stood on a block boundary,
// and the member's CRC-32 and length are checked over the resolved bytes.  Files without a findable start (binary data, stored blocks
// only) simply end up decoded by one thread.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "crc32_fold.h"
#include "fast_inflate.h"

namespace mpz {

// ---------------------------------------------------------------------------------------------- where does a block start?
struct BitPeek {
    const uint8_t *d;
    size_t len;
    // 57 or more valid bits at bit position b (the caller keeps 16 bytes clear of the end)
    uint64_t at(uint64_t b) const { uint64_t w; memcpy(&w, d + (b >> 3), 8); return w >> (b & 7u); }
};

// the header of a dynamic block that is not the last one, with complete code tables, at bit b?  (RFC 1951 3.2.7)
inline bool plausible_dynamic_header(const BitPeek &in, uint64_t b)
{
    uint64_t v = in.at(b);
    if ((v & 7u) != 4u) return false;                                      // BFINAL = 0, BTYPE = 10
    const unsigned hlit = (unsigned)(v >> 3 & 31u) + 257, hdist = (unsigned)(v >> 8 & 31u) + 1, hclen = (unsigned)(v >> 13 & 15u) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    uint64_t p = b + 17;
    unsigned kraft = 0;
    for (unsigned i = 0; i < hclen; i++, p += 3) {
        const unsigned l = (unsigned)(in.at(p) & 7u);
        cl[order[i]] = (uint8_t)l;
        if (l) kraft += 128u >> l;
    }
    if (kraft != 128u) return false;                                       // the code-length code is complete
    // canonical codes of the code-length code, as a 7-bit look-up
    uint8_t sym_of[128], len_of[128];
    {
        unsigned count[8] = {0}, next[8], code = 0;
        for (int s = 0; s < 19; s++) count[cl[s]]++;
        count[0] = 0;
        for (int l = 1; l <= 7; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
        for (int s = 0; s < 19; s++) {
            const unsigned l = cl[s];
            if (!l) continue;
            unsigned c = next[l]++, r = 0;
            for (unsigned i = 0; i < l; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
            for (unsigned i = r; i < 128; i += 1u << l) { sym_of[i] = (uint8_t)s; len_of[i] = (uint8_t)l; }
        }
    }
    uint8_t lens[286 + 30 + 138];
    unsigned n = 0;
    while (n < hlit + hdist) {
        if ((p >> 3) + 16 > in.len) return false;
        v = in.at(p);
        const unsigned s = sym_of[v & 127u], l = len_of[v & 127u];
        p += l; v >>= l;
        if (s < 16) { lens[n++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (!n) return false; val = lens[n - 1]; rep = 3 + (unsigned)(v & 3u); p += 2; }
        else if (s == 17) { rep = 3 + (unsigned)(v & 7u); p += 3; }
        else { rep = 11 + (unsigned)(v & 127u); p += 7; }
        if (n + rep > hlit + hdist) return false;
        while (rep--) lens[n++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;
    unsigned long k = 0;
    for (unsigned s = 0; s < hlit; s++) if (lens[s]) k += 32768u >> lens[s];
    if (k != 32768u) return false;                                         // literal/length code complete
    k = 0;
    unsigned nd = 0;
    for (unsigned s = hlit; s < hlit + hdist; s++) if (lens[s]) { k += 32768u >> lens[s]; nd++; }
    return k == 32768u || nd <= 1;                                         // distance code complete, or at most one code
}

// ---------------------------------------------------------------------------------------------- markers -> bytes
inline void resolve_scalar(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) { const uint16_t s = sym[i]; out[i] = (s & 0x8000u) ? window[s & 0x7FFFu] : (uint8_t)s; }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void resolve_avx2(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out)
{
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(sym + i)), b = _mm256_loadu_si256((const __m256i *)(sym + i + 16));
        if (_mm256_movemask_epi8(_mm256_or_si256(a, b)) & 0xAAAAAAAA) { resolve_scalar(sym + i, 32, window, out + i); continue; }     // a marker among them
        const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8);
        _mm256_storeu_si256((__m256i *)(out + i), pk);
    }
    resolve_scalar(sym + i, n - i, window, out + i);
}
#endif
inline void resolve_symbols(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { resolve_avx2(sym, n, window, out); return; }
#endif
    resolve_scalar(sym, n, window, out);
}

// ---------------------------------------------------------------------------------------------- the parallel reader
// Hands out the inflated stream as pieces in file order (pop), which the caller gives back when it has used them (recycle).
class ParGzip {
public:
    static constexpr size_t H = 32768;
    struct MemberEnd { size_t pos; uint32_t crc, isize; };                 // pos: bytes of the piece that belong to the member ending here
    struct Piece {                                                         // one resolved segment
        std::vector<uint8_t> bytes;
        size_t n = 0;
        bool eof = false;
        std::string err;
        // (the reader's own) lengths and CRC-32s of the stretches between member ends, and the trailers of the members that end in this piece
        std::vector<std::pair<size_t, uint32_t>> crc_parts;
        std::vector<MemberEnd> ends;
    };

    // max_decoding: segments being decoded or waiting to be resolved (each holds 2 bytes per inflated byte); max_pieces: resolved pieces the
    // consumer has not given back yet -- how far the reader may run ahead of it
    ParGzip(const uint8_t *data, size_t len, int threads, size_t seg_bytes, int max_decoding, int max_pieces)
        : d_(data), len_(len), seg_(seg_bytes < 65536 ? 65536 : seg_bytes), ahead_(max_decoding < 2 ? 2 : max_decoding), max_pieces_(max_pieces < 2 ? 2 : max_pieces)
    {
        n_seg_ = (len_ + seg_ - 1) / seg_;
        if (n_seg_ == 0) n_seg_ = 1;
        cap_sym_ = H + 8 * seg_ + ((size_t)16 << 20);                      // symbols a segment holds before it hands over what it has (flush_part)
        segs_.reset(new Seg[n_seg_]);
        for (size_t k = 0; k < n_seg_; k++) segs_[k].find_state = k == 0 ? 2 : 0;
        memset(win_prev_, 0, H);
        if (const char *e = getenv("MP_PARGZ_TEST_BOGUS_START")) test_bogus_every_ = (size_t)std::max(0, atoi(e));      // test hook (tests/test_host_logic.py): every n-th segment gets a false start
        running_crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
        for (int t = 0; t < (threads < 1 ? 1 : threads); t++) th_.emplace_back([this]() { worker(); });
    }
    ~ParGzip()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
    }
    // the next piece in file order (blocks until it is there); the caller hands it back with recycle()
    Piece *pop()
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !out_.empty(); });
        Piece *p = out_.front();
        out_.pop_front();
        return p;
    }
    void recycle(Piece *p) { { std::lock_guard<std::mutex> lk(mu_); free_pieces_.push_back(p); outstanding_--; } cv_.notify_all(); }
    // statistics for the log / tests
    size_t segments() const { return n_seg_; }
    size_t segments_started_blind() const { return blind_.load(); }
    size_t false_starts() const { return false_starts_.load(); }
    size_t parts_handed_over_early() const { return parts_.load(); }
    int segments_in_flight() { std::lock_guard<std::mutex> lk(mu_); return in_flight_; }     // 0 once the last piece has been popped
    // thread-seconds spent looking for starts / decoding / resolving (over all threads)
    double seconds_find() const { return ns_find_.load() * 1e-9; }
    double seconds_decode() const { return ns_decode_.load() * 1e-9; }
    double seconds_resolve() const { return ns_resolve_.load() * 1e-9; }

private:
    static constexpr uint64_t NONE = ~0ull;
    struct Seg {
        // where the segment's decoding starts (bit), found by whoever needs it first
        int find_state = 0;                         // 0 nobody looked, 1 somebody is looking, 2 known (start_bit, NONE: no start inside the segment)
        uint64_t start_bit = 0;
        // decoding
        bool claimed = false, decoded = false, dead = false;
        std::vector<uint16_t> sym;                  // [0, H) markers, then the symbols
        size_t n_sym = 0;                           // symbols after the markers
        std::vector<MemberEnd> ends;                // pos: symbols of this segment up to the member's end
        size_t next_seg = 0;                        // the segment whose start the decoder stopped on (n_seg_: the end of the file)
        bool eof = false;
        std::string err;
    };
    const uint8_t *d_;
    size_t len_, seg_, n_seg_, cap_sym_ = 0;
    int ahead_, max_pieces_;
    std::unique_ptr<Seg[]> segs_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    size_t next_decode_ = 0;                        // the next segment nobody has claimed
    size_t test_bogus_every_ = 0;
    size_t chain_ = 0;                              // the next segment of the chain to be resolved (its predecessor's window is known)
    bool chain_busy_ = false, chain_done_ = false;
    uint8_t win_prev_[H];                           // the 32 KiB in front of segment chain_
    int in_flight_ = 0, outstanding_ = 0;           // claimed and not yet resolved (or dead); pieces with the consumer or queued for it
    uint64_t order_next_ = 0;                       // pieces leave in chain order
    std::deque<Piece *> out_, free_pieces_;
    std::vector<std::unique_ptr<Piece>> all_pieces_;
    std::deque<std::pair<uint64_t, Piece *>> finished_;  // resolved out of order
    // member check, in delivery order
    uint32_t running_crc_ = 0, running_size_ = 0;
    std::atomic<size_t> blind_{0}, false_starts_{0}, parts_{0}, none_found_{0}, starts_found_{0};
    std::atomic<int> searches_given_up_{0};
    std::atomic<uint64_t> ns_find_{0}, ns_decode_{0}, ns_resolve_{0};
    static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // ---- finding a start inside segment k (k >= 1): first bit >= 8 * k * seg_ and < 8 * (k + 1) * seg_ that looks like a block and decodes like one
    uint64_t find_start(size_t k)
    {
        const uint64_t lo = (uint64_t)k * seg_ * 8u;
        uint64_t hi = (uint64_t)std::min(len_, (k + 1) * seg_) * 8u;
        if (len_ < 64) return NONE;
        const uint64_t safe = (uint64_t)(len_ - 40) * 8u;                   // the look-ahead of the checks stays inside the file
        if (hi > safe) hi = safe;
        if (searches_given_up_.load()) return NONE;
        if (test_bogus_every_ && k % test_bogus_every_ == test_bogus_every_ - 1 && lo + 3 < hi) return lo + 3;     // test hook: a start that is none
        BitPeek in{d_, len_};
        std::vector<uint16_t> trial;
        uint64_t word = 0;
        for (uint64_t b = lo; b < hi; b++) {
            // BFINAL = 0, BTYPE = 10 first: eight positions per load
            if ((b & 7u) == 0 || b == lo) word = in.at(b & ~(uint64_t)7);
            if (((word >> (b & 7u)) & 7u) != 4u) continue;
            if (!plausible_dynamic_header(in, b)) continue;
            // decode this block and the next one: no error, text only
            if (trial.empty()) { trial.resize(H + (1u << 20)); for (size_t i = 0; i < H; i++) trial[i] = (uint16_t)(0x8000u | i); }
            Inflater z;
            z.init_at_block(d_, len_, b);
            z.stop_at_blocks(true);
            size_t pos = H;
            int blocks = 0;
            bool ok = true;
            for (;;) {
                const int rc = z.run(trial.data(), pos, trial.size());
                if (rc == AT_BLOCK) { if (++blocks == 3) break; continue; }          // (the first stop is the start itself)
                if (rc == NEED_OUTPUT) { if (trial.size() > (64u << 20)) { ok = false; break; } trial.resize(trial.size() * 2); continue; }
                if (rc == MEMBER_END || rc == STREAM_END) break;                      // the block after this one ended the member: fine
                ok = false;
                break;
            }
            if (!ok) continue;
            for (size_t i = H; i < pos && ok; i++) {
                const uint16_t s = trial[i];
                if (s & 0x8000u) continue;
                ok = (s >= 32 && s < 127) || s == '\n' || s == '\r' || s == '\t';
            }
            if (ok && pos > H) return b;
        }
        return NONE;
    }
    // the start of segment k, looked for by the caller unless somebody else already has or is (then waited for)
    uint64_t ensure_start(size_t k)
    {
        std::unique_lock<std::mutex> lk(mu_);
        Seg &s = segs_[k];
        if (s.find_state == 0) {
            s.find_state = 1;
            lk.unlock();
            const uint64_t t0 = now_ns();
            const uint64_t b = find_start(k);
            ns_find_ += now_ns() - t0;
            // a file in which the first few segments hold no start at all (stored blocks, input that is not text) is not searched any further:
            // scanning every bit costs more than decoding, and one thread decodes such a file anyway
            if (b == NONE) { if (++none_found_ >= 3 && starts_found_.load() == 0) searches_given_up_ = 1; } else starts_found_++;
            lk.lock();
            s.start_bit = b; s.find_state = 2;
            cv_.notify_all();
        } else cv_.wait(lk, [&] { return s.find_state == 2 || stop_; });
        return s.start_bit;
    }

    void decode_segment(size_t k)
    {
        Seg &s = segs_[k];
        Inflater z;
        if (k == 0) z.init(d_, len_);
        else {
            const uint64_t st = ensure_start(k);
            if (st == NONE) { finish_decode(k, true); return; }             // nothing to start from: the segment before runs through this one
            z.init_at_block(d_, len_, st);
            blind_++;
        }
        z.stop_at_blocks(true);
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!sym_pool_.empty()) { s.sym.swap(sym_pool_.back()); sym_pool_.pop_back(); }
        }
        if (s.sym.size() < H + (1u << 20)) s.sym.resize(H + seg_ * 5 + (1u << 20));
        for (size_t i = 0; i < H; i++) s.sym[i] = (uint16_t)(0x8000u | i);
        s.ends.clear(); s.err.clear(); s.eof = false; s.next_seg = n_seg_;
        size_t pos = H;
        const uint64_t own_end = (uint64_t)std::min(len_, (k + 1) * seg_) * 8u;
        size_t cand = k + 1;                                                // the next segment whose start could end this one
        bool first = true;
        for (;;) {
            { std::lock_guard<std::mutex> lk(mu_); if (stop_ || s.dead) break; }
            const int rc = z.run(s.sym.data(), pos, s.sym.size());
            if (rc == NEED_OUTPUT) {
                if (s.sym.size() < cap_sym_) { s.sym.resize(std::min(cap_sym_, s.sym.size() + s.sym.size() / 2)); continue; }
                // a segment that inflates to more than eight times its size (or runs through many segments without a start): what it has
                // is handed over as a piece of its own as soon as the segment heads the chain, and the buffer is used again
                if (!flush_part(k, pos)) break;
                continue;
            }
            if (rc == MEMBER_END) { s.ends.push_back(MemberEnd{pos - H, z.member_crc(), z.member_isize()}); continue; }
            if (rc == STREAM_END) { s.eof = true; break; }
            if (rc == ERR) { s.err = z.error(); break; }
            // AT_BLOCK
            if (first) { first = false; continue; }
            const uint64_t b = z.bit_position();
            if (b < own_end) continue;
            bool stop_here = false;
            while (cand < n_seg_) {
                const uint64_t cand_lo = (uint64_t)cand * seg_ * 8u;
                if (b < cand_lo) break;                                     // not yet inside that segment
                const uint64_t c = ensure_start(cand);
                if (c == b) { s.next_seg = cand; stop_here = true; break; }
                if (c == NONE || c < b) {
                    // no start in that segment, or one this decoder has passed without standing on it: a false one -- if THIS decoder is on
                    // the true chain, which only the chain knows (resolve_segment declares the segments it jumps over dead).  A decoder cannot tell
                    // whether it or the starts it passes are off the rails, so passing them is never an error here: a decoder that is off the rails
                    // is declared dead by the chain (checked at every block) or runs into a real decoding error, and one the chain reaches is on
                    // the true chain by construction and decodes on, as zlib would.
                    cand++;
                    continue;
                }
                break;                                                      // its start lies ahead
            }
            if (stop_here) break;
        }
        s.n_sym = pos - H;
        finish_decode(k, false);
    }
    void finish_decode(size_t k, bool no_start)
    {
        std::lock_guard<std::mutex> lk(mu_);
        Seg &s = segs_[k];
        if (no_start) s.dead = true;
        s.decoded = true;
        cv_.notify_all();
    }

    // symbols -> bytes of a piece, the CRC-32 of its stretches between member ends
    void fill_piece(Piece *piece, const uint16_t *sym, size_t n, const uint8_t *window, const std::vector<MemberEnd> &ends)
    {
        if (piece->bytes.size() < n + 64) piece->bytes.resize(n + 64);
        resolve_symbols(sym, n, window, piece->bytes.data());
        piece->n = n;
        piece->crc_parts.clear(); piece->ends.clear();
        size_t from = 0;
        for (const MemberEnd &e : ends) {
            piece->crc_parts.emplace_back(e.pos - from, fast_crc32(0u, piece->bytes.data() + from, e.pos - from));
            piece->ends.push_back(e);
            from = e.pos;
        }
        piece->crc_parts.emplace_back(n - from, fast_crc32(0u, piece->bytes.data() + from, n - from));
    }
    // hands a finished piece over in order (mu_ held); the member checks run here, in order, on the combined CRCs
    void deliver_locked(Piece *piece, uint64_t order)
    {
        finished_.emplace_back(order, piece);
        for (bool moved = true; moved;) {
            moved = false;
            for (size_t i = 0; i < finished_.size(); i++)
                if (finished_[i].first == order_next_) {
                    Piece *p = finished_[i].second;
                    finished_.erase(finished_.begin() + (long)i);
                    check_members(p);
                    out_.push_back(p);
                    order_next_++;
                    moved = true;
                    break;
                }
        }
    }
    // a job of the chain, if there is one (mu_ held): the segment at its head is decoded and a piece may be taken
    bool claim_resolve(size_t &k, Piece *&piece, uint64_t &order)
    {
        if (chain_done_ || chain_busy_ || chain_ >= n_seg_ || !segs_[chain_].decoded || segs_[chain_].dead || outstanding_ >= max_pieces_) return false;
        k = chain_; chain_busy_ = true;
        piece = take_piece_locked();
        order = order_seq_++;
        return true;
    }
    Piece *take_piece_locked()
    {
        Piece *piece;
        if (!free_pieces_.empty()) { piece = free_pieces_.front(); free_pieces_.pop_front(); }
        else { all_pieces_.emplace_back(new Piece()); piece = all_pieces_.back().get(); }
        outstanding_++;
        return piece;
    }

    // Segment k's buffer is full: wait until k heads the chain (doing the chain's work meanwhile), hand over the symbols so far as a piece,
    // keep the last 32 Ki symbols as the history of what follows.  The markers keep referring to the window in front of the SEGMENT, which
    // does not change before the segment is done.  false: the reader is shutting down or the segment is dead.
    bool flush_part(size_t k, size_t &pos)
    {
        Seg &s = segs_[k];
        Piece *piece = nullptr;
        uint64_t order = 0;
        {
            std::unique_lock<std::mutex> lk(mu_);
            for (;;) {
                if (stop_ || s.dead) return false;
                if (chain_ == k && !chain_busy_ && outstanding_ < max_pieces_) break;
                size_t rk; Piece *rp; uint64_t ro;
                if (claim_resolve(rk, rp, ro)) { lk.unlock(); resolve_segment(rk, rp, ro); lk.lock(); continue; }
                cv_.wait(lk);
            }
            chain_busy_ = true;
            piece = take_piece_locked();
            order = order_seq_++;
        }
        fill_piece(piece, s.sym.data() + H, pos - H, win_prev_, s.ends);     // (win_prev_ is written by the head of the chain only: this thread)
        piece->eof = false; piece->err.clear();
        s.ends.clear();
        memmove(s.sym.data(), s.sym.data() + pos - H, H * sizeof(uint16_t));
        pos = H;
        std::lock_guard<std::mutex> lk(mu_);
        deliver_locked(piece, order);
        chain_busy_ = false;
        parts_++;
        cv_.notify_all();
        return true;
    }

    // the chain: segment `k` is the next one in file order, win_prev_ the 32 KiB in front of it
    void resolve_segment(size_t k, Piece *piece, uint64_t order)
    {
        Seg &s = segs_[k];
        uint8_t window[H];
        memcpy(window, win_prev_, H);
        // the window of what follows first, so that the chain can move on while the bulk of this segment is resolved: the last 32 Ki
        // symbols of the buffer, history included (a fresh segment's history are the markers themselves)
        {
            uint8_t next_win[H];
            resolve_symbols(s.sym.data() + s.n_sym, H, window, next_win);
            std::lock_guard<std::mutex> lk(mu_);
            memcpy(win_prev_, next_win, H);
            // the segments between this one and the one its decoder stopped on never enter the chain: whatever they decoded is void
            for (size_t i = k + 1; i < s.next_seg && i < n_seg_; i++) {
                Seg &f = segs_[i];
                if (f.find_state == 2 && f.start_bit != NONE) false_starts_++;
                f.dead = true;
                // its decoder may be done already and past its own "am I dead" check (worker()): then nobody else gives the slot and the buffer back.
                // `claimed` is the token: whoever clears it releases, the other side finds it cleared.
                if (f.claimed && f.decoded) release_dead_locked(f);
            }
            chain_ = s.next_seg;
            if (s.eof || !s.err.empty() || s.next_seg >= n_seg_) chain_done_ = true;
            chain_busy_ = false;
            cv_.notify_all();
        }
        fill_piece(piece, s.sym.data() + H, s.n_sym, window, s.ends);
        piece->eof = s.eof || s.next_seg >= n_seg_;
        piece->err = s.err;
        if (piece->err.empty() && !s.eof && s.next_seg >= n_seg_) piece->err = "truncated deflate stream";     // (the last segment ends with the stream or not at all)
        std::lock_guard<std::mutex> lk(mu_);
        deliver_locked(piece, order);
        in_flight_--;
        s.claimed = false; s.decoded = false;
        sym_pool_.emplace_back();
        sym_pool_.back().swap(s.sym);
        cv_.notify_all();
    }
    // a dead segment's slot and buffer are free again (mu_ held)
    void release_dead_locked(Seg &f)
    {
        in_flight_--; f.claimed = false; f.decoded = false;
        if (!f.sym.empty()) { sym_pool_.emplace_back(); sym_pool_.back().swap(f.sym); }
        cv_.notify_all();
    }
    void check_members(Piece *p)
    {
        for (size_t i = 0; i < p->crc_parts.size(); i++) {
            running_crc_ = (uint32_t)crc32_combine(running_crc_, p->crc_parts[i].second, (z_off_t)p->crc_parts[i].first);
            running_size_ += (uint32_t)p->crc_parts[i].first;
            if (i < p->ends.size()) {
                if (p->err.empty() && (p->ends[i].crc != running_crc_ || p->ends[i].isize != running_size_)) p->err = "incorrect data check (gzip CRC-32 or length)";
                running_crc_ = (uint32_t)crc32(0L, Z_NULL, 0); running_size_ = 0;
            }
        }
    }
    std::vector<std::vector<uint16_t>> sym_pool_;

    void worker()
    {
        for (;;) {
            size_t k = 0;
            int job = 0;                                                    // 1 resolve segment k, 2 decode segment k
            Piece *piece = nullptr;
            uint64_t order = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    if (stop_) return;
                    // dead segments at the head of the chain cannot occur (the chain only ever points at segments a decoder stopped on), but a
                    // dead segment that finished decoding gives its slot back
                    if (claim_resolve(k, piece, order)) { job = 1; break; }
                    if (next_decode_ < n_seg_ && in_flight_ < ahead_ && !chain_done_) {
                        k = next_decode_++;
                        if (segs_[k].dead) continue;                        // already known to be run through by the segment before it
                        segs_[k].claimed = true; job = 2; in_flight_++;
                        break;
                    }
                    cv_.wait(lk);
                }
            }
            const uint64_t t0 = now_ns();
            if (job == 1) { resolve_segment(k, piece, order); ns_resolve_ += now_ns() - t0; }
            else {
                decode_segment(k);
                ns_decode_ += now_ns() - t0;
                std::lock_guard<std::mutex> lk(mu_);
                if (segs_[k].dead && segs_[k].claimed) release_dead_locked(segs_[k]);   // nothing of it enters the chain (unless the chain has released it already)
            }
        }
    }
    uint64_t order_seq_ = 0;
};

}  // namespace mpz
