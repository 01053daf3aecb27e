// tests/tools/pargz_check.cpp -- test tool: inflates a .gz file with the executable's sequential decoder (fast_inflate.h) and with the parallel
// reader (par_gzip.h) and compares both with zlib's gzread, byte for byte.  Prints rates and the parallel reader's statistics.
//   pargz_check <file.gz> <threads> <segment bytes> [repeats]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../megapath_amd/csrc/host/par_gzip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: pargz_check <file.gz> <threads> <segment bytes> [repeats]\n"); return 2; }
    const int threads = atoi(argv[2]);
    const size_t seg = (size_t)atoll(argv[3]);
    const int reps = argc > 4 ? atoi(argv[4]) : 1;
    const int fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { perror(argv[1]); return 2; }
    const uint8_t *d = (const uint8_t *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (d == MAP_FAILED) { perror("mmap"); return 2; }
    // zlib's reading
    std::vector<uint8_t> want;
    {
        const double t0 = now();
        gzFile f = gzopen(argv[1], "rb");
        gzbuffer(f, 1 << 20);
        std::vector<uint8_t> buf(1 << 22);
        for (;;) { const int n = gzread(f, buf.data(), (unsigned)buf.size()); if (n <= 0) break; want.insert(want.end(), buf.begin(), buf.begin() + n); }
        gzclose(f);
        printf("zlib       %8.1f MB/s (%zu bytes)\n", want.size() / (now() - t0) / 1e6, want.size());
    }
    int rc = 0;
    for (int r = 0; r < reps; r++) {
        // sequential decoder
        {
            const double t0 = now();
            mpz::Inflater z;
            z.init(d, (size_t)st.st_size);
            std::vector<uint8_t> out(want.size() + 65536 + 512);
            size_t pos = 0;
            int code;
            while ((code = z.run(out.data(), pos, out.size())) == mpz::MEMBER_END) {}
            const double dt = now() - t0;
            const bool same = code == mpz::STREAM_END && pos == want.size() && memcmp(out.data(), want.data(), pos) == 0;
            printf("sequential %8.1f MB/s %s\n", pos / dt / 1e6, same ? "equal" : "DIFFERENT");
            if (!same) rc = 1;
        }
        // parallel reader
        {
            const double t0 = now();
            mpz::ParGzip pg(d, (size_t)st.st_size, threads, seg, threads + 2, threads + 4);
            size_t pos = 0;
            bool same = true;
            std::string err;
            for (;;) {
                mpz::ParGzip::Piece *p = pg.pop();
                if (pos + p->n > want.size() || memcmp(p->bytes.data(), want.data() + pos, p->n) != 0) same = false;
                pos += p->n;
                const bool last = p->eof || !p->err.empty();
                if (!p->err.empty()) err = p->err;
                pg.recycle(p);
                if (last) break;
            }
            const double dt = now() - t0;
            printf("           thread-seconds: finding starts %.3f, decoding (incl. finding) %.3f, resolving %.3f; wall %.3f\n", pg.seconds_find(), pg.seconds_decode(), pg.seconds_resolve(), dt);
            if (pos != want.size() || !err.empty()) same = false;
            // every claimed segment has given its slot back (decoders of dead segments notice at their next block)
            int left = pg.segments_in_flight();
            for (int w = 0; w < 200 && left != 0; w++) { usleep(10000); left = pg.segments_in_flight(); }
            printf("           segments still holding a slot after the last piece: %d\n", left);
            if (left != 0) same = false;
            printf("parallel   %8.1f MB/s %s  (%d threads, %zu segments, %zu started without their window, %zu false starts, %zu early parts)%s%s\n", pos / dt / 1e6, same ? "equal" : "DIFFERENT",
                   threads, pg.segments(), pg.segments_started_blind(), pg.false_starts(), pg.parts_handed_over_early(), err.empty() ? "" : "  error: ", err.c_str());
            if (!same) rc = 1;
        }
    }
    return rc;
}
