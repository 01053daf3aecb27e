#!/bin/bash
# Sanitizer runs of the host code (CPU only; GPU ASan is not available on this pool): builds host/soap4_main.cpp with
# -fsanitize=thread and with -fsanitize=address,undefined and drives the reader -> workers -> printer plumbing (MP_PLUMBING_TEST:
# no device, nothing aligned) with several fake GPUs, jittered batch times, tiny batches, two chunks, gz input and a corrupt gz.
# Usage: bash scripts/sanitize_host.sh        (exit code 0 = both clean)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
B="$ROOT/build/sanitize"; mkdir -p "$B"
G1="$ROOT/tests/golden/g1"
SRC="$ROOT/megapath_amd/csrc/host/soap4_main.cpp"
fail=0
gzip -c "$G1/reads_a_1.fq" > "$B/a_1.fq.gz"; gzip -c "$G1/reads_a_2.fq" > "$B/a_2.fq.gz"
head -c 6000 "$B/a_1.fq.gz" > "$B/bad_1.fq.gz"
# shapes that leave the strict fast path of the reader: wrapped FASTA, blank lines + CRLF + lower case / N
python3 - "$G1" "$B" <<'PY'
import sys
g1, b = sys.argv[1], sys.argv[2]
for m in (1, 2):
    lines = open("%s/reads_a_%d.fq" % (g1, m)).read().split("\n")
    fa, odd = [], []
    for i in range(0, len(lines) - 3, 4):
        seq = lines[i + 1]
        fa.append(">" + lines[i][1:] + "\n" + "\n".join(seq[j:j + 60] for j in range(0, len(seq), 60)) + "\n")
        odd.append(lines[i] + "\r\n" + seq.lower().replace("a", "N", 1) + "\r\n+\r\n" + lines[i + 3] + "\r\n" + ("\r\n" if (i // 4) % 5 == 0 else ""))
    open("%s/w_%d.fa" % (b, m), "w").write("".join(fa))
    open("%s/odd_%d.fq" % (b, m), "w", newline="").write("".join(odd))
    # BGZF: members inflated by several threads, handed over in file order
    import struct, zlib
    data = open("%s/reads_a_%d.fq" % (g1, m), "rb").read()
    out = b""
    for i in list(range(0, len(data), 3000)) + [None]:
        raw = b"" if i is None else data[i:i + 3000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(raw) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(raw), len(raw))
    open("%s/bg_%d.fq.gz" % (b, m), "wb").write(out)
PY
for kind in thread address,undefined; do
    exe="$B/soap4-$(echo $kind | tr , -)"
    g++ -O1 -g -std=c++17 -Wall -pthread -fno-omit-frame-pointer -fsanitize=$kind -o "$exe" "$SRC" -L"$ROOT/megapath_amd" -lmegapath_hip -lz -Wl,-rpath,"$ROOT/megapath_amd" || { echo "build ($kind) failed"; exit 2; }
    export TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=1 exitcode=66" UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=0 exitcode=66"
    run() {   # name, expected rc, args...
        local name="$1" want="$2"; shift 2
        MP_PLUMBING_TEST=1 "$@" > "$B/$name.$kind.out" 2> "$B/$name.$kind.err"; local rc=$?
        if grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:\|ERROR: LeakSanitizer" "$B/$name.$kind.err"; then echo "[$kind] $name: SANITIZER REPORT (see $B/$name.$kind.err)"; fail=1;
        elif [ $rc -ne $want ]; then echo "[$kind] $name: exit code $rc, expected $want"; tail -5 "$B/$name.$kind.err"; fail=1;
        else echo "[$kind] $name: clean (rc $rc, $(grep -c '^@' "$B/$name.$kind.out") records)"; fi
    }
    common=(-o "$B/o" -C "$ROOT/tests/soap4-nt2.ini" -L 151 -u 750)
    MP_PLUMBING_JITTER_US=4000 MP_BATCH_PAIRS=7 run order 0 "$exe" pair x,y "$G1/reads_a_1.fq" "$G1/reads_a_2.fq" "${common[@]}" -T 3 -F -nc -c 0,1,2
    MP_PLUMBING_JITTER_US=500 MP_BATCH_PAIRS=3 MP_WORKERS_PER_GPU=4 run tiny 0 "$exe" pair x "$B/a_1.fq.gz" "$B/a_2.fq.gz" "${common[@]}" -T 8 -F -c 0,1
    MP_BATCH_PAIRS=1000 run onebatch 0 "$exe" pair x,y,z "$G1/reads_d_1.fq" "$G1/reads_d_2.fq" "${common[@]}" -T 2 -P
    MP_BATCH_PAIRS=11 run wrapped_fasta 0 "$exe" pair x "$B/w_1.fa" "$B/w_2.fa" "${common[@]}" -T 4 -F -nc
    MP_BATCH_PAIRS=13 run odd_shapes 0 "$exe" pair x "$B/odd_1.fq" "$B/odd_2.fq" "${common[@]}" -T 4 -F -nc
    MP_BATCH_PAIRS=9 MP_INFLATE_THREADS=4 run bgzf 0 "$exe" pair x "$B/bg_1.fq.gz" "$B/bg_2.fq.gz" "${common[@]}" -T 4 -F -nc
    MP_BATCH_PAIRS=9 MP_INFLATE_THREADS=4 MP_PARGZ_MIN=0 MP_PARGZ_SEG=65536 run pargz 0 "$exe" pair x "$B/a_1.fq.gz" "$B/a_2.fq.gz" "${common[@]}" -T 4 -F -nc
    TSAN_OPTIONS="$TSAN_OPTIONS report_thread_leaks=0" ASAN_OPTIONS="$ASAN_OPTIONS detect_leaks=0" MP_INFLATE_THREADS=4 MP_PARGZ_MIN=0 MP_PARGZ_SEG=65536 MP_BATCH_PAIRS=20 run pargz_corrupt 1 "$exe" pair x "$B/bad_1.fq.gz" "$G1/reads_a_2.fq" "${common[@]}" -T 2 -F -nc
    MP_BATCH_PAIRS=50 run lsam 0 "$exe" pair x "$G1/reads_e_1.fq" "$G1/reads_e_2.fq" "${common[@]}" -T 4 -F -top 95 -lsam
    # fatal() leaves through _exit() from whichever thread met the error, on purpose (no atexit handlers while other stages write):
    # the threads it leaves behind are not a leak to report
    TSAN_OPTIONS="$TSAN_OPTIONS report_thread_leaks=0" ASAN_OPTIONS="$ASAN_OPTIONS detect_leaks=0" MP_BATCH_PAIRS=20 run corrupt 1 "$exe" pair x "$B/bad_1.fq.gz" "$G1/reads_a_2.fq" "${common[@]}" -T 2 -F -nc
done
[ $fail -eq 0 ] && echo "sanitizers: all clean" || echo "sanitizers: REPORTS ABOVE"
exit $fail
