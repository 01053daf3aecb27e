"""`soap4`-compatible command line for the MI355X path (drop-in for ${SOAP4} in runMegaPath.sh:136,162,206,208).

    python -m megapath_amd.soap4 pair <indexPrefix> <r1.fq[.gz]> <r2.fq[.gz]> -o <outPrefix> -C <ini> -L <maxReadLen>
                                 [-T threads] [-u maxInsert] [-v minInsert] -F|-P [-nc] [-top N] [-b]

Flags follow parseInputArgs (reference soap4/IniParam.cpp:542-942); ini keys follow ParseIniFile (:242-438).
stdout carries the interleaved annotated FASTQ (MegaPath mode); stderr a free-form log.  This Python mirror exists for
tests and scripting; `-b` (BAM side files) and `-lsam` are implemented by the C++ executable bin/soap4-mi355x only --
here `-b` is accepted and reported.
"""
from __future__ import annotations

import configparser
import os
import sys
import time

from . import hipapi as H
from .pipeline import Soap4Params, align_fastq


def parse_ini(path: str, p: Soap4Params) -> None:
    """Only the keys that reach this path are read (SURVEY section 5 lists the parsed-but-ignored ones)."""
    cp = configparser.ConfigParser(strict=False, inline_comment_prefixes=(";", "#"))
    cp.optionxform = str
    with open(path) as f:
        cp.read_file(f)

    def geti(sec, key, dflt):
        try:
            return int(cp.get(sec, key))
        except Exception:  # noqa: BLE001
            return dflt

    def getf(sec, key, dflt):
        try:
            return float(cp.get(sec, key))
        except Exception:  # noqa: BLE001
            return dflt

    if geti("DP", "MatchScore", 1) != 1 or geti("DP", "MismatchScore", -2) != -2 or geti("DP", "GapOpenScore", -3) != -3 \
            or geti("DP", "GapExtendScore", -1) != -1:
        raise SystemExit("only the shipped DP scores (+1/-2/-3/-1) are compiled into the kernels")
    if cp.get("PairEnd", "StrandArrangement", fallback="+/-").strip() != "+/-":
        raise SystemExit("only StrandArrangement=+/- is supported")
    if geti("OtherSettings", "SkipSOAP3Alignment", 1) != 1:
        raise SystemExit("SkipSOAP3Alignment must be 1 (the reference asserts the same, alignment.cpp:50-67)")
    p.clip_left = geti("Clipping", "MaxFrontLenClipped", p.clip_left)
    p.clip_right = geti("Clipping", "MaxEndLenClipped", p.clip_right)
    p.skip_default_dp = geti("OtherSettings", "SkipDefaultDP", 0) != 0
    m = p.mmp
    m.sa_size_threshold = geti("MMP", "mmpSeedSAsizeThreshold", 30)
    m.seed_min_length = geti("MMP", "mmpSeedMinLength", 17)
    m.uniq_threshold = geti("MMP", "mmpUniqThreshold", 6)
    m.indel_fuzz = geti("MMP", "mmpIndelFuzz", 5)
    m.good_seed_len = geti("MMP", "mmpGoodSeedLen", 27)
    m.reseed_len = geti("MMP", "mmpReseedLen", 18)
    m.reseed_rlt_ratio = getf("MMP", "mmpReseedRLTratio", 0.85)
    m.reseed_abs_diff = geti("MMP", "mmpReseedAbsDiff", 4)
    m.short_seed_ratio = getf("MMP", "mmpShortSeedRatio", 0.5)


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 4 or argv[0] != "pair":
        sys.stderr.write(__doc__)
        return 1
    prefix, fq1, fq2 = argv[1:4]
    p = Soap4Params()
    ini = None
    device = int(os.environ.get("MP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    want_bam = False
    mode = 0
    i = 4
    while i < len(argv):
        a = argv[i]
        if a in ("-o", "-T", "-D", "-A", "-R", "-e"):   # arities follow parseInputArgs (IniParam.cpp:651-896)
            i += 1                                 # value consumed, not used on this path
        elif a == "-h":
            i += 1
            if not (1 <= int(argv[i]) <= 4):
                sys.stderr.write("The output option should be 1, 2, 3 or 4\n")
                return 1
        elif a == "-s":
            if i + 1 < len(argv) and argv[i + 1][:1].isdigit():
                i += 1
        elif a == "-c":
            i += 1
            device = int(argv[i])                  # GPU device id (IniParam.cpp:822-838)
        elif a in ("-p", "-I"):
            pass
        elif a == "-C":
            i += 1
            ini = argv[i]
        elif a in ("-L", "-l"):
            i += 1
            p.max_read_length = int(argv[i])
        elif a == "-u":
            i += 1
            p.insert_high = int(argv[i])
        elif a == "-v":
            i += 1
            p.insert_low = int(argv[i])
        elif a == "-top":
            i += 1
            p.top_percentage = int(argv[i]) / 100.0
        elif a == "-F":
            mode = 1
        elif a == "-P":
            mode = 2
        elif a == "-nc":
            p.ignore_comments = True               # drop FASTA/Q comments before alignment (SOAP4.cpp:537-547)
        elif a == "-b":
            want_bam = True
        i += 1
    if p.insert_low > p.insert_high:
        sys.stderr.write("The minimum value of insert size should not be greater than the maximum value of insert size.\n")
        return 1
    if mode == 0:
        sys.stderr.write("[soap4-mi355x] only MegaPath modes -F / -P are implemented (plain mode segfaults in the reference, SURVEY App. D15)\n")
        return 1
    p.megapath_mode = mode
    if ini is None:
        ini = os.path.join(os.path.dirname(os.path.abspath(sys.argv[0])), "soap4.ini")
    parse_ini(ini, p)
    if want_bam:
        sys.stderr.write("[soap4-mi355x] -b: the Python mirror writes no BAM files; use bin/soap4-mi355x (the default of bin/soap4)\n")
    t0 = time.time()
    index = H.Index(prefix, device)
    t1 = time.time()
    sys.stderr.write("[Main] index %s loaded into HBM in %.2f s (%.2f GB)\n" % (prefix, t1 - t0, index.info.device_bytes / 1e9))
    st = align_fastq(index, p, fq1, fq2, sys.stdout.buffer)
    sys.stdout.buffer.flush()
    t2 = time.time()
    sys.stderr.write("[Main] %d pairs: %d aligned by deep DP, %d by mate rescue, %d unpaired; %d DP instances; alignment time %.2f s\n"
                     % (st.n_pairs, st.n_deep_pairs, st.n_rescued_pairs, st.n_unpaired_pairs, st.n_dp_instances, t2 - t1))
    index.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
