"""Minimal BGZF/BAM reader for the tests: decodes the BAM files written by `soap4 -b` (reference and this repo's
driver alike) into SAM-like text lines so that records can be compared field by field.  No samtools needed.
Layout: SAM spec section 4 (the bundled samtools-0.1.x `bam_write1` writes the same records)."""
from __future__ import annotations

import struct
import zlib

_CIGAR_OPS = "MIDNSHP=X"
_SEQ = "=ACMGRSVTWYHKDBN"


def bgzf_decompress(path: str) -> bytes:
    raw = open(path, "rb").read()
    out = []
    while raw:
        d = zlib.decompressobj(31)
        out.append(d.decompress(raw))
        raw = d.unused_data
    return b"".join(out)


def _tags(buf: bytes) -> list[str]:
    i, out = 0, []
    while i < len(buf):
        tag = buf[i:i + 2].decode()
        t = chr(buf[i + 2])
        i += 3
        if t == "Z":
            j = buf.index(b"\0", i)
            out.append("%s:Z:%s" % (tag, buf[i:j].decode("latin1")))
            i = j + 1
        elif t == "A":
            out.append("%s:A:%s" % (tag, chr(buf[i])))
            i += 1
        else:
            fmt, n = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4), "f": ("<f", 4)}[t]
            out.append("%s:i:%s" % (tag, struct.unpack_from(fmt, buf, i)[0]))
            i += n
    return out


def read_bam(path: str):
    """Returns (header_text, [(name, length)], [record line])."""
    b = bgzf_decompress(path)
    assert b[:4] == b"BAM\1", "not a BAM file: %s" % path
    l_text, = struct.unpack_from("<i", b, 4)
    text = b[8:8 + l_text].rstrip(b"\0").decode("latin1")
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", b, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", b, p)
        name = b[p + 4:p + 4 + l_name - 1].decode("latin1")
        l_ref, = struct.unpack_from("<i", b, p + 4 + l_name)
        refs.append((name, l_ref))
        p += 8 + l_name
    recs = []
    while p < len(b):
        bs, = struct.unpack_from("<i", b, p)
        ref_id, pos, bin_mq_nl, flag_nc, l_seq, mref, mpos, tlen = struct.unpack_from("<iiIIiiii", b, p + 4)
        q = p + 36
        l_name = bin_mq_nl & 0xFF
        mapq = (bin_mq_nl >> 8) & 0xFF
        n_cig = flag_nc & 0xFFFF
        flag = flag_nc >> 16
        name = b[q:q + l_name - 1].decode("latin1")
        q += l_name
        cig = "".join("%d%s" % (c >> 4, _CIGAR_OPS[c & 15]) for c in struct.unpack_from("<%dI" % n_cig, b, q)) or "*"
        q += 4 * n_cig
        sq = b[q:q + (l_seq + 1) // 2]
        seq = "".join(_SEQ[(sq[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = "".join(chr(min(126, 33 + x)) if x < 93 else "~" for x in b[q:q + l_seq])
        qual_raw = ",".join(str(x) for x in b[q:q + l_seq][:3])          # first three raw values: pins the offset exactly
        q += l_seq
        tags = _tags(b[q:p + 4 + bs])
        recs.append("\t".join([name, str(flag), refs[ref_id][0] if ref_id >= 0 else "*", str(pos + 1), str(mapq), cig,
                               refs[mref][0] if mref >= 0 else "*", str(mpos + 1), str(tlen), seq, qual, "q0:" + qual_raw] + tags))
        p += 4 + bs
    return text, refs, recs
