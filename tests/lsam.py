"""Test-side restatement of cc/fastq2lsam.cpp (reference /cc/fastq2lsam.cpp:28-112): annotated interleaved FASTQ -> LSAM.
Pinned against the real tool's output in tests/golden (test_host_logic.py); used to check the driver's -lsam mode."""


def _split(s, d):
    # misc.h splitBy: no trailing empty element
    out, i = [], 0
    while i < len(s):
        j = s.find(d, i)
        if j < 0:
            j = len(s)
        out.append(s[i:j])
        i = j + 1
    return out


def lsam_line(name, comm, seq, qual, which_end, output_seq=True):
    ignore = comm == b"IGNORE"
    try:
        score = -1 if ignore else int(_leading_int(comm[6:]))
    except ValueError:
        score = 0
    f = [name, {1: b"64", 2: b"128"}.get(which_end, b"0"), str(score).encode()]
    f += [seq, qual] if output_seq else [b"*", b"*"]
    if score <= 0:
        f.append(b"*")
    else:
        parts = []
        for el in _split(comm, b";")[1:]:
            sub = _split(el, b",")
            parts += [sub[0] + b"," + x for x in sub[1:]]
        f.append(b";".join(parts))
    if ignore:
        f.append(b"IGNORE")
    return b"\t".join(f) + b"\n"


def _leading_int(b):
    # atoi: optional sign and digits at the start, 0 when none
    i = 0
    while i < len(b) and (b[i:i + 1].isdigit() or (i == 0 and b[i:i + 1] in b"+-")):
        i += 1
    return b[:i] if i and b[:i] not in (b"+", b"-") else b"0"


def fastq_to_lsam(data: bytes, output_seq=True) -> bytes:
    lines = data.split(b"\n")
    recs = []
    for i in range(0, len(lines) - 3, 4):
        h = lines[i][1:]
        cut = len(h)
        for k, ch in enumerate(h):
            if ch in b" \t":
                cut = k
                break
        name, comm = h[:cut], h[cut + 1:]
        if len(name) > 2 and name[-2:-1] == b"/" and name[-1:].isdigit():
            name = name[:-2]
        recs.append((name, comm, lines[i + 1], lines[i + 3]))
    out, held = [], None
    for r in recs:
        if held is not None:
            if held[0] == r[0]:
                out.append(lsam_line(*held, 1, output_seq))
                out.append(lsam_line(*r, 2, output_seq))
                held = None
                continue
            out.append(lsam_line(*held, 0, output_seq))
        held = r
    if held is not None:
        out.append(lsam_line(*held, 0, output_seq))
    return b"".join(out)
