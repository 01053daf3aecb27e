import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def g1_prefix():
    return os.path.join(GOLDEN, "g1", "ref.fa.index")


def fastq_records(path):
    """annotated FASTQ -> sorted list of (name, mate#, header, seq, qual); order-insensitive comparison (App. D12)."""
    with open(path) as f:
        lines = f.read().split("\n")
    out = []
    seen = {}
    i = 0
    while i + 3 < len(lines):
        h = lines[i]
        name = h[1:].split("\t")[0]
        seen[name] = seen.get(name, 0) + 1
        out.append((name, seen[name], h, lines[i + 1], lines[i + 3]))
        i += 4
    return sorted(out)


def bam_golden(path):
    """Parses tests/golden/**/expected_*.bam.txt -> {"header": str, "dpout": [...], "gout": [...], "unpair": [...]}."""
    out = {"header": "", "dpout": [], "gout": [], "unpair": []}
    cur = None
    for line in open(path).read().splitlines():
        if line.startswith("#") and line[1:] in out:
            cur = line[1:]
        elif cur == "header":
            out["header"] += line + "\n"
        else:
            out[cur].append(line)
    return out
