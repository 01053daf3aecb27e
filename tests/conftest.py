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
