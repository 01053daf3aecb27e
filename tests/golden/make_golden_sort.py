#!/usr/bin/env python3
"""Regenerates tests/golden/sort_vectors.npz: permutations that std::sort of THIS container's libstdc++ (g++ 11, the
library the reference binary of oracle/Makefile.ref is built against) applies to arrays of seed lengths under the
comparator of singleMerge (LongerSeedLength: a.len > b.len, DV-DPfunctions.cpp:289-293,334).  Data only: lengths in,
permutation out.  Complements tests/golden/g2 (set "t"), where the same order is pinned through the reference itself."""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = r'''
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
struct E { uint32_t id, len; };
int main() {
    int n_arrays; if (scanf("%d", &n_arrays) != 1) return 1;
    for (int a = 0; a < n_arrays; a++) {
        int n; if (scanf("%d", &n) != 1) return 1;
        std::vector<E> v(n);
        for (int i = 0; i < n; i++) { v[i].id = i; if (scanf("%u", &v[i].len) != 1) return 1; }
        std::sort(v.begin(), v.end(), [](const E &x, const E &y) { return x.len > y.len; });
        for (int i = 0; i < n; i++) printf("%u ", v[i].id);
        printf("\n");
    }
}
'''


def main():
    rng = np.random.Generator(np.random.PCG64(77))
    arrays = []
    for n in [0, 1, 2, 15, 16, 17, 18, 31, 32, 33, 64, 100, 199, 200, 201, 217, 279, 341, 512, 1000]:
        for kinds in (1, 2, 3, 7):
            arrays.append((17 + rng.integers(0, kinds, size=n)).astype(np.uint32))
    for n in (40, 300, 777):                      # patterns quicksort dislikes: sorted, reversed, organ pipe, saw
        i = np.arange(n)
        arrays += [(17 + i).astype(np.uint32), (17 + n - i).astype(np.uint32), (17 + np.minimum(i, n - i)).astype(np.uint32), (17 + i % 5).astype(np.uint32)]
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "s.cpp"), os.path.join(td, "s")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-O2", "-o", exe, src])
        inp = "%d\n" % len(arrays) + "".join("%d %s\n" % (len(a), " ".join(map(str, a))) for a in arrays)
        out = subprocess.run([exe], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    perms = [np.array(line.split(), dtype=np.int32) for line in out[:len(arrays)]]
    off = np.concatenate([[0], np.cumsum([len(a) for a in arrays])]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "sort_vectors.npz"), seed_len=np.concatenate(arrays), perm=np.concatenate(perms), off=off)
    print("sort_vectors.npz:", len(arrays), "arrays,", int(off[-1]), "elements")


if __name__ == "__main__":
    main()
