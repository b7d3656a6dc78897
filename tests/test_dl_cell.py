"""The linear-space unrestricted Damerau-Levenshtein recurrence of the pair-table kernel (pclean_amd/csrc/dl_cell.h, used by
dist_kernels.hip: dl_seg_kernel) — host build tests/dl_host, driven through the kernel's own schedule (NSEG lanes per pair,
lane s one row behind lane s-1, packed row state handed on) — against the oracle's full-matrix Lowrance-Wagner DP
(oracle/densities.h: dl_distance = StringDistances' DamerauLevenshtein, add_typos.jl:56).  CPU only: the GPU tests hold the
kernel itself against the oracle (tests/test_gpu_tables.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "dl_host", "dl_host.cpp")
LIB = os.path.join(HERE, "dl_host", "libdl_host.so")
DEPS = [SRC, os.path.join(HERE, "..", "pclean_amd", "csrc", "dl_cell.h"), os.path.join(HERE, "..", "oracle", "densities.h")]


@pytest.fixture(scope="module")
def dlh():
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    L = C.CDLL(LIB)
    L.dlh_fuzz.restype = C.c_long
    L.dlh_fuzz.argtypes = [C.c_uint64, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
    for f in (L.dlh_distance, L.dlh_reference, L.dlh_osa):
        f.restype = C.c_int
    return L


def _arr(s):
    a = np.array([ord(c) for c in s], dtype=np.uint16)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint16))


@pytest.mark.parametrize("nseg", [1, 2, 4, 8])
def test_linear_space_dl_equals_full_matrix_dp_on_random_pairs(dlh, nseg):
    """random and mutated pairs (substitutions, insertions, deletions, adjacent and gapped transpositions) over small and
    large alphabets, lengths 0 .. 254; thousands of them have an unrestricted distance below the restricted one (the two
    transposition terms at work)"""
    n_diff = 0
    for alpha, max_len, n in ((3, 20, 60000), (4, 30, 40000), (10, 12, 60000), (26, 40, 20000), (3, 100, 4000), (5, 254, 600)):
        d = C.c_long(0)
        bad = dlh.dlh_fuzz(alpha * 1000 + nseg, n, alpha, max_len, nseg, C.byref(d))
        assert bad == 0, (alpha, max_len, nseg, bad)
        n_diff += d.value
    assert n_diff > 3000  # (the fuzz really reaches pairs where DL < OSA)


def test_linear_space_dl_known_cases(dlh):
    cases = [("ca", "abc", 2), ("", "", 0), ("", "abc", 3), ("abc", "", 3), ("abcdef", "abcdef", 0), ("ab", "ba", 1),
             ("abcd", "acbd", 1), ("birmingham", "birmingahm", 1), ("a" * 254, "a" * 253 + "b", 1), ("xaby", "xbya", 2)]
    for a, b, want in cases:
        aa, ap = _arr(a)
        bb, bp = _arr(b)
        assert dlh.dlh_reference(ap, len(a), bp, len(b)) == want, (a, b)
        for nseg in (1, 2, 4, 8):
            for cap in (0, 5):
                assert dlh.dlh_distance(ap, len(a), bp, len(b), nseg, cap) == want, (a, b, nseg, cap)
