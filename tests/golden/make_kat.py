"""Generate tests/golden/kat.json: formula-derived known-answer values for the
densities on PClean's hot path.  Computed with scipy /
math from the formulas in the cited reference files — NOT outputs of the Julia
reference (which cannot run in this image; parity is unpinned).

Run from the repo root:  python tests/golden/make_kat.py
"""
import json
import math
import os
import sys

import numpy as np
from scipy import stats

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pclean_amd.encode import load_lm_params, ALPHABET  # data loader only


def simple_osa(a, b):
    la, lb = len(a), len(b)
    d = [[0] * (lb + 1) for _ in range(la + 1)]
    for i in range(la + 1):
        d[i][0] = i
    for j in range(lb + 1):
        d[0][j] = j
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            c = 0 if a[i - 1] == b[j - 1] else 1
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + c)
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                d[i][j] = min(d[i][j], d[i - 2][j - 2] + 1)
    return d[la][lb]


def add_typos(obs, word, max_typos=None):
    # src/distributions/add_typos.jl:50-66
    d = simple_osa(obs, word)
    if max_typos is not None and d > max_typos:
        return -1e5
    r = math.ceil(len(word) / 5.0)
    l = stats.nbinom.logpmf(d, r, 0.9)
    l -= math.log(len(word)) * d
    l -= math.log(26) * d / 2
    return float(l)


def string_prior(s, lo, hi):
    # src/distributions/string_prior.jl:43-61
    init, trans = load_lm_params()
    if len(s) < lo or len(s) > hi:
        return float("-inf")
    score = -math.log(hi - lo + 1)
    prev = None
    idx = {c: i for i, c in enumerate(ALPHABET)}
    for ch in s:
        dist = init if prev is None else trans[prev]
        prev = idx.get(ch.lower())
        if prev is None:
            score += -math.log(28)
        else:
            p = dist[prev]
            score += max(math.log(p), -1000) if p > 0 else -1000
    return float(score)


kat = {
    "add_typos": [
        {"obs": o, "word": w, "max_typos": m, "value": add_typos(o, w, m)}
        for (o, w, m) in [("abc", "abc", None), ("abd", "abc", None), ("birmingham", "birmingham", None),
                          ("birminghxm", "birmingham", None), ("bxrmxngham", "birmingham", None),
                          ("al", "ak", None), ("ab", "ba", None), ("mahoning county", "mahoning cnty", 2),
                          ("clark county", "mahoning county", 2)]
    ],
    "string_prior": [
        {"s": s, "min": lo, "max": hi, "value": string_prior(s, lo, hi)}
        for (s, lo, hi) in [("birmingham", 3, 30), ("al", 3, 30), ("2053258100", 10, 10), ("the cat.", 3, 30),
                            ("a", 1, 1), ("Birmingham", 3, 30)]
    ],
    "negbin": [{"r": r, "p": 0.9, "k": k, "value": float(stats.nbinom.logpmf(k, r, 0.9))}
               for r in (1, 2, 3, 7, 37) for k in (0, 1, 2, 5, 17, 60)],
    "normal": [{"x": 1267.0, "mu": 1500.0, "sigma": 150.0, "value": float(stats.norm.logpdf(1267.0, 1500.0, 150.0))}],
    # transformed_gaussian.jl:15-16 with the unit-2 transformation of experiments/rents/run.jl:6
    "transformed_gaussian": [{"obs": 1.267, "mu": 1500.0, "sigma": 150.0, "backward": 1267.0, "abs_deriv": 1 / 1000.0,
                              "value": float(stats.norm.logpdf(1267.0, 1500.0, 150.0) - math.log(1 / 1000.0))}],
    # maybe_swap.jl:13-28
    "maybe_swap": [{"same": 1, "n": 4, "p": 0.1, "value": math.log1p(-0.1)},
                   {"same": 0, "n": 4, "p": 0.1, "value": math.log(0.1) - math.log(4)}],
    "time_prior": -math.log(1440.0),
    "time_regex": {"7:10 a.m.": True, "10:30 p.m.": True, "**:** p.m.": False, "7:1 a.m.": False,
                   "123:10 a.m.": False, "7:10 a.m": False, "7:10 x.m.": False},
}
out = os.path.join(os.path.dirname(__file__), "kat.json")
with open(out, "w") as f:
    json.dump(kat, f, indent=1, allow_nan=True)
print("wrote", out)
