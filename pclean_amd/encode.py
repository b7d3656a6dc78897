"""Dictionary encoding of string-valued columns for the HIP path.

Julia strings are indexed by code point (`length(word)`, add_typos.jl:61-62), so
every string is decoded to code points first; the edit-distance kernels only
need symbol equality, so code points are remapped to dense uint16 symbols.
The StringPrior alphabet index (string_prior.jl:11-12,55-56: a-z, ' ', '.',
after `lowercase`) is kept as a parallel uint8 array.
"""
import os

import numpy as np

ALPHABET = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [" ", "."]
_ALPHA_IDX = {c: i for i, c in enumerate(ALPHABET)}

_LM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lmparams")


def load_lm_params():
    """(init_p[28], trans_p[28][28]) with trans_p[prev][next]: the reference's
    english_letter_transitions[next, prev] (string_prior.jl:9-10,32,55)."""
    init = np.loadtxt(os.path.join(_LM_DIR, "letter_probabilities.csv"), delimiter=",").reshape(-1)
    mat = np.loadtxt(os.path.join(_LM_DIR, "letter_transition_matrix.csv"), delimiter=",")
    assert init.shape == (28,) and mat.shape == (28, 28)
    return init, np.ascontiguousarray(mat.T)


def lm_log_tables():
    """log tables floored at -UNUSUAL_LETTER_PENALTY (string_prior.jl:41,56)."""
    init, trans = load_lm_params()
    with np.errstate(divide="ignore"):
        return np.maximum(np.log(init), -1000.0), np.maximum(np.log(trans), -1000.0)


class StringPool:
    """Global pool of unique strings -> int id, with flat symbol / lm arrays."""

    def __init__(self):
        self.strings = []
        self.index = {}
        self._sym_of_cp = {}
        self._dirty = True

    def add(self, s):
        i = self.index.get(s)
        if i is None:
            i = len(self.strings)
            self.index[s] = i
            self.strings.append(s)
            self._dirty = True
        return i

    def add_all(self, values):
        return np.array([self.add(v) for v in values], dtype=np.int32)

    def __len__(self):
        return len(self.strings)

    def _build(self):
        if not self._dirty:
            return
        lens = np.array([len(s) for s in self.strings], dtype=np.int64)
        self.off = np.zeros(len(self.strings) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.off[1:])
        total = int(self.off[-1])
        self.cp = np.zeros(total, dtype=np.uint32)
        self.sym = np.zeros(total, dtype=np.uint16)
        self.lm = np.zeros(total, dtype=np.uint8)
        pos = 0
        for s in self.strings:
            for ch in s:
                c = ord(ch)
                sid = self._sym_of_cp.get(c)
                if sid is None:
                    sid = len(self._sym_of_cp)
                    self._sym_of_cp[c] = sid
                self.cp[pos] = c
                self.sym[pos] = sid
                low = ch.lower()
                self.lm[pos] = _ALPHA_IDX.get(low, 255) if len(low) == 1 else 255
                pos += 1
        self.lens = lens
        self._dirty = False

    def arrays(self):
        self._build()
        return self.sym, self.off, self.lm, self.cp

    def letter_symbols(self):
        """Pool symbol id of each of the 28 letters random(StringPrior) emits (string_prior.jl:28-39), 0xFFFF for a
        letter that occurs in no pool string (it then equals no observed symbol)."""
        self._build()
        return np.array([self._sym_of_cp.get(ord(ch), 0xFFFF) for ch in ALPHABET], dtype=np.uint16)


class Domain:
    """Ordered list of pool ids (unique observed values of a column, or the
    values a latent attribute may take)."""

    def __init__(self, pool, values=()):
        self.pool = pool
        self.ids = []
        self.local = {}
        for v in values:
            self.add(v)

    def add(self, s):
        pid = self.pool.add(s)
        j = self.local.get(pid)
        if j is None:
            j = len(self.ids)
            self.local[pid] = j
            self.ids.append(pid)
        return j

    def add_extra(self, s):
        """A value that is NOT one of the attribute's options even if its string equals one (a string drawn by
        random(StringPrior / TimePrior) for a chosen dummy): always a new id after the ones added so far; `extra`
        maps the string to it.  get / index_of keep answering with the first id of a string (the option)."""
        if not hasattr(self, "extra"):
            self.extra = {}
        j = self.extra.get(s)
        if j is None:
            pid = self.pool.add(s)
            j = len(self.ids)
            self.ids.append(pid)
            self.local.setdefault(pid, j)
            self.extra[s] = j
        return j

    def n_base(self):
        """number of values that are not extras (for a StringPrior / TimePrior attribute: atoms + the dummy)"""
        return len(self.ids) - len(getattr(self, "extra", {}))

    def index_of(self, s):
        return self.local[self.pool.index[s]]

    def get(self, s, default=-1):
        pid = self.pool.index.get(s)
        if pid is None:
            return default
        return self.local.get(pid, default)

    def string(self, j):
        return self.pool.strings[self.ids[j]]

    def __len__(self):
        return len(self.ids)

    def id_array(self):
        return np.array(self.ids, dtype=np.int32)
