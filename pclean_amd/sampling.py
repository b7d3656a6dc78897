"""`random(dist, args...)` of the noise models on the HIP path (distributions.jl:3-43 protocol;
add_typos.jl:9-45, string_prior.jl:28-39, choose_proportionally.jl:3-5, choose_uniformly.jl:3-5,
add_noise.jl:5, transformed_gaussian.jl:13, maybe_swap.jl:5-11, time_prior.jl:21-23).

Batched: every call draws n values at once on the GPU; results depend only on
(seed, stream, element index).  There is no CPU fallback — the functions take the engine's
HipContext and fail if the library or the device is missing.
"""
import numpy as np

from .encode import ALPHABET, load_lm_params


def random_add_typos(hip, words, max_typos=None, seed=0, stream=0):
    """Typo-ed copies of `words` (list of str)."""
    cps = [np.array([ord(c) for c in w], dtype=np.uint32) for w in words]
    off = np.zeros(len(words) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cps], out=off[1:])
    cp = np.concatenate(cps) if len(cps) and off[-1] else np.zeros(0, np.uint32)
    longest = int(max([len(c) for c in cps], default=0))
    stride = longest + (longest + 4) // 5 * 4 + 16  # inserts beyond this many are dropped (see pclean_hip.h)
    out, lens = hip.random_add_typos(cp, off, -1 if max_typos is None else int(max_typos), seed, stream, stride)
    return ["".join(chr(c) for c in out[i, :lens[i]]) for i in range(len(words))]


def random_string_prior(hip, n, min_len, max_len, seed=0, stream=0):
    init, trans = load_lm_params()
    out, lens = hip.random_string_prior(n, min_len, max_len, init, trans, seed, stream)
    return ["".join(ALPHABET[k] for k in out[i, :lens[i]]) for i in range(n)]


def dummy_seed(seed, site, particle, sweep):
    """pclean_dummy_seed (include/pclean_philox.h): key of the private draw stream of a value sampled for a chosen
    ProposalDummyValue at draw site `site` by `particle` in sweep `sweep`."""
    m = 0xFFFFFFFFFFFFFFFF
    x = (int(seed) ^ ((int(site) << 32) | (int(sweep) & 0xFFFFFFFF))) & m
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & m
    x ^= ((int(particle) + 1) & 0xFFFFFFFF) * 0x94d049bb133111eb & m
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & m
    return x ^ (x >> 31)


def random_string_prior_at(hip, seeds, elems, min_len, max_len):
    """random(StringPrior(min_len, max_len)) with a private stream per element (seeds[i], elems[i])."""
    init, trans = load_lm_params()
    out, lens = hip.random_string_prior_at(seeds, elems, min_len, max_len, init, trans)
    return ["".join(ALPHABET[k] for k in out[i, :lens[i]]) for i in range(len(lens))]


def random_choose_proportionally(hip, n, options, probs, seed=0, stream=0):
    with np.errstate(divide="ignore"):
        idx = hip.random_categorical(n, np.log(np.asarray(probs, dtype=np.float64)), seed, stream)
    return [options[k] for k in idx]


def random_choose_uniformly(hip, n, options, seed=0, stream=0):
    idx = hip.random_categorical(n, np.zeros(len(options)), seed, stream)
    return [options[k] for k in idx]


def random_add_noise(hip, mean, std, seed=0, stream=0):
    return hip.random_normal(np.asarray(mean, dtype=np.float64), std, 1.0, seed, stream)


def random_transformed_gaussian(hip, mean, std, forward_scale, seed=0, stream=0):
    """t.forward(rand(Normal(mean, std))) for a linear transformation x -> forward_scale * x."""
    return hip.random_normal(np.asarray(mean, dtype=np.float64), std, float(forward_scale), seed, stream)


def random_maybe_swap(hip, vals, options, probs, seed=0, stream=0):
    """vals[i] kept, or replaced by a uniformly chosen element of options[i] with probability probs[i]."""
    n_opt = np.array([len(o) for o in options], dtype=np.int32)
    idx = hip.random_maybe_swap(np.asarray(probs, dtype=np.float64), n_opt, seed, stream)
    return [v if k < 0 else o[k] for v, o, k in zip(vals, options, idx)]


def random_time_prior(hip, n, seed=0, stream=0):
    hm = hip.random_time_prior(n, seed, stream)
    return [f"{h}:{m} {'a.m.' if am else 'p.m.'}" for h, m, am in hm]
