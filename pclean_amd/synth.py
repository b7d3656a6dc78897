"""Synthetic hospital-shaped table (BASELINE.json configs[4]).

Generator owned by the build, seed 20250926: N rows over K_H latent hospitals
(~N/K_H rows each, multinomial), 28 measures x 5 conditions, 1 hospital type,
8 owners, 2 services, ~50 states, ~3000 counties, ~4000 cities; the 15 queried
string columns with a length profile close to hospital_dirty.csv; each cell is
independently corrupted w.p. 0.03 by k 'x'-substitutions, k drawn from the
empirical histogram of the real dirty/clean pair (261:108:42:18:26).  A clean
copy is kept for F1.  Rows are grouped by hospital like the real CSV.
"""
import numpy as np

_SYL = ["ba", "ce", "di", "fo", "gu", "ha", "je", "ki", "lo", "mu", "na", "pe", "qui", "ro", "su", "ta", "ve", "wi",
        "yo", "za", "an", "el", "in", "or", "un", "ar", "es", "ir", "os", "ur", "mar", "len", "tor", "bel", "shi"]
TYPO_HIST = np.array([261, 108, 42, 18, 26], dtype=np.float64)


def _word(rng, lo, hi):
    n = rng.integers(lo, hi + 1)
    s = ""
    while len(s) < n:
        s += _SYL[rng.integers(len(_SYL))]
    return s[:n]


def _phrase(rng, target_len, jitter=0.25):
    n = max(3, int(rng.normal(target_len, target_len * jitter)))
    words = []
    while sum(len(w) + 1 for w in words) < n:
        words.append(_word(rng, 3, 9))
    return " ".join(words)[:max(3, n)].strip()


def _unique(rng, gen, n):
    out, seen = [], set()
    while len(out) < n:
        s = gen()
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def _corrupt(rng, s):
    k = 1 + rng.choice(5, p=TYPO_HIST / TYPO_HIST.sum())
    k = min(k, len(s))
    pos = rng.choice(len(s), size=k, replace=False)
    cs = list(s)
    for p in pos:
        cs[p] = "x"
    return "".join(cs)


COLUMNS = ["ProviderNumber", "HospitalName", "Address1", "City", "State", "ZipCode", "CountyName", "PhoneNumber",
           "HospitalType", "HospitalOwner", "EmergencyService", "Condition", "MeasureCode", "MeasureName", "Stateavg"]


def synth_hospital(n_rows=1_000_000, n_hosp=10_000, seed=20250926, n_states=50, n_counties=3000, n_cities=4000,
                   n_measures=28, n_conditions=5, n_owners=8, p_typo=0.03):
    rng = np.random.default_rng(seed)
    n_counties = min(n_counties, max(n_states, n_hosp))
    n_cities = min(n_cities, max(n_counties, n_hosp))
    letters = "abcdefghijklmnopqrstuvwyz"  # no 'x': typos stay recognisable
    states = _unique(rng, lambda: letters[rng.integers(25)] + letters[rng.integers(25)], n_states)
    county_names = _unique(rng, lambda: _word(rng, 5, 11), n_counties)
    county_state = rng.integers(0, n_states, n_counties)
    city_names = _unique(rng, lambda: _phrase(rng, 9, 0.3), n_cities)
    city_county = rng.integers(0, n_counties, n_cities)
    owners = _unique(rng, lambda: _phrase(rng, 27, 0.3), n_owners)
    services = ["yes", "no"]
    htype = _phrase(rng, 20, 0.05)
    cond_names = _unique(rng, lambda: _phrase(rng, 17, 0.3), n_conditions)
    m_code = _unique(rng, lambda: f"{_word(rng, 2, 4)}-{_word(rng, 2, 4)}-{rng.integers(1, 10)}", n_measures)
    m_name = _unique(rng, lambda: _phrase(rng, 90, 0.4)[:184], n_measures)
    m_cond = rng.integers(0, n_conditions, n_measures)

    h_city = rng.integers(0, n_cities, n_hosp)
    h_provider = [str(v) for v in rng.choice(np.arange(10000, 100000), size=n_hosp, replace=False)]
    h_name = _unique(rng, lambda: _phrase(rng, 26, 0.25)[:50], n_hosp)
    h_addr = _unique(rng, lambda: (f"{rng.integers(100, 9999)} " + _phrase(rng, 15, 0.25))[:30].ljust(10, "a"), n_hosp)
    h_phone = _unique(rng, lambda: "".join(str(d) for d in rng.integers(0, 10, 10)), n_hosp)
    h_zip = [str(v) for v in rng.integers(10000, 100000, n_hosp)]
    h_owner = rng.integers(0, n_owners, n_hosp)
    h_service = rng.integers(0, 2, n_hosp)

    sizes = rng.multinomial(n_rows - n_hosp, np.ones(n_hosp) / n_hosp) + 1  # every hospital has >= 1 row
    row_h = np.repeat(np.arange(n_hosp), sizes)
    row_m = rng.integers(0, n_measures, n_rows)

    def col(vals, idx):
        a = np.array(vals, dtype=object)
        return a[idx]

    h_state = county_state[city_county[h_city]]
    clean = {
        "ProviderNumber": col(h_provider, row_h), "HospitalName": col(h_name, row_h), "Address1": col(h_addr, row_h),
        "City": col(city_names, h_city[row_h]), "State": col(states, h_state[row_h]), "ZipCode": col(h_zip, row_h),
        "CountyName": col(county_names, city_county[h_city][row_h]), "PhoneNumber": col(h_phone, row_h),
        "HospitalType": np.array([htype] * n_rows, dtype=object), "HospitalOwner": col(owners, h_owner[row_h]),
        "EmergencyService": col(services, h_service[row_h]), "Condition": col(cond_names, m_cond[row_m]),
        "MeasureCode": col(m_code, row_m), "MeasureName": col(m_name, row_m),
    }
    clean["Stateavg"] = np.array([f"{s}_{c}" for s, c in zip(clean["State"], clean["MeasureCode"])], dtype=object)
    dirty = {}
    for c in COLUMNS:
        d = clean[c].copy()
        hit = np.nonzero(rng.random(n_rows) < p_typo)[0]
        for i in hit:
            d[i] = _corrupt(rng, d[i])
        dirty[c] = d
    latent = dict(
        row_h=row_h, row_m=row_m,
        hospital=dict(provider=h_provider, name=h_name, addr=h_addr, phone=h_phone, zip=h_zip,
                      owner=[owners[i] for i in h_owner], service=[services[i] for i in h_service],
                      city_idx=h_city, type=htype),
        city=dict(name=city_names, county_idx=city_county),
        county=dict(name=county_names, state=[states[i] for i in county_state]),
        measure=dict(code=m_code, name=m_name, cond=[cond_names[i] for i in m_cond]),
    )
    return {c: list(dirty[c]) for c in COLUMNS}, {c: list(clean[c]) for c in COLUMNS}, latent
