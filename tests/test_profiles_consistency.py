"""profiles/hbm_traffic.json (what bench.py echoes as roofline.traffic / traffic_source) names files that exist, and the
round's collection script writes its summaries under the names those strings cite."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(d, out):
    if isinstance(d, dict):
        for k, v in d.items():
            if k == "source" and isinstance(v, str):
                out.append(v)
            else:
                _sources(v, out)
    return out


def test_traffic_sources_exist():
    d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    srcs = _sources(d, [])
    assert srcs
    cited = sorted({m for s in srcs for m in re.findall(r"profiles/[A-Za-z0-9_./-]+\.(?:txt|json|py)", s)})
    assert cited
    missing = [c for c in cited if not os.path.exists(os.path.join(ROOT, c))]
    assert not missing, missing


def test_collection_script_names_match_its_tag():
    sh = open(os.path.join(ROOT, "profiles", "collect_r06.sh")).read()
    assert 'TAG=${1:-r06}' in sh
    assert '--source "profiles/${TAG}_pmc_root_kernels.txt"' in sh
    assert 'cp "$OUT/pmc_root_kernels.txt" "$P/${TAG}_pmc_root_kernels.txt"' in sh
