"""Every stage's byte model answers to its counters: profiles/r06_bench.json is the default bench line of the round-6 tree, written on the box
that also counted FETCH_SIZE / WRITE_SIZE for it (tools/final_profile.sh), so `all_stages_GB_per_step` holds model bytes and counter bytes of
the same code and the same workload.  A stage whose counters leave the band around its model is a model nobody can check a kernel against —
the state bench.py's pairs_insert model was in for a round (2.7 x its counters)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = os.path.join(ROOT, "profiles", "r06_bench.json")


@pytest.mark.skipif(not os.path.exists(LINE), reason="no round-6 bench line committed yet")
def test_every_stage_model_is_within_reach_of_its_counters():
    j = json.loads(open(LINE).read().strip().splitlines()[-1])
    roof = j["roofline"]
    assert roof["traffic"] and roof["traffic_withheld"] is None, "the committed line must carry counters of its own code"
    assert roof["build_csrc_id"] == roof["traffic_csrc_id"]
    upper = roof["upper_bound_models"]
    seen = 0
    for stage, gb in roof["all_stages_GB_per_step"].items():
        if gb["counters"] is None:
            continue
        seen += 1
        ratio = gb["counters"] / gb["model"]
        if stage in upper:
            assert upper[stage] and ratio <= 1.2, (stage, ratio)          # an upper bound, with its reason written down
        else:
            assert 0.6 <= ratio <= 1.2, (stage, ratio, gb)
    assert seen >= 8
    # the dominant kernel's line: achieved = model bytes / measured time, traffic = counters, both per launch and close to each other
    assert 0.9 <= roof["traffic"] / roof["algorithmic_bytes_per_launch"] <= 1.1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # the host-resident leg is on the line, equal filters, and within reach of the HBM-resident figure
    hr = j["host_resident"]
    assert hr["filters_equal_resident"] is True and hr["value"] >= 0.9 * j["value"]
