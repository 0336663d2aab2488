"""Row (c) of SURVEY.md section 8, the one statement that can still be earned about an UNPINNED oracle: how many rays change
if a detail of madmann91/bvh v1 recalled in SURVEY section 3.2 (libs/bvh is absent; reference call sites
source/objects/AccelStruct.h:23-31, AccelStruct.cpp:818) is in fact the other plausible reading.

tests/golden/recall_sensitivity.json is written by scripts/recall_sensitivity.py (variants of oracle/vt_oracle.c built with
-DVTO_ALT_<X>).  These tests check (1) the oracle every other test uses is the shipped reading, (2) the committed table is
what the script produces today (the small workloads are re-run here), (3) the claims DESIGN.md section 2 quotes from it.
"""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JSON = os.path.join(ROOT, "tests", "golden", "recall_sensitivity.json")


@pytest.fixture(scope="module")
def rs():
    spec = importlib.util.spec_from_file_location("recall_sensitivity", os.path.join(ROOT, "scripts", "recall_sensitivity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def table():
    with open(JSON) as f:
        return json.load(f)


def test_the_checker_is_the_shipped_reading(O):
    """No VTO_ALT switch in the library tests / smoke / bench load; every variant library reports exactly its own switch."""
    assert O.lib().vto_alt_mask() == 0
    for i, name in enumerate(O.ALT_NAMES):
        assert O.alt_lib(name).vto_alt_mask() == 1 << i


def test_no_alt_switch_outside_the_oracle():
    """The switches live in oracle/vt_oracle.c only: the product has no second reading to fall into."""
    for top in ("vistrace_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            if "_build" in dirpath or "__pycache__" in dirpath:
                continue
            for fn in files:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                with open(os.path.join(dirpath, fn), errors="replace") as f:
                    assert "VTO_ALT" not in f.read(), os.path.join(dirpath, fn)


def test_committed_table_reproduces_on_the_small_workloads(rs, table):
    committed = {r["workload"]: r for r in table["workloads"]}
    n = 0
    for wl in rs.small_workloads():
        row = rs.measure(*wl)
        assert row == committed[row["workload"]], row["workload"]
        n += 1
    assert n == 7                      # two fixtures, four soups, weird rays
    assert len(table["workloads"]) == 9 and table["workloads"][0]["rays"] == 1 << 20


def test_what_design_md_quotes(table):
    tot = table["totals"]
    n = tot["FMA"]["rays"]
    assert n > 2_000_000
    # six of the eight readings: not one hit record differs (index, t, u, v, hit/miss) -- counters only
    for k in ("PLAIN_INVERSE", "SWAP_GE", "FMA", "RETEST_RIGHT", "PUSH_NODE_CULL", "FMINMAX"):
        assert tot[k]["miss_flip"] == tot[k]["prim"] == tot[k]["t"] == tot[k]["uv"] == 0, k
    # leaf-slot order and tree shape: only the tie-broken index (and with it that triangle's u, v); t never
    for k in ("LEAF_DESC", "TREE_PLOC"):
        assert tot[k]["miss_flip"] == 0 and tot[k]["t"] == 0 and 0 < tot[k]["prim"] < n // 1000, k
        assert tot[k]["uv"] <= tot[k]["prim"]
    # first < second is not a subtle alternative: flat (axis-aligned) leaves become unhittable
    assert tot["ACCEPT_LT"]["miss_flip"] > n // 2
