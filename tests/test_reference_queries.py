"""The reference's own SQL tests against the GPU shells: every `query` block of test/sql/*.test whose relations are calls
of this path's table functions (tests/golden/query_cases.json, extracted by tests/golden/make_query_cases.py), with the
SQL around the calls -- select lists, WHERE, GROUP BY, ORDER BY, LIMIT, aggregates, subqueries, DESCRIBE -- evaluated
by tests/sqlmini.py over the rows the shells return, and the result compared with the rows the reference's test
expects (numbers with a tolerance: a double's last digits depend on summation order)."""
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


class Relation:
    def __init__(self, names, types, rows):
        self.names, self.types, self.rows = names, types, rows


def companion_reader(function, path):
    """read_pvar / read_psam as join partners in the reference's queries: the oracle's text loaders stand in for them
    (the readers themselves are outside this path)."""
    import oracle.oracle as O

    if function == "read_pvar":
        v = O.load_pvar(path)
        return Relation(["CHROM", "POS", "ID", "REF", "ALT"], ["VARCHAR", "INTEGER", "VARCHAR", "VARCHAR", "VARCHAR"],
                        list(zip(v["chrom"], v["pos"], v["id"], v["ref"], v["alt"])))
    p = O.load_psam(path)
    sex = [int(x) if int(x) in (1, 2) else None for x in p["sex"]]
    if any(f is not None for f in p["fid"]):
        return Relation(["FID", "IID", "SEX"], ["VARCHAR", "VARCHAR", "INTEGER"], list(zip(p["fid"], p["iid"], sex)))
    return Relation(["IID", "SEX"], ["VARCHAR", "INTEGER"], list(zip(p["iid"], sex)))


def provider(settings, threads, variables=None):
    def call(function, args, named):
        args = [localise(a) for a in args]
        if function in ("read_pvar", "read_psam"):
            return companion_reader(function, args[0])
        named = {k: (localise(v) if k in PATH_PARAMS else v) for k, v in named.items()}
        return F.query(function, *args, settings=settings, threads=threads, **named)

    call.variables = variables
    return call


def run_case(case, threads=4):
    q = sqlmini.parse_sql(case["sql"])
    _, _, got = sqlmini.run_select(q, provider(case.get("settings"), threads, case.get("variables")))
    want = case["expected"]
    types = case["types"]
    assert len(got) == len(want), f"{len(got)} rows, the reference's test expects {len(want)}"
    rows = [tuple(sqlmini.duck_str(v) for v in g) for g in got]
    top = q
    while "body" in top:
        top = top["body"]
    ordered = bool(top.get("order")) and not case.get("rowsort")
    pairs = list(zip(got, want))
    if not ordered:
        pairs = list(zip([g for _, g in sorted(zip(rows, got), key=lambda t: t[0])], sorted(want)))

    def row_ok(g, w):
        return len(g) == len(w) and all(sqlmini.matches(v, t, types[k] if k < len(types) else "T") for k, (v, t) in enumerate(zip(g, w)))

    if all(row_ok(g, w) for g, w in pairs):
        return
    # numbers that print differently can sort differently as text, and ORDER BY keys need not make the order total:
    # match the rows up one by one
    left = list(want)
    for g in got:
        hit = next((w for w in left if row_ok(g, w)), None)
        assert hit is not None, f"row {tuple(sqlmini.duck_str(v) for v in g)} is not among the expected rows {left[:6]}"
        left.remove(hit)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["source"].split("/")[-1])
def test_reference_query(case):
    run_case(case)


@pytest.mark.parametrize("threads", [1, 13])
@pytest.mark.parametrize("case", CASES[::4], ids=lambda c: c["source"].split("/")[-1])
def test_reference_query_at_other_thread_counts(case, threads):
    """Every fourth query again with one scan thread and with thirteen: the rows must not depend on how the scan is cut
    (streaming_threading.test's point, over the whole suite)."""
    if "plinking_max_threads" in (case.get("settings") or {}):
        pytest.skip("the case sets its own thread cap")
    run_case(case, threads)
