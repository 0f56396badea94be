"""The six table functions end to end on the GPU, asserting the reference's own
known answers (tests/golden/known_answers.json <- test/sql/*.test).  Each call
goes SQL-shaped parameters -> C++ shell (bind/init/scan, multi-threaded) ->
libpgenhip C ABI -> HIP kernels."""

import json
import os

import numpy as np
import pytest

from conftest import data_path

pytestmark = pytest.mark.gpu

F = pytest.importorskip("plinking_duck_amd.functions")

with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
    KA = json.load(f)

EX = data_path("pgen_example.pgen")
W = [1.0, 0.5, -0.5, 2.0]


def r6(x):
    return None if x is None else round(x, 6)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_lib):
    return gpu_lib


# ---- plink_freq (plink_freq.test) ------------------------------------------

def test_freq_known_answers():
    r = F.query("plink_freq", EX, columns=["CHROM", "POS", "ID", "REF", "ALT", "ALT_FREQ", "OBS_CT"])
    assert r.sorted("CHROM", "POS") == [("1", 10000, "rs1", "A", "G", 0.5, 6), ("1", 20000, "rs2", "C", "T", 0.5, 8),
                                        ("1", 30000, "rs3", "G", "A", 0.5, 6), ("2", 15000, "rs4", "T", "C", 0.375, 8)]
    assert r.types == ["VARCHAR", "INTEGER", "VARCHAR", "VARCHAR", "VARCHAR", "DOUBLE", "INTEGER"]
    r = F.query("plink_freq", EX, counts=True,
                columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT", "OBS_CT"])
    assert r.sorted("ID") == [("rs1", 1, 1, 1, 1, 6), ("rs2", 1, 2, 1, 0, 8), ("rs3", 1, 1, 1, 1, 6),
                              ("rs4", 2, 1, 1, 0, 8)]
    assert np.mean(F.query("plink_freq", EX, columns=["ALT_FREQ"]).column("ALT_FREQ")) == 0.46875


def test_freq_subsets_regions_companions():
    for samples in (["SAMPLE1", "SAMPLE3"], [0, 2]):
        r = dict((row[0], row[1:]) for row in
                 F.query("plink_freq", EX, samples=samples, columns=["ID", "ALT_FREQ", "OBS_CT"]).rows)
        assert r["rs1"] == (0.5, 4) and r["rs2"] == (0.25, 4) and r["rs4"] == (0.25, 4)
    r = F.query("plink_freq", EX, samples=["SAMPLE1", "SAMPLE3"], counts=True,
                columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT", "OBS_CT"])
    assert ("rs2", 1, 1, 0, 0, 4) in r.rows
    assert F.query("plink_freq", data_path("pgen_orphan.pgen"), samples=[0, 2],
                   columns=["ID", "ALT_FREQ", "OBS_CT"]).sorted("ID")[0] == ("rs1", 0.5, 4)
    r = F.query("plink_freq", EX, region="1:10000-20000", columns=["ID", "ALT_FREQ", "OBS_CT"])
    assert r.sorted("ID") == [("rs1", 0.5, 6), ("rs2", 0.5, 8)]
    assert F.query("plink_freq", EX, region="2:15000-15000", columns=["ID", "ALT_FREQ", "OBS_CT"]).rows == [
        ("rs4", 0.375, 8)]
    assert len(F.query("plink_freq", EX, region="1:1-100000")) == 3
    assert F.query("plink_freq", EX, region="1:10000-10000", samples=["SAMPLE1", "SAMPLE3"],
                   columns=["ID", "ALT_FREQ", "OBS_CT"]).rows == [("rs1", 0.5, 4)]
    assert ("rs1", 0.5, 6) in F.query("plink_freq", EX, pvar=data_path("pgen_example.bim"),
                                      columns=["ID", "ALT_FREQ", "OBS_CT"]).rows


def test_freq_all_missing_and_large():
    r = F.query("plink_freq", data_path("all_missing.pgen"), counts=True,
                columns=["ID", "ALT_FREQ", "OBS_CT", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"])
    assert r.sorted("ID") == [("rs_miss1", None, 0, 0, 0, 0, 2), ("rs_miss2", None, 0, 0, 0, 0, 2)]
    r = F.query("plink_freq", data_path("large_example.pgen"), columns=["CHROM", "ID", "ALT_FREQ", "OBS_CT"])
    assert len(r) == 3000 and len(set(r.column("ID"))) == 3000
    assert set((a, o) for _, _, a, o in r.rows) == {(0.5, 12)}
    chroms = r.column("CHROM")
    assert [chroms.count(c) for c in ("1", "2", "3")] == [1000, 1000, 1000]
    assert len(F.query("plink_freq", data_path("large_example.pgen"), region="1:100-1000")) == 10
    assert len(F.query("plink_freq", data_path("large_example.pgen"), region="2:100-500")) == 5


def test_freq_thread_counts_give_the_same_rows():
    """streaming_threading.test: exact row count, no duplicates, any thread count."""
    big = data_path("streaming_example.pgen")
    base = None
    for threads, cap in ((1, None), (4, None), (16, 2)):
        settings = {"plinking_max_threads": cap} if cap else None
        r = F.query("plink_freq", big, threads=threads, settings=settings, counts=True,
                    columns=["ID", "ALT_FREQ", "OBS_CT", "HOM_REF_CT", "MISSING_CT"])
        rows = sorted(r.rows, key=lambda t: t[0])
        assert len(rows) == 50000 and len(set(t[0] for t in rows)) == 50000
        base = base or rows
        assert rows == base


def test_freq_dosage():
    ka = KA["dosage_example"]
    dz = data_path("dosage_example.pgen")
    r = F.query("plink_freq", EX, dosage=True, columns=["ID", "ALT_FREQ", "OBS_CT", "IMP_R2"])
    assert r.sorted("ID") == [("rs1", 0.5, 6, None), ("rs2", 0.5, 8, None), ("rs3", 0.5, 6, None),
                              ("rs4", 0.375, 8, None)]
    r = F.query("plink_freq", EX, dosage=True, samples=["SAMPLE1", "SAMPLE3"], columns=["ID", "ALT_FREQ", "OBS_CT"])
    assert r.sorted("ID") == [("rs1", 0.5, 4), ("rs2", 0.25, 4), ("rs3", 0.75, 4), ("rs4", 0.25, 4)]
    h = F.query("plink_freq", dz, columns=["ID", "ALT_FREQ", "OBS_CT"]).sorted("ID")
    d = F.query("plink_freq", dz, dosage=True, columns=["ID", "ALT_FREQ", "OBS_CT", "IMP_R2"]).sorted("ID")
    for v in range(4):
        assert h[v][1:] == (ka["hardcall_freq"][v], ka["hardcall_obs"][v])
        assert d[v][1:3] == (ka["dosage_freq"][v], ka["dosage_obs"][v])
        assert d[v][3] == pytest.approx(ka["imp_r2"][v], rel=1e-12)


# ---- plink_hardy (plink_hardy.test, plink_sexchr.test) ---------------------------

def test_hardy_known_answers():
    cols = ["ID", "A1", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "O_HET", "E_HET", "P_HWE"]
    rows = F.query("plink_hardy", EX, columns=cols).sorted("ID")
    mid = F.query("plink_hardy", EX, midp=True, columns=["ID", "P_HWE"]).sorted("ID")
    for v, exp in enumerate(KA["hardy_pgen_example"]["rows"]):
        vid, a1, c0, c1, c2, o_het, e_het, p = rows[v]
        assert (vid, [c0, c1, c2]) == (exp["id"], exp["counts"])
        assert (r6(o_het), r6(e_het), r6(p)) == (exp["o_het"], exp["e_het"], exp["p"])
        assert r6(mid[v][1]) == exp["p_midp"]
        assert mid[v][1] <= p + 1e-6
    assert [r[1] for r in rows] == ["G", "T", "A", "C"]
    assert r6(float(np.mean([r[7] for r in rows]))) == 0.857143
    r = F.query("plink_hardy", EX, samples=["SAMPLE1", "SAMPLE2"], columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT"])
    assert ("rs1", 1, 1, 0) in r.rows
    r = F.query("plink_hardy", data_path("pgen_orphan.pgen"), samples=[0, 2], columns=["ID", "P_HWE"])
    assert r6(dict(r.rows)["rs1"]) == 0.333333
    r = F.query("plink_hardy", EX, region="1:10000-10000", samples=["SAMPLE1", "SAMPLE3"],
                columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT"])
    assert r.rows == [("rs1", 1, 0, 1)]
    r = F.query("plink_hardy", data_path("all_missing.pgen"),
                columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "O_HET", "P_HWE"])
    assert r.sorted("ID") == [("rs_miss1", 0, 0, 0, None, None), ("rs_miss2", 0, 0, 0, None, None)]
    r = F.query("plink_hardy", data_path("large_example.pgen"), columns=["ID", "P_HWE"])
    assert len(r) == 3000 and set(r6(p) for _, p in r.rows) == {0.480519}


def test_sex_chromosomes():
    sx = data_path("sexchr_example.pgen")
    ka = KA["sexchr"]
    fr = dict((r[0], r[1:]) for r in F.query("plink_freq", sx, columns=["ID", "ALT_FREQ", "OBS_CT"]).rows)
    for vid, (af, obs) in ka["freq"].items():
        assert (r6(fr[vid][0]), fr[vid][1]) == (af, obs)
    hw = dict((r[0], r[1:]) for r in F.query(
        "plink_hardy", sx, columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "O_HET", "E_HET", "P_HWE"]).rows)
    for vid, exp in ka["hardy"].items():
        c0, c1, c2, o_het, e_het, p = hw[vid]
        assert [c0, c1, c2] == exp["counts"]
        if "p" in exp:
            assert (r6(o_het), r6(e_het), r6(p)) == (exp["o_het"], exp["e_het"], exp["p"])
        else:
            assert (o_het, e_het, p) == (None, None, None)  # haploid: HWE undefined
    mid = dict(F.query("plink_hardy", sx, midp=True, columns=["ID", "P_HWE"]).rows)
    assert r6(mid["x1"]) == ka["hardy"]["x1"]["p_midp"]
    cnt = dict((r[0], list(r[1:])) for r in F.query(
        "plink_freq", sx, counts=True, columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"]).rows)
    assert cnt["y1"] == ka["freq_counts_y1"]
    xp = data_path("sexchr_xpar.pvar")
    r = dict((r[0], r[1:]) for r in F.query("plink_freq", sx, pvar=xp, columns=["ID", "ALT_FREQ", "OBS_CT"]).rows)
    assert (r6(r["par1"][0]), r["par1"][1]) == (0.333333, 12)
    r = dict((r[0], r[1:]) for r in
             F.query("plink_freq", sx, pvar=xp, build="none", columns=["ID", "ALT_FREQ", "OBS_CT"]).rows)
    assert list(r["par1"]) == ka["xpar_build_none_par1"]


# ---- plink_missing (plink_missing.test, plink_missing_sample.test) ---------------------

def test_missing_variant_and_sample_modes():
    r = F.query("plink_missing", EX, columns=["ID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("ID") == [("rs1", 1, 3, 0.25), ("rs2", 0, 4, 0.0), ("rs3", 1, 3, 0.25), ("rs4", 0, 4, 0.0)]
    r = F.query("plink_missing", EX, mode="sample", columns=["FID", "IID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("IID") == [(None, "SAMPLE1", 0, 4, 0.0), (None, "SAMPLE2", 1, 3, 0.25),
                               (None, "SAMPLE3", 0, 4, 0.0), (None, "SAMPLE4", 1, 3, 0.25)]
    r = F.query("plink_missing", EX, mode="sample", samples=["SAMPLE1", "SAMPLE2"],
                columns=["IID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("IID") == [("SAMPLE1", 0, 4, 0.0), ("SAMPLE2", 1, 3, 0.25)]
    r = F.query("plink_missing", EX, mode="sample", region="1:10000-20000",
                columns=["IID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("IID") == [("SAMPLE1", 0, 2, 0.0), ("SAMPLE2", 0, 2, 0.0), ("SAMPLE3", 0, 2, 0.0),
                               ("SAMPLE4", 1, 1, 0.5)]
    r = F.query("plink_missing", EX, mode="sample", region="1:10000-20000", samples=["SAMPLE1", "SAMPLE4"],
                columns=["IID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("IID") == [("SAMPLE1", 0, 2, 0.0), ("SAMPLE4", 1, 1, 0.5)]
    r = F.query("plink_missing", data_path("all_missing.pgen"), mode="sample",
                columns=["IID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert r.sorted("IID") == [("SAMPLE1", 2, 0, 1.0), ("SAMPLE2", 2, 0, 1.0)]
    r = F.query("plink_missing", data_path("large_example.pgen"), mode="sample",
                columns=["MISSING_CT", "OBS_CT", "F_MISS"])
    assert len(r) == 8 and set(r.rows) == {(750, 2250, 0.25)}
    # variant mode with a subset: denominator is the subset size
    r = F.query("plink_missing", EX, samples=[0, 3], columns=["ID", "MISSING_CT", "OBS_CT", "F_MISS"])
    assert ("rs1", 1, 1, 0.5) in r.rows


# ---- plink_score (plink_score.test) -----------------------------------------------

def test_score_known_answers():
    ka = KA["score_pgen_example"]
    cols = ["FID", "IID", "ALLELE_CT", "DENOM", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM", "SCORE_AVG"]
    rows = F.query("plink_score", EX, weights=W, columns=cols).sorted("IID")
    for i, row in enumerate(rows):
        assert row == (None, f"SAMPLE{i + 1}", ka["default"]["allele_ct"][i], ka["default"]["allele_ct"][i],
                       ka["default"]["dosage_sum"][i], ka["default"]["score_sum"][i], ka["default"]["score_avg"][i])
    idw = [{"id": "rs1", "allele": "G", "weight": 1.0}, {"id": "rs2", "allele": "T", "weight": 0.5},
           {"id": "rs3", "allele": "A", "weight": -0.5}, {"id": "rs4", "allele": "C", "weight": 2.0}]
    assert dict(F.query("plink_score", EX, weights=idw, columns=["IID", "SCORE_SUM"]).rows)["SAMPLE1"] == -0.5
    flip = dict(F.query("plink_score", EX, weights=[{"id": "rs1", "allele": "A", "weight": 1.0}],
                        columns=["IID", "SCORE_SUM"]).rows)
    assert (flip["SAMPLE1"], flip["SAMPLE2"], flip["SAMPLE3"], flip["SAMPLE4"]) == (2.0, 1.0, 0.0, 1.0)
    part = dict(F.query("plink_score", EX, weights=idw[:2], columns=["IID", "SCORE_SUM"]).rows)
    assert part["SAMPLE1"] == 0.5
    skip = dict(F.query("plink_score", EX, weights=[idw[0], {"id": "rs_nonexistent", "allele": "A", "weight": 99.0}],
                        columns=["IID", "SCORE_SUM"]).rows)
    assert skip["SAMPLE2"] == 1.0
    z = F.query("plink_score", EX, weights=[0.0] * 4, columns=["IID", "ALLELE_CT", "SCORE_SUM", "SCORE_AVG"])
    assert ("SAMPLE1", 0, 0.0, 0.0) in z.rows
    sub = F.query("plink_score", EX, weights=W, samples=["SAMPLE1", "SAMPLE3"], columns=["IID", "SCORE_SUM"])
    assert len(sub) == 2 and dict(sub.rows)["SAMPLE1"] == -0.5
    assert len(F.query("plink_score", EX, weights=W, samples=[0, 2])) == 2
    nm = dict((r[0], r[1:]) for r in F.query("plink_score", EX, weights=W, no_mean_imputation=True,
                                             columns=["IID", "ALLELE_CT", "SCORE_SUM", "NAMED_ALLELE_DOSAGE_SUM"]).rows)
    assert nm["SAMPLE2"] == (6, 1.5, 2.0) and nm["SAMPLE4"] == (6, 5.0, 4.0) and nm["SAMPLE1"][1] == -0.5
    am = F.query("plink_score", data_path("all_missing.pgen"), weights=[1.0, 0.5],
                 columns=["IID", "ALLELE_CT", "SCORE_SUM", "SCORE_AVG"])
    assert ("SAMPLE1", 0, 0.0, 0.0) in am.rows
    rg = dict(F.query("plink_score", EX, weights=[1.0, 0.5], region="1:10000-20000",
                      columns=["IID", "SCORE_SUM"]).rows)
    assert rg["SAMPLE1"] == 0.5
    ce = dict((r[0], r[1:]) for r in F.query("plink_score", EX, weights=[1.0], region="1:20000-20000", center=True,
                                             columns=["IID", "ALLELE_CT", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM"]).rows)
    assert ce["SAMPLE1"] == (2, 0.0, 0.0)
    assert ce["SAMPLE3"][2] == pytest.approx(-1.414213562373095, rel=1e-15)
    assert ce["SAMPLE4"][2] == pytest.approx(1.414213562373095, rel=1e-15)
    c2 = dict(F.query("plink_score", EX, weights=W, center=True, columns=["IID", "ALLELE_CT"]).rows)
    assert c2["SAMPLE2"] == 6
    assert sum(1 for s in F.query("plink_score", EX, weights=W, columns=["SCORE_SUM"]).column("SCORE_SUM") if s > 0) == 3


def test_score_on_a_dosage_file_uses_the_dosages(oracle):
    """Dosage tracks are honoured (PgrGetD semantics), checked against the oracle."""
    dz = data_path("dosage_example.pgen")
    w = [0.3, -1.2, 0.7, 2.0]
    rows = F.query("plink_score", dz, weights=w, columns=["IID", "ALLELE_CT", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM"])
    pg = oracle.Pgen(dz)
    s, d, ac = oracle.score(pg, range(4), w)
    got = rows.sorted("IID")
    for i in range(4):
        assert got[i][1] == ac[i]
        assert got[i][2] == pytest.approx(d[i], rel=1e-12)
        assert got[i][3] == pytest.approx(s[i, 0], rel=1e-12)


# ---- read_pgen (read_pgen.test, read_pgen_genotypes/filter/dosage/phased.test) ------------------

def test_read_pgen_genotypes():
    exp = [[(None if g == -9 else g) for g in row] for row in KA["pgen_example_genotypes"]["matrix"]]
    for mode, typ in (("array", "TINYINT[4]"), ("list", "TINYINT[]"), ("auto", "TINYINT[4]")):
        r = F.query("read_pgen", EX, genotypes=mode, columns=["ID", "genotypes"])
        assert r.types[1] == typ
        assert [g for _, g in r.sorted("ID")] == exp
    r = F.query("read_pgen", EX, samples=[0, 2], columns=["ID", "genotypes"])
    assert r.types[1] == "TINYINT[2]" and [g for _, g in r.sorted("ID")] == [[row[0], row[2]] for row in exp]
    # a samples list is applied as a mask: output order is file order
    r = F.query("read_pgen", EX, samples=["SAMPLE3", "SAMPLE1"], columns=["ID", "genotypes"])
    assert [g for _, g in r.sorted("ID")] == [[row[0], row[2]] for row in exp]
    am = F.query("read_pgen", data_path("all_missing.pgen"), columns=["ID", "genotypes"])
    assert [g for _, g in am.sorted("ID")] == [[None, None], [None, None]]
    big = F.query("read_pgen", data_path("large_example.pgen"), columns=["ID", "genotypes"], threads=4)
    assert len(big) == 3000 and all(len(g) == 8 for _, g in big.rows)


def test_read_pgen_counts_stats_and_filters():
    r = F.query("read_pgen", EX, genotypes="counts", columns=["ID", "genotypes"])
    got = {vid: [g["hom_ref"], g["het"], g["hom_alt"], g["missing"]] for vid, g in r.rows}
    assert [got[v] for v in ("rs1", "rs2", "rs3", "rs4")] == KA["pgen_example_freq"]["counts"]
    r = F.query("read_pgen", EX, genotypes="stats", columns=["ID", "genotypes"])
    st = dict(r.rows)["rs4"]
    assert (st["n"], st["af"], st["maf"], st["missing_rate"], st["carrier_count"], st["het_rate"]) == (
        4, 0.375, 0.375, 0.0, 2, 0.25)
    st = dict(F.query("read_pgen", data_path("all_missing.pgen"), genotypes="stats",
                      columns=["ID", "genotypes"]).rows)["rs_miss1"]
    assert st["n"] == 0 and np.isnan(st["af"]) and st["missing_rate"] == 1.0
    # af_range keeps rs4 only (0.375); ac_range on the allele count
    assert F.query("read_pgen", EX, af_range={"max": 0.4}, columns=["ID"]).column("ID") == ["rs4"]
    assert sorted(F.query("read_pgen", EX, ac_range={"min": 4}, columns=["ID"]).column("ID")) == ["rs2"]
    # include_genotypes nulls out the other calls but keeps the row if any sample matches
    r = dict(F.query("read_pgen", EX, include_genotypes=["het"], columns=["ID", "genotypes"]).rows)
    assert r["rs1"] == [None, 1, None, None] and r["rs2"] == [1, 1, None, None]
    r = dict(F.query("read_pgen", EX, genotype_range={"min": 2}, columns=["ID", "genotypes"]).rows)
    assert r["rs1"] == [None, None, 2, None] and "rs1" in r and len(r) == 4
    r = F.query("read_pgen", data_path("all_missing.pgen"), include_genotypes=["het"], columns=["ID"])
    assert len(r) == 0


def test_read_pgen_dosages_and_phase():
    ka = KA["dosage_example"]
    r = F.query("read_pgen", data_path("dosage_example.pgen"), dosages=True, columns=["ID", "genotypes"])
    assert r.types[1] == "DOUBLE[4]"
    assert [g for _, g in r.sorted("ID")] == ka["dosages"]
    h = dict(F.query("read_pgen", data_path("dosage_example.pgen"), columns=["ID", "genotypes"]).rows)
    assert h["rs2"][0] is None
    ph = data_path("phased_example.pgen")
    r = F.query("read_pgen", ph, phased=True, columns=["ID", "genotypes"])
    assert r.types[1] == "TINYINT[2][4]"
    assert [g for _, g in r.sorted("ID")] == KA["phased_example"]["pairs"]
    r = F.query("read_pgen", ph, phased=True, genotypes="list", columns=["ID", "genotypes"])
    assert r.types[1] == "TINYINT[2][]" and dict(r.rows)["rs2"] == [[0, 1], [1, 0], [0, 0], [1, 1]]
    r = dict(F.query("read_pgen", ph, phased=True, samples=[0, 2], columns=["ID", "genotypes"]).rows)
    assert r["rs1"] == [[0, 0], [1, 0]] and r["rs2"] == [[0, 1], [0, 0]]
    r = dict(F.query("read_pgen", ph, phased=True, samples=["SAMPLE2", "SAMPLE4"], columns=["ID", "genotypes"]).rows)
    assert r["rs3"] == [None, [0, 0]]
    assert F.query("read_pgen", ph, columns=["genotypes"]).types == ["TINYINT[4]"]


def test_read_pgen_columns_and_struct_modes():
    """read_pgen_genotypes_columns.test, read_pfile_genotypes_struct.test:48-60, read_pgen_dosage.test:85-95."""
    exp = {vid: [(None if g == -9 else g) for g in row]
           for vid, row in zip(("rs1", "rs2", "rs3", "rs4"), KA["pgen_example_genotypes"]["matrix"])}
    S = ["SAMPLE1", "SAMPLE2", "SAMPLE3", "SAMPLE4"]
    r = F.query("read_pgen", EX, genotypes="columns", columns=["ID"] + S)
    assert r.all_names == ["CHROM", "POS", "ID", "REF", "ALT"] + S
    assert r.types == ["VARCHAR"] + ["TINYINT"] * 4 and len(r) == 4
    assert {row[0]: list(row[1:]) for row in r.rows} == exp
    assert exp["rs1"] == [0, 1, 2, None] and exp["rs2"] == [1, 1, 0, 2]
    r = F.query("read_pgen", EX, genotypes="columns", samples=["SAMPLE3", "SAMPLE1"],
                columns=["ID", "SAMPLE1", "SAMPLE3"])
    assert r.all_names[5:] == ["SAMPLE1", "SAMPLE3"]
    assert dict((row[0], row[1:]) for row in r.rows) == {"rs1": (0, 2), "rs2": (1, 0), "rs3": (2, 1), "rs4": (0, 1)}
    # projecting one sample column out of order still reads the right sample
    r = F.query("read_pgen", EX, genotypes="columns", columns=["SAMPLE4", "ID"])
    assert sorted(r.rows, key=lambda t: t[1]) == [(None, "rs1"), (2, "rs2"), (0, "rs3"), (2, "rs4")]
    r = F.query("read_pgen", EX, genotypes="columns", columns=["CHROM", "POS"])
    assert r.sorted("CHROM", "POS") == [("1", 10000), ("1", 20000), ("1", 30000), ("2", 15000)]
    r = F.query("read_pgen", data_path("all_missing.pgen"), genotypes="columns", columns=["ID", "SAMPLE1", "SAMPLE2"])
    assert r.sorted("ID") == [("rs_miss1", None, None), ("rs_miss2", None, None)]
    r = F.query("read_pgen", EX, genotypes="columns", dosages=True, columns=["ID", "SAMPLE1", "SAMPLE4"])
    assert r.types == ["VARCHAR", "DOUBLE", "DOUBLE"] and dict((a, (b, c)) for a, b, c in r.rows)["rs1"] == (0.0, None)
    r = F.query("read_pgen", EX, genotypes="columns", include_genotypes=["het"], columns=["ID"] + S)
    assert dict((row[0], list(row[1:])) for row in r.rows)["rs2"] == [1, 1, None, None]

    r = F.query("read_pgen", EX, genotypes="struct", columns=["ID", "genotypes"])
    assert r.types[1] == "STRUCT(SAMPLE1 TINYINT, SAMPLE2 TINYINT, SAMPLE3 TINYINT, SAMPLE4 TINYINT)"
    assert {vid: [g[s] for s in S] for vid, g in r.rows} == exp
    r = F.query("read_pgen", EX, genotypes="struct", samples=["SAMPLE1", "SAMPLE3"], columns=["ID", "genotypes"])
    assert dict(r.rows) == {"rs1": {"SAMPLE1": 0, "SAMPLE3": 2}, "rs2": {"SAMPLE1": 1, "SAMPLE3": 0},
                            "rs3": {"SAMPLE1": 2, "SAMPLE3": 1}, "rs4": {"SAMPLE1": 0, "SAMPLE3": 1}}
    ph = data_path("phased_example.pgen")
    r = dict(F.query("read_pgen", ph, phased=True, genotypes="struct", columns=["ID", "genotypes"]).rows)
    assert [r["rs2"][f"SAMPLE{i}"] for i in range(1, 5)] == [[0, 1], [1, 0], [0, 0], [1, 1]]
    d = dict(F.query("read_pgen", data_path("dosage_example.pgen"), dosages=True, genotypes="struct",
                     columns=["ID", "genotypes"]).rows)
    assert [[d[v][f"SAMPLE{i}"] for i in range(1, 5)] for v in sorted(d)] == KA["dosage_example"]["dosages"]


def test_read_pgen_variants_parameter(oracle):
    """read_pgen_variants.test with genotype columns, plus lists that cross the list-batch size."""
    exp = [[(None if g == -9 else g) for g in row] for row in KA["pgen_example_genotypes"]["matrix"]]
    r = F.query("read_pgen", EX, variants=[3, 0], columns=["ID", "genotypes"])
    assert r.rows == [("rs4", exp[3]), ("rs1", exp[0])]
    r = F.query("read_pgen", EX, variants={"start": 1, "stop": 2}, genotypes="counts", columns=["ID", "genotypes"])
    assert [(v, [g["hom_ref"], g["het"], g["hom_alt"], g["missing"]]) for v, g in r.rows] == [
        ("rs2", KA["pgen_example_freq"]["counts"][1]), ("rs3", KA["pgen_example_freq"]["counts"][2])]
    assert F.query("read_pgen", EX, variants=[0, 1, 2, 3], af_range={"max": 0.4}, columns=["ID"]).column("ID") == ["rs4"]
    big = data_path("large_example.pgen")
    pg = oracle.Pgen(big)
    rng = np.random.default_rng(5)
    pick = [int(v) for v in rng.permutation(3000)[:700]]
    allrows = F.query("read_pgen", big, columns=["CHROM", "POS", "genotypes"])
    full = {(c, p): g for c, p, g in allrows.rows}
    sel = F.query("read_pgen", big, variants=pick, columns=["CHROM", "POS", "genotypes"], threads=3)
    assert len(sel) == 700 and len(set((c, p) for c, p, _ in sel.rows)) == 700
    assert all(full[(c, p)] == g for c, p, g in sel.rows)
    got = sorted(tuple(-9 if x is None else x for x in g) for _, _, g in sel.rows)
    assert got == sorted(tuple(int(x) for x in pg.geno(v)) for v in pick)
    # one thread walks the list in caller order
    one = F.query("read_pgen", big, variants=pick[:300], columns=["genotypes"], threads=1)
    assert [tuple(-9 if x is None else x for x in g) for (g,) in one.rows] == [
        tuple(int(x) for x in pg.geno(v)) for v in pick[:300]]


def test_read_pgen_filters_across_device_batches(oracle):
    """50,000 variants = 4 device batches: filtered chunks must not straddle a batch claim."""
    path = data_path("streaming_example.pgen")
    pg = oracle.Pgen(path)
    c = np.array([pg.counts(v) for v in range(pg.M)], dtype=np.int64)
    obs = c[:, 0] + c[:, 1] + c[:, 2]
    af = np.where(obs > 0, (c[:, 1] + 2 * c[:, 2]) / np.maximum(2 * obs, 1), np.nan)
    keep = np.flatnonzero((obs > 0) & (af >= 0.3) & (af <= 0.6))
    for threads in (1, 4):
        r = F.query("read_pgen", path, af_range={"min": 0.3, "max": 0.6}, genotypes="counts",
                    columns=["POS", "CHROM", "genotypes"], threads=threads)
        assert len(r) == len(keep)
        got = sorted((g["hom_ref"], g["het"], g["hom_alt"], g["missing"]) for _, _, g in r.rows)
        assert got == sorted(tuple(int(x) for x in c[v]) for v in keep)
    r = F.query("read_pgen", path, af_range={"min": 0.3, "max": 0.6}, columns=["genotypes"], threads=2)
    got = sorted(tuple(-9 if x is None else x for x in g) for (g,) in r.rows)
    assert got == sorted(tuple(int(x) for x in pg.geno(v)) for v in keep)


# ---- plink_pca (plink_pca.test) ------------------------------------------------------

def test_pca_known_answers():
    pca = data_path("pca_example.pgen")
    ka = KA["pca_example"]
    for threads in (1, 4):
        r = F.query("plink_pca", pca, n_pcs=3, mode="pcs", threads=threads)
        assert r.names == ["PC", "EIGENVALUE", "VARIANCE_PROPORTION", "CUMULATIVE_VARIANCE"]
        rows = r.sorted("PC")
        assert [row[0] for row in rows] == [1, 2, 3]
        assert [round(row[1], 10) for row in rows] == ka["eigenvalues_round10"]
        assert rows[0][1] == pytest.approx(ka["eigenvalue1_full"], rel=1e-9)
        assert round(sum(row[2] for row in rows), 6) == 1.0 and max(row[3] for row in rows) == pytest.approx(1.0)
    r = F.query("plink_pca", pca, n_pcs=3)
    assert r.names == ["FID", "IID", "PC1", "PC2", "PC3"] and len(r) == 250
    assert sorted(r.column("IID"))[:3] == ["per0", "per1", "per10"] and None not in r.column("PC1")
    vecs = np.array([row[2:] for row in r.rows])
    assert np.allclose(vecs.T @ vecs, np.eye(3), atol=1e-9)
    assert len(F.query("plink_pca", pca, columns=["IID"]).all_names) == 12  # default n_pcs = 10
    both = F.query("plink_pca", pca, n_pcs=3, mode="both")
    assert len(both) == 1 and len(both.rows[0][0]) == 250
    assert [round(x, 10) for x in both.rows[0][1]] == ka["eigenvalues_round10"]
    assert len(F.query("plink_pca", pca, n_pcs=3, region="1:1-5000")) == 250
    ten = [f"per{i}" for i in range(10)]
    assert len(F.query("plink_pca", pca, n_pcs=1, samples=ten)) == 10
    with pytest.raises(F.InvalidInputException, match="too few samples"):
        F.query("plink_pca", data_path("large_example.pgen"), n_pcs=3)
    with pytest.raises(F.InvalidInputException, match="too few variants"):
        F.query("plink_pca", data_path("pgen_example.pgen"), n_pcs=1)
    with pytest.raises(F.InvalidInputException, match="n_pcs"):
        F.query("plink_pca", pca, n_pcs=250)


def test_pca_matches_oracle_with_subset(gpu_lib, oracle):
    """pgh_pca against the numpy restatement, eigenvectors up to sign."""
    path = data_path("pca_example.pgen")
    pg = oracle.Pgen(path)
    rng = np.random.default_rng(4)
    mask = rng.random(pg.N) < 0.8
    ev, vecs, m_eff = oracle.pca(pg, 2, include=mask.astype(np.uint8))
    names = oracle.load_psam(data_path("pca_example.psam"))["iid"]
    r = F.query("plink_pca", path, n_pcs=2, samples=[names[i] for i in np.flatnonzero(mask)])
    got = np.array([row[2:] for row in sorted(r.rows, key=lambda t: names.index(t[1]))])
    for pc in range(2):
        sign = np.sign(np.dot(got[:, pc], vecs[:, pc]))
        assert np.allclose(got[:, pc] * sign, vecs[:, pc], rtol=0, atol=1e-7)
    ev_got = F.query("plink_pca", path, n_pcs=2, mode="pcs",
                     samples=[names[i] for i in np.flatnonzero(mask)]).sorted("PC")
    assert [row[1] for row in ev_got] == pytest.approx(list(ev), rel=1e-6)


# ---- plink_ld (plink_ld.test, plink_ld_window.test) ------------------------------------------

def test_ld_pairwise_known_answers():
    ka = KA["plink_ld"]
    for a, b, r2, dp, n in ka["pairwise"]:
        r = F.query("plink_ld", EX, variant1=a, variant2=b)
        assert len(r) == 1
        row = dict(zip(r.names, r.rows[0]))
        assert (row["ID_A"], row["ID_B"], row["OBS_CT"]) == (a, b, n) and row["R2"] == pytest.approx(r2, rel=1e-12)
        if dp is not None:
            assert row["D_PRIME"] == pytest.approx(dp, rel=1e-12)
    r = F.query("plink_ld", EX, variant1="rs1", variant2="rs2")
    assert r.rows[0][:6] == ("1", 10000, "rs1", "1", 20000, "rs2") and r.rows[0][8] == 3
    assert r.rows[0][6:8] == pytest.approx((0.75, 0.5), rel=1e-12)  # sqllogictest's R columns compare to ~1e-10 too
    assert r.types == ["VARCHAR", "INTEGER", "VARCHAR", "VARCHAR", "INTEGER", "VARCHAR", "DOUBLE", "DOUBLE", "INTEGER"]
    r = F.query("plink_ld", EX, variant1="rs1", variant2="rs4", columns=["ID_A", "CHROM_A", "ID_B", "CHROM_B", "R2", "OBS_CT"])
    assert r.rows[0][:4] == ("rs1", "1", "rs4", "2") and r.rows[0][4:] == pytest.approx((0.75, 3))
    for a, r2, n in ka["self"]:
        assert F.query("plink_ld", EX, variant1=a, variant2=a, columns=["R2", "OBS_CT"]).rows[0] == pytest.approx((r2, n))
    for samples in (["SAMPLE1", "SAMPLE2"], [0, 1]):
        r = F.query("plink_ld", EX, variant1="rs1", variant2="rs2", samples=samples, columns=["R2", "D_PRIME", "OBS_CT"])
        assert list(r.rows[0]) == ka["subset_s1_s2"]
    r = F.query("plink_ld", data_path("all_missing.pgen"), variant1="rs_miss1", variant2="rs_miss2",
                columns=["R2", "D_PRIME", "OBS_CT"])
    assert list(r.rows[0]) == ka["all_missing"]
    for kw in (dict(pvar=data_path("pgen_example.pvar"), psam=data_path("pgen_example.psam")),
               dict(pvar=data_path("pgen_example.bim"))):
        assert F.query("plink_ld", EX, variant1="rs1", variant2="rs2", columns=["R2", "OBS_CT"],
                       **kw).rows[0] == pytest.approx((0.75, 3))


def test_ld_windowed_known_answers():
    ka = KA["plink_ld"]
    cols = ["ID_A", "ID_B", "R2", "D_PRIME", "OBS_CT"]
    r = F.query("plink_ld", EX, window_kb=1000, r2_threshold=0.0, columns=cols)
    for got, want in zip(r.sorted("ID_A", "ID_B"), ka["window_1000kb_r2_0"]):
        assert got[:2] == tuple(want[:2]) and got[2:] == pytest.approx(want[2:], rel=1e-12)
    pairs = lambda **kw: [list(x) for x in F.query("plink_ld", EX, columns=["ID_A", "ID_B"], **kw).sorted("ID_A", "ID_B")]
    assert pairs(window_kb=15, r2_threshold=0.0) == ka["window_15kb_pairs"]
    assert len(pairs(window_kb=5, r2_threshold=0.0)) == ka["window_5kb_count"]
    r = F.query("plink_ld", EX, window_kb=10000, r2_threshold=0.0, columns=["CHROM_A", "CHROM_B"])
    assert all(a == b for a, b in r.rows)
    r = F.query("plink_ld", EX, window_kb=10000, r2_threshold=0.0, inter_chr=True, columns=["CHROM_A", "CHROM_B"])
    assert len(r) == ka["inter_chr_count"] and sum(a != b for a, b in r.rows) == ka["inter_chr_cross_count"]
    assert pairs(window_kb=15, r2_threshold=0.0, inter_chr=True) == ka["inter_chr_15kb_pairs"]
    assert len(pairs(window_kb=1000, r2_threshold=0.5)) == ka["threshold_counts"]["0.5"]
    assert pairs(window_kb=1000, r2_threshold=0.5) == [["rs1", "rs2"], ["rs1", "rs3"]]
    assert len(pairs(window_kb=1000, r2_threshold=0.8)) == ka["threshold_counts"]["0.8"]
    assert len(pairs(window_kb=1000)) == ka["threshold_counts"]["default"]
    r = F.query("plink_ld", EX, region="1:10000-20000", r2_threshold=0.0, columns=["ID_A", "ID_B", "R2"])
    assert r.rows[0][:2] == ("rs1", "rs2") and r.rows[0][2] == pytest.approx(0.75, rel=1e-12)
    r = F.query("plink_ld", EX, window_kb=1000, r2_threshold=0.0, columns=["POS_A", "POS_B", "CHROM_A", "CHROM_B"])
    assert all(pa < pb for pa, pb, ca, cb in r.rows if ca == cb)
    big = data_path("large_example.pgen")
    exp = ka["large_example_region_1_100_1000_window_1kb"]
    r = F.query("plink_ld", big, region="1:100-1000", window_kb=1, r2_threshold=0.0, columns=["ID_A", "ID_B", "R2"])
    assert len(r) == exp["pairs"] and sorted(set(round(x, 12) for x in r.column("R2"))) == exp["distinct_r2"]
    assert len(set((a, b) for a, b, _ in r.rows)) == exp["pairs"]


def test_ld_windowed_matches_oracle_across_threads(oracle):
    """pca_example: 500 variants x 250 samples with missing calls; every pair inside a window vs the oracle."""
    path = data_path("pca_example.pgen")
    pg = oracle.Pgen(path)
    pv = oracle.load_pvar(data_path("pca_example.pvar"))
    want = {}
    kb = 50
    for a in range(pg.M):
        for b in range(a + 1, pg.M):
            if pv["chrom"][a] != pv["chrom"][b] or pv["pos"][b] - pv["pos"][a] > kb * 1000:
                break
            r2, dp, n = oracle.ld_stats(pg.ld_sums(a, b))
            if r2 is not None and r2 >= 0.05:
                want[(pv["id"][a], pv["id"][b])] = (r2, dp, n)
    assert len(want) > 50
    for threads in (1, 5):
        r = F.query("plink_ld", path, window_kb=kb, r2_threshold=0.05, threads=threads,
                    columns=["ID_A", "ID_B", "R2", "D_PRIME", "OBS_CT"])
        got = {(a, b): (r2, dp, n) for a, b, r2, dp, n in r.rows}
        assert got.keys() == want.keys()
        for k, (r2, dp, n) in want.items():
            assert got[k][2] == n and got[k][0] == r2 and got[k][1] == dp  # same sums, same double arithmetic
    mask = np.random.default_rng(8).random(pg.N) < 0.6
    names = oracle.load_psam(data_path("pca_example.psam"))["iid"]
    r = F.query("plink_ld", path, window_kb=20, r2_threshold=0.0, samples=[names[i] for i in np.flatnonzero(mask)],
                columns=["ID_A", "ID_B", "R2", "OBS_CT"])
    idx = {v: i for i, v in enumerate(pv["id"])}
    for a, b, r2, n in r.rows[:200]:
        e_r2, _, e_n = oracle.ld_stats(pg.ld_sums(idx[a], idx[b], include=mask.astype(np.uint8)))
        assert (r2, n) == (e_r2, e_n)


# ---- read_pfile (read_pfile_genotypes_counts.test, read_pfile_sample_counts_sparse.test) ----------

def test_read_pfile_counts_known_answers():
    ka = KA["read_pfile"]
    PFX = data_path("pgen_example")
    cts = lambda g: [g["hom_ref"], g["het"], g["hom_alt"], g["missing"]]
    r = F.query("read_pfile", PFX, genotypes="counts", columns=["ID", "genotypes"])
    assert r.types[1] == "STRUCT(hom_ref UINTEGER, het UINTEGER, hom_alt UINTEGER, missing UINTEGER)"
    assert {vid: cts(g) for vid, g in r.rows} == ka["variant_counts"]
    r = F.query("read_pfile", PFX, genotypes="counts", samples=["SAMPLE1", "SAMPLE3"], columns=["ID", "genotypes"])
    assert {vid: cts(g) for vid, g in r.rows} == ka["variant_counts_subset_s1_s3"]
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", columns=["IID", "genotypes"])
    assert {iid: cts(g) for iid, g in r.rows} == ka["sample_counts"]
    v = F.query("read_pfile", PFX, genotypes="counts", columns=["genotypes"])
    assert sum(g["het"] + g["hom_alt"] for (g,) in v.rows) == sum(g["het"] + g["hom_alt"] for _, g in r.rows)
    r = F.query("read_pfile", PFX, genotypes="counts", af_range={"max": 0.4}, columns=["ID", "genotypes"])
    assert [(vid, cts(g)) for vid, g in r.rows] == [("rs4", [2, 1, 1, 0])]
    r = F.query("read_pfile", PFX, genotypes="counts", variants=["rs1", "rs2"], columns=["ID", "genotypes"])
    assert {vid: cts(g) for vid, g in r.rows} == {k: ka["variant_counts"][k] for k in ("rs1", "rs2")}
    # sample orient under the same filters: per-sample tallies over the surviving variants only
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", af_range={"max": 0.4}, columns=["IID", "genotypes"])
    assert {iid: cts(g) for iid, g in r.rows} == {"SAMPLE1": [1, 0, 0, 0], "SAMPLE2": [1, 0, 0, 0],
                                                  "SAMPLE3": [0, 1, 0, 0], "SAMPLE4": [0, 0, 1, 0]}
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", samples=[3, 0], region="1:10000-20000",
                columns=["IID", "genotypes"])
    assert [(iid, cts(g)) for iid, g in r.sorted("IID")] == [("SAMPLE1", [1, 1, 0, 0]), ("SAMPLE4", [0, 0, 1, 1])]


def test_read_pfile_sample_counts_on_rare_small(oracle):
    ka = KA["read_pfile"]
    PFX = data_path("rare_small")
    for threads in (1, 4):
        r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", columns=["IID", "genotypes"], threads=threads)
        assert len(r) == 256
        tot = {k: sum(g[k] for _, g in r.rows) for k in ("hom_ref", "het", "hom_alt", "missing")}
        assert tot == ka["rare_small_totals"]
    pg = oracle.Pgen(PFX + ".pgen")
    want = pg.sample_counts()
    names = oracle.load_psam(PFX + ".psam")["iid"]
    got = dict(r.rows)
    for k, iid in enumerate(names):
        g = got[iid]
        assert [g["hom_ref"], g["het"], g["hom_alt"], g["missing"]] == [int(x) for x in want[k]]
    s = F.query("read_pfile", PFX, orient="sample", genotypes="stats", columns=["genotypes"])
    st = ka["rare_small_stats"]
    n = sum(g["n"] for (g,) in s.rows)
    assert n == st["sum_n"] and sum(g["carrier_count"] for (g,) in s.rows) == st["sum_carrier_count"]
    assert round(sum(g["het"] for (g,) in s.rows) / n, 6) == st["het_over_n_6dp"]
    one = s.rows[0][0]
    assert one["af"] == pytest.approx((one["het"] + 2 * one["hom_alt"]) / (2 * one["n"]))
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", include_genotypes=["het", "hom_alt"], columns=["IID"])
    assert len(r) == ka["rare_small_include_het_homalt_rows"]
    carriers = int(((want[:, 1] + want[:, 2]) > 0).sum())
    assert carriers == 256
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", include_genotypes=["hom_alt"], columns=["IID"])
    assert len(r) == int((want[:, 2] > 0).sum())
    e = ka["rare_small_empty_region"]
    r = F.query("read_pfile", PFX, orient="sample", genotypes="counts", region="chrZ:1-2", columns=["genotypes"])
    assert (len(r), sum(g["het"] for (g,) in r.rows), sum(g["hom_ref"] for (g,) in r.rows)) == (e["rows"], e["het"], e["hom_ref"])


def test_read_pfile_variant_orient_is_the_read_pgen_scan():
    exp = [[(None if g == -9 else g) for g in row] for row in KA["pgen_example_genotypes"]["matrix"]]
    r = F.query("read_pfile", data_path("pgen_example"), columns=["ID", "genotypes"])
    assert [g for _, g in r.sorted("ID")] == exp
    r = F.query("read_pfile", data_path("pgen_example"), region="1:20000-30000", genotypes="list", columns=["ID", "genotypes"])
    assert r.types[1] == "TINYINT[]" and dict(r.rows) == {"rs2": exp[1], "rs3": exp[2]}
    big = F.query("read_pfile", data_path("large_example"), region="2:1-100000", columns=["CHROM", "genotypes"], threads=3)
    assert len(big) == 1000 and set(big.column("CHROM")) == {"2"}


@pytest.mark.parametrize("threads", [1, 4])
def test_read_pfile_shards_reproduce_the_whole_file(threads):
    """read_pfile_list_shards.test sections 2, 4, 5, 6, 7 and read_pfile_list.test: three variant-disjoint
    shards of large_example, read as a list, give the whole file's rows in every genotype mode."""
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    whole = data_path("large_example")
    for kw in ({}, {"genotypes": "list"}, {"genotypes": "counts"}, {"genotypes": "stats"}, {"dosages": True},
               {"phased": True}, {"region": "1:5000-50000"}, {"samples": ["SAMP7", "SAMP2"]},
               {"psam": shards[1] + ".psam", "genotypes": "counts"}):
        mf = F.query("read_pfile", shards, columns=["ID", "genotypes"], threads=threads, **kw)
        wf = F.query("read_pfile", whole, columns=["ID", "genotypes"], threads=threads, **kw)
        assert len(mf) == len(wf) == (451 if "region" in kw else 3000), kw
        assert sorted(mf.rows, key=lambda r: r[0]) == sorted(wf.rows, key=lambda r: r[0]), kw
    cols = ["ID"] + ["SAMP%d" % i for i in range(1, 9)]
    mf = F.query("read_pfile", shards, genotypes="columns", columns=cols, threads=threads)
    wf = F.query("read_pfile", whole, genotypes="columns", columns=cols, threads=threads)
    assert len(mf) == 3000 and sorted(mf.rows) == sorted(wf.rows)
    # filters fire on every shard (AF = 0.5, AC = 6 throughout this fixture)
    for kw, n in (({"af_range": {"max": 0.5}}, 3000), ({"af_range": {"min": 0.6}}, 0),
                  ({"ac_range": {"min": 2, "max": 10}}, 3000), ({"ac_range": {"min": 12}}, 0)):
        assert len(F.query("read_pfile", shards, columns=["ID", "genotypes"], threads=threads, **kw)) == n, kw
    # the same file twice: every row twice, values intact across the source boundary
    ex = data_path("pgen_example")
    twice = F.query("read_pfile", [ex, ex], columns=["ID", "genotypes"], threads=threads)
    once = F.query("read_pfile", ex, columns=["ID", "genotypes"])
    assert sorted(twice.rows) == sorted(once.rows + once.rows)
    assert len(F.query("read_pfile", [ex, ex], genotypes="columns", columns=["SAMPLE1"], threads=threads)) == 8
    assert len(F.query("read_pfile", [ex, ex], af_range={"max": 0.5}, columns=["genotypes"], threads=threads)) == 8
    big2 = F.query("read_pfile", [whole, whole], genotypes="counts", columns=["ID", "genotypes"], threads=threads)
    one = dict(F.query("read_pfile", whole, genotypes="counts", columns=["ID", "genotypes"]).rows)
    assert len(big2) == 6000 and all(g is not None and g == one[i] for i, g in big2.rows)


def test_read_pfile_sample_orient_counts_add_up_over_shards():
    """read_pfile_list.test:97-107 for the aggregate modes: the sources' variants concatenate, so every
    sample's tallies over the shards are its tallies over the whole file."""
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    whole = data_path("large_example")
    for kw in ({"genotypes": "counts"}, {"genotypes": "stats"}, {"genotypes": "counts", "region": "1:5000-50000"},
               {"genotypes": "counts", "samples": ["SAMP7", "SAMP2"]}, {"genotypes": "counts", "af_range": {"max": 0.5}},
               {"genotypes": "counts", "af_range": {"min": 0.6}}, {"genotypes": "counts", "include_genotypes": ["missing"]}):
        mf = F.query("read_pfile", shards, orient="sample", columns=["IID", "genotypes"], threads=3, **kw)
        wf = F.query("read_pfile", whole, orient="sample", columns=["IID", "genotypes"], **kw)
        assert sorted(mf.rows, key=lambda r: r[0]) == sorted(wf.rows, key=lambda r: r[0]), kw
    ex = data_path("pgen_example")
    twice = dict(F.query("read_pfile", [ex, ex], orient="sample", genotypes="counts", columns=["IID", "genotypes"]).rows)
    once = dict(F.query("read_pfile", ex, orient="sample", genotypes="counts", columns=["IID", "genotypes"]).rows)
    assert len(twice) == 4 and all({k: 2 * v for k, v in once[i].items()} == twice[i] for i in once)


def test_read_pfile_sample_orient_matrices(oracle):
    """read_pfile_orient.test:93-215, read_pfile_genotypes_columns.test:92-139, read_pfile_genotype_filter.test:64-93,
    read_pfile_dosage.test: one row per sample, its calls over the effective variants."""
    P = data_path("pfile_example")
    want = {"SAMPLE1": [0, 1, 2, 0], "SAMPLE2": [1, 1, None, 0], "SAMPLE3": [2, 0, 1, 1], "SAMPLE4": [None, 2, 0, 2]}
    r = F.query("read_pfile", P, orient="sample", columns=["IID", "genotypes"])
    assert r.types[1] == "TINYINT[4]" and dict(r.rows) == want
    r = F.query("read_pfile", P, orient="sample", genotypes="list", columns=["IID", "genotypes"])
    assert r.types[1] == "TINYINT[]" and dict(r.rows) == want
    r = F.query("read_pfile", P, orient="sample", samples=["SAMPLE1", "SAMPLE3"], columns=["IID", "genotypes"])
    assert dict(r.rows) == {k: want[k] for k in ("SAMPLE1", "SAMPLE3")}
    r = F.query("read_pfile", P, orient="sample", variants=["rs1", "rs2"], columns=["IID", "genotypes"])
    assert r.types[1] == "TINYINT[2]" and dict(r.rows)["SAMPLE1"] == [0, 1]
    r = F.query("read_pfile", P, orient="sample", region="1:10000-30000", variants=["rs1", "rs3"], samples=["SAMPLE1"],
                columns=["IID", "genotypes"])
    assert r.rows == [("SAMPLE1", [0, 2])]
    r = F.query("read_pfile", P, orient="sample", genotypes="columns", columns=["IID", "rs1", "rs2", "rs3", "rs4"])
    assert {row[0]: list(row[1:]) for row in r.rows} == want
    r = F.query("read_pfile", P, orient="sample", genotypes="columns", region="1:10000-20000", columns=["IID", "rs2", "rs1"])
    assert dict((a, (b, c)) for a, b, c in r.rows)["SAMPLE1"] == (1, 0)
    r = F.query("read_pfile", P, orient="sample", genotypes="struct", columns=["IID", "genotypes"])
    assert {iid: [g[f"rs{i}"] for i in range(1, 5)] for iid, g in r.rows} == want
    # genotype filter: a sample stays if any call is allowed; calls outside read as NULL where not all pass
    f = lambda **kw: sorted(F.query("read_pfile", P, orient="sample", variants=["rs1"], columns=["IID"], **kw).column("IID"))
    assert f(include_genotypes=["hom_ref", "hom_alt"]) == ["SAMPLE1", "SAMPLE3"]
    assert f(include_genotypes=["hom_alt", "missing"]) == ["SAMPLE3", "SAMPLE4"]
    assert f(include_genotypes=[" Hom_Alt ", "MISSING"]) == ["SAMPLE3", "SAMPLE4"]
    assert f(genotype_range={"min": 1, "max": 2}) == ["SAMPLE2", "SAMPLE3"]
    r = F.query("read_pfile", P, orient="sample", include_genotypes=["het"], columns=["IID", "genotypes"])
    assert dict(r.rows) == {"SAMPLE1": [None, 1, None, None], "SAMPLE2": [1, 1, None, None], "SAMPLE3": [None, None, 1, 1]}
    assert len(F.query("read_pfile", data_path("all_missing"), orient="sample", include_genotypes=["het", "hom_alt"],
                       columns=["IID"])) == 0
    # af_range picks the effective variants at bind, so it shows in the type
    r = F.query("read_pfile", P, orient="sample", af_range={"max": 0.4}, columns=["IID", "genotypes"])
    assert r.types[1].startswith("TINYINT[") and len(r) == 4
    # dosages
    dz = data_path("dosage_example")
    pg = oracle.Pgen(dz + ".pgen")
    d = F.query("read_pfile", dz, orient="sample", dosages=True, columns=["IID", "genotypes"])
    assert d.types[1] == f"DOUBLE[{pg.M}]"
    cols = np.stack([pg.dosage(v) for v in range(pg.M)], axis=1)
    for k, (iid, g) in enumerate(sorted(d.rows)):
        assert [(-9.0 if x is None else x) for x in g] == cols[k].tolist()


def test_read_pfile_variant_orient_equals_read_pgen_in_every_mode():
    """read_pfile_genotypes*.test, read_pfile_filter.test, read_pfile_dosage.test, read_pfile_phased.test,
    read_pfile_pgi.test: variant orient is the read_pgen scan, so every mode and filter gives read_pgen's rows."""
    for name, extra in (("pgen_example", {}), ("dosage_example", {"dosages": True}), ("phased_example", {"phased": True})):
        pfx, pgen = data_path(name), data_path(name + ".pgen")
        for kw in ({}, {"genotypes": "list"}, {"genotypes": "struct"}, {"genotypes": "counts"}, {"genotypes": "stats"},
                   {"samples": [3, 1]}, {"af_range": {"max": 0.4}}, {"ac_range": {"min": 3}},
                   {"include_genotypes": ["het"]}, {"genotype_range": {"min": 1, "max": 2}}, {"variants": ["rs3", "rs1"]}):
            kw = dict(kw)
            if extra and ("include_genotypes" in kw or "genotype_range" in kw or kw.get("genotypes") in ("counts", "stats")):
                continue  # incompatible with dosages / phased by the reference's own rules
            kw.update(extra)
            a = F.query("read_pfile", pfx, columns=["ID", "genotypes"], **kw)
            b = F.query("read_pgen", pgen, columns=["ID", "genotypes"], **kw)
            assert a.types == b.types and sorted(a.rows, key=lambda r: r[0]) == sorted(b.rows, key=lambda r: r[0]), (name, kw)
        cols = ["ID", "SAMPLE1", "SAMPLE4"]
        a = F.query("read_pfile", pfx, genotypes="columns", columns=cols, **({"dosages": True} if "dosages" in extra else {}))
        b = F.query("read_pgen", pgen, genotypes="columns", columns=cols, **({"dosages": True} if "dosages" in extra else {}))
        assert sorted(a.rows) == sorted(b.rows)
    split = data_path("pgen_split")
    a = F.query("read_pfile", split, columns=["ID", "genotypes"])
    assert len(a) > 0 and sorted(a.rows) == sorted(F.query("read_pgen", split + ".pgen", columns=["ID", "genotypes"]).rows)


def test_read_pfile_sample_multifile_values():
    """read_pfile_sample_multifile.test: a sample's row over the shards is its rows over each shard, side by side;
    filters apply per shard and shrink the dimension."""
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    mf = dict(F.query("read_pfile", shards, orient="sample", genotypes="list", columns=["IID", "genotypes"]).rows)
    parts = [dict(F.query("read_pfile", sh, orient="sample", genotypes="list", columns=["IID", "genotypes"]).rows)
             for sh in shards]
    for iid in ("SAMP1", "SAMP5"):
        assert mf[iid][:1000] == parts[0][iid] and mf[iid][1000:2000] == parts[1][iid] and mf[iid][2000:] == parts[2][iid]
    kw = dict(samples=["SAMP1", "SAMP2", "SAMP3"], ac_range={"min": 2, "max": 4})
    r = F.query("read_pfile", shards, orient="sample", columns=["IID", "genotypes"], **kw)
    n_var = len(F.query("read_pfile", shards, columns=["ID"], **kw))
    assert r.types[1] == f"TINYINT[{n_var}]" and {len(g) for _, g in r.rows} == {n_var} and len(r) == 3
    ov = dict(F.query("read_pfile", shards[:2], orient="sample", genotypes="list", psam=shards[0] + ".psam",
                      columns=["IID", "genotypes"]).rows)
    au = dict(F.query("read_pfile", shards[:2], orient="sample", genotypes="list", columns=["IID", "genotypes"]).rows)
    assert ov == au and len(ov["SAMP1"]) == 2000


def test_read_pfile_genotype_orient(oracle):
    """read_pfile_genotype_orient.test, read_pfile_genotype_filter.test:150-200, read_pfile_list.test:28-60,
    read_pfile_list_shards.test section 3 and 7: one row per (variant, sample)."""
    P = data_path("pfile_example")
    want = {"rs1": [0, 1, 2, None], "rs2": [1, 1, 0, 2], "rs3": [2, None, 1, 0], "rs4": [0, 0, 1, 2]}
    for threads in (1, 4):
        r = F.query("read_pfile", P, orient="genotype", columns=["ID", "IID", "genotype"], threads=threads)
        assert len(r) == 16 and r.types[2] == "TINYINT"
        got = {}
        for vid, iid, g in r.rows:
            got.setdefault(vid, {})[iid] = g
        assert {v: [got[v][f"SAMPLE{i}"] for i in range(1, 5)] for v in got} == want
    r = F.query("read_pfile", P, orient="genotype", columns=["CHROM", "POS", "ID", "REF", "ALT", "FID", "IID", "SEX"])
    assert ("1", 10000, "rs1", "A", "G", "FAM001", "SAMPLE1", 1) in r.rows and len(r) == 16
    r = F.query("read_pfile", P, orient="genotype", samples=["SAMPLE3", "SAMPLE1"], columns=["ID", "IID", "genotype"])
    assert sorted(x[1:] for x in r.rows if x[0] == "rs1") == [("SAMPLE1", 0), ("SAMPLE3", 2)] and len(r) == 8
    f = lambda **kw: sorted(F.query("read_pfile", P, orient="genotype", variants=["rs1"], **kw).rows)
    assert f(include_genotypes=["het", "hom_alt"], columns=["IID", "genotype"]) == [("SAMPLE2", 1), ("SAMPLE3", 2)]
    assert sorted(f(include_genotypes=["het", "missing"], columns=["IID", "genotype"]), key=lambda x: x[0]) == [
        ("SAMPLE2", 1), ("SAMPLE4", None)]
    assert f(include_genotypes=["het", "hom_alt"], columns=["IID"]) == [("SAMPLE2",), ("SAMPLE3",)]
    ex = data_path("pgen_example")
    assert len(F.query("read_pfile", [ex, ex], orient="genotype", columns=["ID"])) == 32
    assert len(F.query("read_pfile", [ex, ex], orient="genotype", include_genotypes=["het"], columns=["ID"])) == 10
    r = F.query("read_pfile", [ex, ex], orient="genotype", columns=["CHROM", "POS", "IID", "genotype"])
    assert sorted(r.rows, key=lambda x: (x[1], x[2], x[3] is None, x[3]))[:4] == [
        ("1", 10000, "SAMPLE1", 0), ("1", 10000, "SAMPLE1", 0), ("1", 10000, "SAMPLE2", 1), ("1", 10000, "SAMPLE2", 1)]
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    whole = data_path("large_example")
    for threads in (1, 4):
        mf = F.query("read_pfile", shards, orient="genotype", columns=["ID", "IID", "genotype"], threads=threads)
        wf = F.query("read_pfile", whole, orient="genotype", columns=["ID", "IID", "genotype"], threads=threads)
        key = lambda x: (x[0], x[1])
        assert len(mf) == 24000 and sorted(mf.rows, key=key) == sorted(wf.rows, key=key)
    het_mf = F.query("read_pfile", shards, orient="genotype", include_genotypes=["het"], columns=["ID", "IID"], threads=3)
    het_wf = F.query("read_pfile", whole, orient="genotype", include_genotypes=["het"], columns=["ID", "IID"])
    assert len(het_mf) == 6000 and sorted(het_mf.rows) == sorted(het_wf.rows)
    # dosages: the scalar is a DOUBLE
    dz = data_path("dosage_example")
    pg = oracle.Pgen(dz + ".pgen")
    d = F.query("read_pfile", dz, orient="genotype", dosages=True, columns=["ID", "IID", "genotype"])
    assert d.types[2] == "DOUBLE" and len(d) == pg.M * pg.N


def test_read_pfile_sample_orient_over_shards_and_batches(tmp_path, gpu_lib, oracle):
    """read_pfile_list.test:97-107 and a file wider than one transpose tile: the sources' effective variants side by
    side in list order; every sample's row equals the whole file's column."""
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    whole = data_path("large_example")
    for kw in ({}, {"region": "2:1-60000"}, {"samples": ["SAMP7", "SAMP2"]}, {"genotypes": "list"}):
        mf = dict(F.query("read_pfile", shards, orient="sample", columns=["IID", "genotypes"], threads=3, **kw).rows)
        ids = {}
        for sh in shards:
            ids[sh] = F.query("read_pfile", sh, columns=["ID"], **({"region": kw["region"]} if "region" in kw else {})).column("ID")
        order = [i for sh in shards for i in ids[sh]]
        wf = F.query("read_pfile", whole, genotypes="columns", columns=["ID"] + sorted(mf), **{k: v for k, v in kw.items() if k == "region"})
        by_id = {row[0]: row[1:] for row in wf.rows}
        for col, iid in enumerate(sorted(mf)):
            assert mf[iid] == [by_id[i][col] for i in order], (kw, iid)
    ex = data_path("pgen_example")
    twice = dict(F.query("read_pfile", [ex, ex], orient="sample", columns=["IID", "genotypes"]).rows)
    once = dict(F.query("read_pfile", ex, orient="sample", columns=["IID", "genotypes"]).rows)
    assert all(twice[i] == once[i] + once[i] for i in once) and len(twice) == 4
    # several tiles in both directions, against the oracle
    prefix = str(tmp_path / "wide")
    gpu_lib.synth_write_files(prefix, 300, 1003, 77, 0.04)
    pg = oracle.Pgen(prefix + ".pgen")
    r = F.query("read_pfile", prefix, orient="sample", genotypes="list", region="3:1-100000", threads=5,
                columns=["IID", "genotypes"])
    vsel = [v for v in range(300) if v // ((300 + 21) // 22) + 1 == 3]
    cols = np.stack([pg.geno(v) for v in vsel], axis=1)
    assert len(r) == 1003
    for iid, g in r.rows:
        assert [(-9 if x is None else x) for x in g] == cols[int(iid[1:])].tolist()


# ---- every function over a file that spans many claims and several pass batches, against the ORACLE --------------

@pytest.fixture(scope="module")
def midsize(tmp_path_factory, gpu_lib, oracle):
    """40,000 variants x 3,001 samples (ten 4,096-variant claims per thread round; ragged row tail), 22 chromosomes:
    the oracle's scan of the file on disk is what the shells are held to (the library's own calls only where the
    oracle has no counterpart at this size)."""
    prefix = str(tmp_path_factory.mktemp("midsize") / "mid")
    gpu_lib.synth_write_files(prefix, 40_000, 3001, 20260807, 0.03)
    ds = gpu_lib.Dataset.open(prefix + ".pgen")
    pg = oracle.Pgen(prefix + ".pgen")
    return prefix, ds, pg


@pytest.mark.parametrize("threads", [1, 6])
def test_shells_agree_with_the_oracle_across_device_batches(midsize, gpu_lib, oracle, threads, monkeypatch):
    prefix, ds, pg = midsize
    monkeypatch.setenv("PGH_TALLY_BATCH", "12288")  # the range's tally pass in four batches
    path = prefix + ".pgen"
    m, n = 40_000, 3001
    counts = pg.scan_counts_mt(0, m, 4).astype(np.int64)
    pos_key = lambda chrom, pos: (int(chrom) - 1) * ((m + 21) // 22) + pos // 100 - 1  # variant index from the pvar text
    r = F.query("plink_freq", path, counts=True, threads=threads,
                columns=["CHROM", "POS", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT", "OBS_CT", "ALT_FREQ"])
    assert len(r) == m
    got = np.zeros((m, 4), dtype=np.int64)
    seen = np.zeros(m, dtype=bool)
    for chrom, pos, a, b, c, d, obs, af in r.rows:
        v = pos_key(chrom, pos)
        got[v] = (a, b, c, d)
        seen[v] = True
        assert obs == 2 * (a + b + c) and af == (b + 2 * c) / (2 * (a + b + c))
    assert seen.all() and np.array_equal(got, counts)
    r = F.query("plink_missing", path, threads=threads, columns=["CHROM", "POS", "MISSING_CT", "OBS_CT"])
    assert len(r) == m and all(counts[pos_key(c, p), 3] == mc and oc == n - mc for c, p, mc, oc in r.rows)
    r = F.query("plink_missing", path, mode="sample", threads=threads, columns=["IID", "MISSING_CT", "OBS_CT"])
    miss = pg.missing_per_sample()
    bad = [(iid, mc, oc, int(miss[int(iid[1:])])) for iid, mc, oc in r.rows if miss[int(iid[1:])] != mc or oc != m - mc]
    assert len(r) == n and not bad, (len(bad), bad[:8], int(miss.sum()), int(counts[:, 3].sum()))
    r = F.query("plink_hardy", path, threads=threads, region="7:1-100000000",
                columns=["POS", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "P_HWE"])
    per = (m + 21) // 22
    assert len(r) == per
    for pos, a, b, c, p in r.rows:
        i = pos // 100 - 1
        assert (a, b, c) == tuple(counts[6 * per + i, :3])
        assert p == pytest.approx(oracle.hardy_from_counts(counts[6 * per + i])[2], rel=1e-6)
    r = F.query("read_pgen", path, genotypes="counts", ac_range={"min": 3000}, threads=threads, columns=["POS", "CHROM", "genotypes"])
    keep = np.flatnonzero(counts[:, 1] + 2 * counts[:, 2] >= 3000)
    assert len(r) == len(keep)
    assert sorted(pos_key(c, p) for p, c, _ in r.rows) == list(keep)
    r = F.query("read_pfile", prefix, orient="sample", genotypes="counts", threads=threads, columns=["IID", "genotypes"])
    sc = pg.sample_counts()
    assert all([g["hom_ref"], g["het"], g["hom_alt"], g["missing"]] == [int(x) for x in sc[int(iid[1:])]] for iid, g in r.rows)
    w = np.linspace(-1.0, 1.0, m)
    r = F.query("plink_score", path, weights=[float(x) for x in w], threads=threads, columns=["IID", "ALLELE_CT", "SCORE_SUM"])
    pick = np.arange(0, m, 7)  # (the oracle's per-sample loop over every variant would take minutes: every 7th here,
    s, d, ac = ds.score(np.arange(m), w)  #  the full list against the library, whose parity tests face the oracle)
    r7 = F.query("plink_score", path, weights=[float(w[v]) if v % 7 == 0 else 0.0 for v in range(m)], threads=threads,
                 columns=["IID", "ALLELE_CT", "SCORE_SUM"])  # zero weights are dropped at bind
    so, do, aco = oracle.score(pg, pick, w[pick])
    for iid, a, ssum in r7.rows[::37]:
        k = int(iid[1:])
        assert a == aco[k] and ssum == pytest.approx(so[k, 0], rel=1e-9, abs=1e-9)
    for iid, a, ssum in r.rows[::37]:
        k = int(iid[1:])
        assert a == ac[k] and ssum == pytest.approx(s[k, 0], rel=1e-9, abs=1e-9)
    r = F.query("plink_ld", path, region="3:100-3000", window_kb=1, r2_threshold=0.0, threads=threads,
                columns=["POS_A", "POS_B", "OBS_CT"])
    base = 2 * per
    pairs = [(base + a, base + b) for a in range(30) for b in range(a + 1, min(30, a + 11))]
    sums = ds.ld_pairs([p[0] for p in pairs], [p[1] for p in pairs])
    got_n = {(pa // 100 - 1 + base, pb // 100 - 1 + base): n_obs for pa, pb, n_obs in r.rows}
    valid = {p: int(s6[0]) for p, s6 in zip(pairs, sums) if s6[0] >= 2}
    # rows exist only for pairs with a defined r2 (neither side monomorphic among the shared calls)
    assert set(got_n) <= set(valid) and len(got_n) >= 0.9 * len(valid)
    assert all(valid[p] == nn for p, nn in got_n.items())


# ---- read_pgen_filter.test, mirrored query by query ---------------------------------------------

def test_read_pgen_filter_test_mirror():
    q = lambda cols, **kw: F.query("read_pgen", EX, columns=cols, **kw)
    ids = lambda **kw: q(["ID"], **kw).column("ID")
    assert ids(af_range={"max": 0.4}) == ["rs4"]
    assert len(ids(af_range={"min": 0.5, "max": 0.5})) == 3
    assert ids(af_range={"min": 0.9}) == []
    assert ids(ac_range={"min": 4}) == ["rs2"]
    assert len(ids(ac_range={"max": 3})) == 3
    assert ids(af_range={"max": 0.4}, ac_range={"min": 3}) == ["rs4"]
    assert q(["ID", "genotypes"], af_range={"max": 0.4}).rows == [("rs4", [0, 0, 1, 2])]
    N = None
    g = lambda **kw: dict(q(["ID", "genotypes"], **kw).rows)
    assert g(genotype_range={"min": 1}) == {"rs1": [N, 1, 2, N], "rs2": [1, 1, N, 2], "rs3": [2, N, 1, N], "rs4": [N, N, 1, 2]}
    assert g(genotype_range={"min": 1, "max": 1}) == {"rs1": [N, 1, N, N], "rs2": [1, 1, N, N], "rs3": [N, N, 1, N],
                                                      "rs4": [N, N, 1, N]}
    assert g(genotype_range={"min": 2, "max": 2}) == {"rs1": [N, N, 2, N], "rs2": [N, N, N, 2], "rs3": [2, N, N, N],
                                                      "rs4": [N, N, N, 2]}
    assert g(af_range={"max": 0.4}, genotype_range={"min": 1}) == {"rs4": [N, N, 1, 2]}
    assert len(F.query("read_pgen", data_path("all_missing.pgen"), genotype_range={"min": 0, "max": 2}, columns=["ID"])) == 0
    assert g(include_genotypes=["het", "hom_alt"]) == g(genotype_range={"min": 1})
    assert g(include_genotypes=["hom_ref", "hom_alt"]) == {"rs1": [0, N, 2, N], "rs2": [N, N, 0, 2], "rs3": [2, N, N, 0],
                                                           "rs4": [0, 0, N, 2]}
    for kw, msg in [(dict(af_range={"min": 0.8, "max": 0.2}), "min (0.8) > max (0.2)"),
                    (dict(af_range={"max": 1.5}), "out of range"), (dict(ac_range={"min": -1}), "out of range"),
                    (dict(af_range={"minimum": 0.1}), "unknown field 'minimum'"),
                    (dict(genotype_range={"min": 2, "max": 0}), "min (2) > max (0)"),
                    (dict(genotype_range={"max": 3}), "out of range"),
                    (dict(genotype_range={"min": 1}, dosages=True), "genotype_range is incompatible with dosages"),
                    (dict(include_genotypes=["het"], genotype_range={"min": 1}),
                     "specify only one of include_genotypes or genotype_range"),
                    (dict(include_genotypes=["het"], dosages=True), "incompatible with dosages"),
                    (dict(include_genotypes=["carrier"]), "unknown category 'carrier'")]:
        with pytest.raises(F.InvalidInputException) as e:
            q(["ID"], **kw)
        assert msg in str(e.value), (kw, str(e.value))


# ---- plinking_max_threads.test ---------------------------------------------------------------

@pytest.mark.parametrize("cap", [1, 2, 0, None])
def test_plinking_max_threads_test_mirror(cap):
    big, pfx = data_path("large_example.pgen"), data_path("large_example")
    st = {} if cap is None else {"plinking_max_threads": cap}
    r = F.query("read_pfile", pfx, columns=["ID"], settings=st, threads=8)
    assert len(r) == 3000 and len(set(r.column("ID"))) == 3000
    assert len(F.query("read_pgen", big, columns=["ID"], settings=st, threads=8)) == 3000
    fr = F.query("plink_freq", big, columns=["ALT_FREQ"], settings=st, threads=8)
    assert len(fr) == 3000 and all(0.0 <= x <= 1.0 for x in fr.column("ALT_FREQ"))
    assert len(F.query("plink_hardy", big, columns=["ID"], settings=st, threads=8)) == 3000
    mv = F.query("plink_missing", big, columns=["MISSING_CT"], settings=st, threads=8)
    ms = F.query("plink_missing", big, mode="sample", columns=["MISSING_CT"], settings=st, threads=8)
    assert len(mv) == 3000 and len(ms) == 8
    assert sum(mv.column("MISSING_CT")) == sum(ms.column("MISSING_CT"))  # cross-function consistency under the cap
    if cap:
        assert r.threads <= cap


def test_negative_thread_cap_is_rejected():
    with pytest.raises(F.InvalidInputException) as e:
        F.query("plink_freq", EX, columns=["ID"], settings={"plinking_max_threads": -1})
    assert "plinking_max_threads must be non-negative" in str(e.value)


def test_read_pgen_genotypes_test_mirror():
    """read_pgen_genotypes.test: mode names are case-insensitive, ARRAY == LIST element for element,
    orphan files and explicit companions."""
    for mode, typ in (("LIST", "TINYINT[]"), ("Auto", "TINYINT[4]"), ("ARRAY", "TINYINT[4]")):
        assert F.query("read_pgen", EX, genotypes=mode, columns=["genotypes"]).types == [typ]
    for path, kw in ((EX, {}), (data_path("large_example.pgen"), {}), (EX, {"samples": [0, 2]})):
        a = dict(F.query("read_pgen", path, genotypes="array", columns=["ID", "genotypes"], **kw).rows)
        l = dict(F.query("read_pgen", path, genotypes="list", columns=["ID", "genotypes"], **kw).rows)
        assert a == l and len(a) in (4, 3000)
    assert dict(F.query("read_pgen", EX, genotypes="list", samples=[3], columns=["ID", "genotypes"]).rows)["rs2"] == [2]
    orphan = data_path("pgen_orphan.pgen")
    assert F.query("read_pgen", orphan, genotypes="list", columns=["genotypes"]).types == ["TINYINT[]"]
    assert F.query("read_pgen", orphan, genotypes="array", columns=["genotypes"]).types == ["TINYINT[4]"]
    assert len(F.query("read_pgen", orphan, genotypes="list", samples=[0, 2], columns=["genotypes"]).rows[0][0]) == 2
    for kw in (dict(pvar=data_path("pgen_example.pvar"), psam=data_path("pgen_example.psam")),
               dict(pvar=data_path("pgen_example.bim"))):
        assert dict(F.query("read_pgen", EX, genotypes="list", columns=["ID", "genotypes"], **kw).rows)["rs1"] == [0, 1, 2, None]


def test_pvar_text_is_parsed_once_per_file_version(tmp_path, gpu_lib):
    """Bind re-uses the parsed .pvar columns of a file it has seen (same path, mtime, size) instead of re-reading
    the text (the reference's bind is dominated by LoadVariantMetadata, src/plink_common.cpp:171-375); a rewritten
    file is parsed again."""
    import shutil
    import time

    m, n = 120_000, 64
    prefix = str(tmp_path / "meta")
    gpu_lib.synth_write_files(prefix, m, n, 3, 0.01)
    first = F.query("plink_freq", prefix + ".pgen", columns=["ID", "ALT_FREQ"], threads=2)
    again = F.query("plink_freq", prefix + ".pgen", columns=["ID", "ALT_FREQ"], threads=2)
    assert sorted(first.rows) == sorted(again.rows) and len(first) == m
    assert again.timing_ms["bind"] < 0.5 * first.timing_ms["bind"], (first.timing_ms, again.timing_ms)
    # a new version of the companion file: ids change, the cache must not serve the old ones
    with open(prefix + ".pvar") as f:
        text = f.read()
    time.sleep(0.01)
    with open(prefix + ".pvar", "w") as f:
        f.write(text.replace("\tsv", "\tqv"))
    renamed = F.query("plink_freq", prefix + ".pgen", columns=["ID"], threads=2)
    assert all(r[0].startswith("sv") for r in first.rows) and all(r[0].startswith("qv") for r in renamed.rows)
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_concurrent_first_binds_share_one_open(tmp_path, gpu_lib):
    """Several sessions binding the same unseen file at once: one of them moves it to HBM, the others wait for that
    entry of the dataset cache (the cache lock is not held across pgh_open, so a bind of ANOTHER file meanwhile is not
    blocked behind it); a file that cannot be opened leaves no entry behind for the next bind to wait on."""
    import shutil
    import threading

    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    gpu_lib.synth_write_files(a, 40_000, 2_000, 5, 0.02)
    gpu_lib.synth_write_files(b, 3_000, 500, 6, 0.02)
    out, errors = {}, []

    def run(key, path):
        try:
            out[key] = sorted(F.query("plink_freq", path + ".pgen", columns=["ID", "ALT_FREQ", "OBS_CT"], threads=2).rows)
        except Exception as e:  # noqa: BLE001 -- reported below
            errors.append((key, repr(e)))

    threads = [threading.Thread(target=run, args=(f"a{i}", a)) for i in range(4)]
    threads += [threading.Thread(target=run, args=(f"b{i}", b)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    assert len(out["a0"]) == 40_000 and all(out[f"a{i}"] == out["a0"] for i in range(4))
    assert len(out["b0"]) == 3_000 and out["b1"] == out["b0"]
    # a failing open (truncated body) twice in a row: the second bind must not hang on a stale in-flight entry
    bad = str(tmp_path / "bad")
    for ext in (".pvar", ".psam"):
        shutil.copy(b + ext, bad + ext)
    with open(b + ".pgen", "rb") as f:
        body = f.read()
    with open(bad + ".pgen", "wb") as f:
        f.write(body[: len(body) // 2])
    for _ in range(2):
        with pytest.raises(Exception):
            F.query("plink_freq", bad + ".pgen", columns=["ID", "ALT_FREQ"], threads=2)
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.parametrize("threads", [1, 6, 1])
def test_host_calls_next_to_scan_threads_repeat_exactly(midsize, gpu_lib, threads):
    """The regression test of the stream-ordered-pool defect (profiles/r02_async_pool_ab.txt): with
    its scratch from hipMallocAsync, pgh_missing_per_sample returned sums 9 % short about once in thirty calls made
    right after table-function queries (scan threads coming and going).  Scratch now lives in per-thread blocks from
    hipMalloc; every repetition must be identical, and equal to the per-variant tallies' total."""
    prefix, ds, _ = midsize
    path = prefix + ".pgen"
    want = int(ds.counts_range()[:, 3].astype(np.int64).sum())
    for _ in range(2):
        F.query("plink_freq", path, counts=True, threads=threads, columns=["CHROM", "POS", "MISSING_CT", "ALT_FREQ"])
        F.query("plink_missing", path, threads=threads, columns=["CHROM", "POS", "MISSING_CT", "OBS_CT"])
        sums = [int(ds.missing_per_sample().astype(np.int64).sum()) for _ in range(6)]
        assert sums == [want] * 6, sums
        F.query("plink_missing", path, mode="sample", threads=threads, columns=["IID", "MISSING_CT"])
        cls = [ds.sample_counts().astype(np.int64).sum(axis=0) for _ in range(3)]
        assert all(np.array_equal(c, cls[0]) for c in cls) and int(cls[0][3]) == want


def test_read_pgen_chunk_pipeline_gives_the_same_rows(gpu_lib, tmp_path, monkeypatch):
    """PLINKING_UNPACK_PIPELINE=1: a scan thread keeps chunk k + 1 on its way (pgh_reader_unpack_start) while it fills
    its output from chunk k.  Same rows as one chunk at a time, filters and subsets included."""
    prefix = str(tmp_path / "pipe")
    m, n = 9000, 301
    gpu_lib.synth_write_files(prefix, m, n, 20260807, 0.05)
    path = prefix + ".pgen"
    calls = [dict(genotypes="list"), dict(genotypes="array", samples=[0, 5, 299]),
             dict(genotypes="list", af_range={"min": 0.3}), dict(genotypes="list", genotype_range={"min": 1})]
    for kw in calls:
        monkeypatch.setenv("PLINKING_UNPACK_PIPELINE", "0")
        a = F.query("read_pgen", path, threads=3, columns=["ID", "genotypes"], **kw)
        monkeypatch.setenv("PLINKING_UNPACK_PIPELINE", "1")
        b = F.query("read_pgen", path, threads=3, columns=["ID", "genotypes"], **kw)
        assert len(a) == len(b) > 0 and sorted(a.rows) == sorted(b.rows)


# ---- read_pfile_sample_counts_streaming.test, mirrored query by query -----------------------------------

def test_read_pfile_sample_counts_streaming_test_mirror():
    """orient := 'sample' with genotypes := 'counts' | 'stats' is a per-sample tally, not a matrix: the matrix guard
    does not apply to it, empty regions give zero rows of counts for every sample, include_genotypes skips samples."""
    S1 = data_path("shard1")
    cts = lambda g: (g["hom_ref"], g["het"], g["hom_alt"], g["missing"])
    q = lambda *a, **k: F.query("read_pfile", *a, orient="sample", **k)
    r = q(S1, genotypes="counts", columns=["IID", "genotypes"])  # :12-16
    assert cts(dict(r.rows)["SAMP1"]) == (250, 250, 250, 250)
    assert (sum(g["het"] for _, g in r.rows), sum(g["missing"] for _, g in r.rows)) == (2000, 2000)  # :18-22
    tiny = {"plinking_max_matrix_elements": 1}  # :26-33
    r = q(S1, genotypes="counts", columns=["genotypes"], settings=tiny)
    assert (len(r), sum(g["het"] for (g,) in r.rows)) == (8, 2000)
    with pytest.raises(Exception, match="plinking_max_matrix_elements"):  # :36-39 (array / list mode stays guarded)
        q(S1, columns=["IID"], settings=tiny)
    r = q(S1, genotypes="stats", columns=["IID", "genotypes"])  # :45-50
    g = dict(r.rows)["SAMP1"]
    assert (g["n"], g["af"], g["maf"], g["carrier_count"], round(g["het_rate"], 6)) == (750, 0.5, 0.5, 500, 0.333333)
    r = q([S1, data_path("shard2"), data_path("shard3")], genotypes="counts", columns=["genotypes"])  # :53-57
    assert tuple(sum(g[k] for (g,) in r.rows) for k in ("het", "missing", "hom_ref")) == (6000, 6000, 6000)
    r = q(S1, genotypes="counts", region="chr16:1-2", columns=["genotypes"])  # :60-64
    assert (len(r), sum(g["het"] for (g,) in r.rows), sum(g["hom_ref"] for (g,) in r.rows)) == (8, 0, 0)
    assert len(q(S1, genotypes="counts", include_genotypes=["het", "hom_alt"], columns=["IID"])) == 8  # :68-71
    AM = data_path("all_missing")
    assert len(q(AM, genotypes="counts", include_genotypes=["het"], columns=["IID"])) == 0  # :74-77
    r = q(AM, genotypes="counts", include_genotypes=["missing"], columns=["genotypes"])  # :79-82
    assert (len(r), sum(g["missing"] for (g,) in r.rows)) == (2, 4)
    r = q(S1, genotypes="counts", samples=["SAMP1", "SAMP2"], columns=["genotypes"])  # :85-88
    assert (len(r), sum(g["het"] for (g,) in r.rows)) == (2, 500)


# ---- edge_cases.test: the queries that reach the .pgen readers -------------------------------------------

def test_edge_cases_test_mirror():
    """edge_cases.test:58-104 (all-missing genotypes through read_pgen / read_pfile), :133-152 (genotype column
    types), :154-175 (a .pgen without a .psam), :178-199 (empty regions, unknown variants).  The read_pvar / read_psam
    queries of that file are the text parsers', outside this path."""
    AM = data_path("all_missing")
    r = F.query("read_pgen", AM + ".pgen", columns=["ID", "POS", "genotypes"])
    assert [(i, g) for i, _, g in r.sorted("POS")] == [("rs_miss1", [None, None]), ("rs_miss2", [None, None])]
    r = F.query("read_pfile", AM, columns=["ID", "POS", "genotypes"])
    assert len(r) == 2 and [(i, g) for i, _, g in r.sorted("POS")] == [("rs_miss1", [None, None]), ("rs_miss2", [None, None])]
    r = F.query("read_pfile", AM, orient="genotype", columns=["genotype"])
    assert len(r) == 4 and all(g is None for (g,) in r.rows)
    PE = data_path("pgen_example.pgen")
    assert F.query("read_pgen", PE, columns=["genotypes"]).types == ["TINYINT[4]"]
    assert F.query("read_pfile", data_path("pfile_example"), columns=["genotypes"]).types == ["TINYINT[4]"]
    assert F.query("read_pfile", data_path("pfile_example"), orient="genotype", columns=["genotype"]).types == ["TINYINT"]
    OR = data_path("pgen_orphan.pgen")
    r = F.query("read_pgen", OR, columns=["ID", "POS", "genotypes"])
    assert len(r) == 4 and [i for i, p, _ in r.rows if p == 10000] == ["rs1"] and len(r.rows[0][2]) == 4
    assert len(F.query("read_pfile", data_path("pfile_example"), region="99:1-100", columns=["ID"])) == 0
    assert len(F.query("read_pfile", data_path("pfile_example"), orient="genotype", region="99:1-100", columns=["ID"])) == 0
    with pytest.raises(Exception, match="variant 'nonexistent' not found"):
        F.query("read_pfile", data_path("pfile_example"), variants=["nonexistent"], columns=["ID"])


# ---- integration.test: the cross-reader consistency queries, with Python standing in for the SQL joins -----

def test_integration_test_mirror(oracle):
    """integration.test: read_pgen and read_pfile agree with each other and with the companion files -- row counts
    (:11-17), variant metadata (:23-41, :155-165), genotype arrays (:47-53, :108-113), column types (:65-83), the
    genotype orient against the array form (:89-102, :118-150) and against the .psam (:170-193)."""
    PG, PF = data_path("pgen_example.pgen"), data_path("pfile_example")
    meta = ["CHROM", "POS", "ID", "REF", "ALT"]
    pvar = oracle.load_pvar(data_path("pgen_example.pvar"))
    want_meta = sorted(zip(pvar["chrom"], pvar["pos"], pvar["id"], pvar["ref"], pvar["alt"]))
    a = F.query("read_pgen", PG, columns=meta + ["genotypes"])
    b = F.query("read_pfile", PF, columns=meta + ["genotypes"])
    assert len(a) == len(b) == len(want_meta) == 4
    assert sorted(tuple(r[:5]) for r in a.rows) == sorted(tuple(r[:5]) for r in b.rows) == [tuple(w) for w in want_meta]
    assert sorted((r[2], tuple(r[5])) for r in a.rows) == sorted((r[2], tuple(r[5])) for r in b.rows)
    assert a.types == b.types == ["VARCHAR", "INTEGER", "VARCHAR", "VARCHAR", "VARCHAR", "TINYINT[4]"]
    assert [tuple(r[:5]) for r in a.sorted("CHROM", "POS")] == [("1", 10000, "rs1", "A", "G"), ("1", 20000, "rs2", "C", "T"),
                                                                ("1", 30000, "rs3", "G", "A"), ("2", 15000, "rs4", "T", "C")]
    g = F.query("read_pfile", PF, orient="genotype", columns=["FID", "IID", "ID", "genotype"])
    assert g.types[3] == "TINYINT" and len(g) == 16 and len({r[1] for r in g.rows}) == 4
    assert [(r[1], r[3]) for r in g.sorted("IID") if r[2] == "rs1"] == [("SAMPLE1", 0), ("SAMPLE2", 1), ("SAMPLE3", 2), ("SAMPLE4", None)]
    psam = oracle.load_psam(PF + ".psam")
    pos = {iid: k for k, iid in enumerate(sorted(psam["iid"]))}  # ROW_NUMBER() OVER (ORDER BY IID)
    arrays = {r[2]: r[5] for r in b.rows}
    assert all(arrays[vid][pos[iid]] == gt for _, iid, vid, gt in g.rows)
    fid_of = dict(zip(psam["iid"], psam["fid"]))
    assert [(r[0], r[1], r[3]) for r in g.sorted("IID") if r[2] == "rs1"] == [("FAM001", "SAMPLE1", 0), ("FAM001", "SAMPLE2", 1),
                                                                            ("FAM002", "SAMPLE3", 2), ("FAM002", "SAMPLE4", None)]
    assert {(r[0], r[1]) for r in g.rows} == {(fid_of[i], i) for i in psam["iid"]}


# ---- tutorial.test: every query of docs/tutorial.md that reaches a .pgen ------------------------------------

def test_tutorial_test_mirror():
    """tutorial.test sections 2-6 (section 1 is read_pvar / read_psam: the text parsers, outside this path)."""
    PG, PF, PS = data_path("pgen_example.pgen"), data_path("pfile_example"), data_path("pfile_example.psam")
    r = F.query("read_pgen", PG, columns=["CHROM", "POS", "ID", "REF", "ALT"])  # :43-49
    assert sorted(r.rows) == [("1", 10000, "rs1", "A", "G"), ("1", 20000, "rs2", "C", "T"), ("1", 30000, "rs3", "G", "A"),
                              ("2", 15000, "rs4", "T", "C")]
    g = F.query("read_pfile", PF, orient="genotype", columns=["CHROM", "POS", "ID", "IID", "genotype"])  # :52-72
    want = {"rs1": [0, 1, 2, None], "rs2": [1, 1, 0, 2], "rs3": [2, None, 1, 0], "rs4": [0, 0, 1, 2]}
    where = {"rs1": ("1", 10000), "rs2": ("1", 20000), "rs3": ("1", 30000), "rs4": ("2", 15000)}
    assert g.sorted("ID", "IID") == [(*where[v], v, f"SAMPLE{k + 1}", gt) for v in sorted(want) for k, gt in enumerate(want[v])]
    alt = {}
    for *_, iid, gt in g.rows:  # :75-85
        alt[iid] = alt.get(iid, 0) + (gt or 0)
    assert alt == {"SAMPLE1": 3, "SAMPLE2": 2, "SAMPLE3": 4, "SAMPLE4": 4}
    assert sorted((v, i) for *_, v, i, gt in g.rows if gt == 1) == [("rs1", "SAMPLE2"), ("rs2", "SAMPLE1"), ("rs2", "SAMPLE2"),
                                                                     ("rs3", "SAMPLE3"), ("rs4", "SAMPLE3")]  # :88-98
    r = F.query("plink_missing", PG, columns=["ID", "MISSING_CT", "OBS_CT", "F_MISS"])  # :102-109
    assert r.sorted("ID") == [("rs1", 1, 3, 0.25), ("rs2", 0, 4, 0.0), ("rs3", 1, 3, 0.25), ("rs4", 0, 4, 0.0)]
    r = F.query("plink_missing", PG, mode="sample", psam=PS, columns=["IID", "MISSING_CT", "OBS_CT", "F_MISS"])  # :112-120
    assert r.sorted("IID") == [("SAMPLE1", 0, 4, 0.0), ("SAMPLE2", 1, 3, 0.25), ("SAMPLE3", 0, 4, 0.0), ("SAMPLE4", 1, 3, 0.25)]
    freq = F.query("plink_freq", PG, columns=["ID", "ALT_FREQ", "OBS_CT"])  # :123-130
    assert freq.sorted("ID") == [("rs1", 0.5, 6), ("rs2", 0.5, 8), ("rs3", 0.5, 6), ("rs4", 0.375, 8)]
    r = F.query("plink_freq", PG, counts=True, columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"])  # :133-140
    assert r.sorted("ID") == [("rs1", 1, 1, 1, 1), ("rs2", 1, 2, 1, 0), ("rs3", 1, 1, 1, 1), ("rs4", 2, 1, 1, 0)]
    hw = F.query("plink_hardy", PG, columns=["ID", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "P_HWE"])  # :143-151
    assert [(a, b, c, d, round(p, 4)) for a, b, c, d, p in hw.sorted("ID")] == [("rs1", 1, 1, 1, 1.0), ("rs2", 1, 2, 1, 1.0),
                                                                                ("rs3", 1, 1, 1, 1.0), ("rs4", 2, 1, 1, 0.4286)]
    assert dict((r[0], r[4]) for r in hw.rows)["rs4"] == pytest.approx(0.4285714285714286, rel=1e-12)  # :154-167
    r = F.query("plink_ld", PG, variant1="rs1", variant2="rs2", columns=["ID_A", "ID_B", "R2", "OBS_CT"])  # :171-176
    assert [(a, b, round(x, 4), n) for a, b, x, n in r.rows] == [("rs1", "rs2", 0.75, 3)]
    r = F.query("plink_ld", PG, r2_threshold=0.0, columns=["ID_A", "ID_B", "R2", "D_PRIME", "OBS_CT"])  # :179-187
    assert [(a, b, round(x, 4), round(d, 4), n) for a, b, x, d, n in r.sorted("ID_A", "ID_B")] == [
        ("rs1", "rs2", 0.75, 0.5, 3), ("rs1", "rs3", 1.0, 1.0, 2), ("rs2", "rs3", 0.25, 0.3333, 3)]
    r = F.query("plink_ld", PG, r2_threshold=0.0, inter_chr=True, columns=["ID_A", "ID_B", "R2", "OBS_CT"])  # :190-200
    assert [(a, b, round(x, 4), n) for a, b, x, n in r.sorted("ID_A", "ID_B")] == [
        ("rs1", "rs2", 0.75, 3), ("rs1", "rs3", 1.0, 2), ("rs1", "rs4", 0.75, 3), ("rs2", "rs3", 0.25, 3),
        ("rs2", "rs4", 0.1818, 4), ("rs3", "rs4", 1.0, 3)]
    r = F.query("plink_score", PG, psam=PS, weights=[0.5, -0.3, 1.2, 0.8], columns=["IID", "SCORE_SUM", "SCORE_AVG"])  # :204-212
    got = {i: (s, a) for i, s, a in r.rows}
    for iid, s, a in (("SAMPLE1", 2.1, 0.2625), ("SAMPLE2", 1.4, 0.175), ("SAMPLE3", 3.0, 0.375), ("SAMPLE4", 1.5, 0.1875)):
        assert got[iid] == (pytest.approx(s, abs=1e-12), pytest.approx(a, abs=1e-12))
    r = F.query("plink_score", PG, psam=PS, columns=["IID", "SCORE_SUM", "SCORE_AVG"],  # :215-229
                weights=[{"id": "rs1", "allele": "G", "weight": 0.5}, {"id": "rs2", "allele": "T", "weight": -0.3},
                         {"id": "rs4", "allele": "C", "weight": 0.8}])
    assert [(i, round(s, 2), round(a, 4)) for i, s, a in r.sorted("IID")] == [
        ("SAMPLE1", -0.3, -0.05), ("SAMPLE2", 0.2, 0.0333), ("SAMPLE3", 1.8, 0.3), ("SAMPLE4", 1.5, 0.25)]
    r = F.query("plink_freq", data_path("large_example.pgen"), region="1:1-50000", columns=["ALT_FREQ"])  # :249-262
    assert len(r) == 500 and round(sum(x for (x,) in r.rows) / 500, 4) == 0.5


# ---- every `statement error` of the reference's *_negative.test files (tests/golden/negative_cases.json) ----------

def _all_negative_cases():
    import test_table_functions_cpu as cpu

    return cpu.negative_cases()


@pytest.mark.parametrize("case", _all_negative_cases(), ids=lambda c: c["source"].split("/")[-1])
def test_reference_negative_cases_on_the_device(case):
    """The twin of test_table_functions_cpu.py::test_reference_negative_cases with a device present: no case is
    skipped, including the checks that sit behind the open of the file."""
    import test_table_functions_cpu as cpu

    cpu.run_negative_case(case, have_device=True)
