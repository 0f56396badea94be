"""Pins the CPU oracle (oracle/) to the reference's own known-answer tests.

Every expected value comes from tests/golden/known_answers.json, which
transcribes the reference's sqllogictest goldens (test/sql/*.test); inputs are
the reference's fixture files under tests/golden/data/.  CPU only.
"""

import json
import math
import os

import numpy as np
import pytest

from conftest import data_path

with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
    KA = json.load(f)


def r6(x):
    return round(x, 6)


def test_decode_matrix(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    assert (pg.M, pg.N) == (4, 4)
    got = [pg.geno(v).tolist() for v in range(4)]
    assert got == KA["pgen_example_genotypes"]["matrix"]


def test_split_index_decodes_like_inline(oracle):
    a = oracle.Pgen(data_path("pgen_example.pgen"))
    b = oracle.Pgen(data_path("pgen_split.pgen"))
    for v in range(4):
        assert a.geno(v).tolist() == b.geno(v).tolist()


def test_freq_known_answers(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    ka = KA["pgen_example_freq"]
    for v in range(4):
        c = pg.counts(v)
        assert c.tolist() == ka["counts"][v]
        af, obs = oracle.freq_from_counts(c)
        assert af == ka["alt_freq"][v] and obs == ka["obs_ct"][v]
    # AVG(ALT_FREQ) = 0.46875 (plink_freq.test:275-278)
    assert np.mean(ka["alt_freq"]) == 0.46875


def test_freq_subset(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    inc = np.array([1, 0, 1, 0], dtype=np.uint8)
    ka = KA["pgen_example_freq_subset_0_2"]
    ids = KA["pgen_example_freq"]["ids"]
    for name, exp in ka["alt_freq"].items():
        af, obs = oracle.freq_from_counts(pg.counts(ids.index(name), inc))
        assert af == exp and obs == ka["obs_ct"]
    assert pg.counts(1, inc).tolist() == ka["rs2_counts"]


def test_all_missing(oracle):
    pg = oracle.Pgen(data_path("all_missing.pgen"))
    for v in range(2):
        c = pg.counts(v)
        assert c.tolist() == KA["all_missing_freq"]["counts"][v]
        assert oracle.freq_from_counts(c) == (None, 0)
        assert oracle.hardy_from_counts(c) is None
    assert pg.missing_per_sample().tolist() == [2, 2]


def test_large_example(oracle):
    pg = oracle.Pgen(data_path("large_example.pgen"))
    ka = KA["large_example"]
    assert (pg.M, pg.N) == (3000, 8)
    counts = pg.counts_range()
    for c in counts:
        assert oracle.freq_from_counts(c) == (ka["alt_freq"], ka["obs_ct"])
    assert r6(oracle.hardy_from_counts(counts[0])[2]) == ka["p_hwe_round6"]
    miss = pg.missing_per_sample()
    assert set(miss.tolist()) == {ka["sample_missing_ct"]}
    assert 3000 - ka["sample_missing_ct"] == ka["sample_obs_ct"]


def test_rare_small_compressed_records(oracle):
    pg = oracle.Pgen(data_path("rare_small.pgen"))
    tot = pg.counts_range().sum(axis=0)
    ka = KA["rare_small_totals"]
    assert tot.tolist() == [ka["hom_ref"], ka["het"], ka["hom_alt"], ka["missing"]]


def test_hardy_known_answers(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    for v, row in enumerate(KA["hardy_pgen_example"]["rows"]):
        c = pg.counts(v)
        assert c[:3].tolist() == row["counts"]
        o_het, e_het, p = oracle.hardy_from_counts(c)
        assert (r6(o_het), r6(e_het), r6(p)) == (row["o_het"], row["e_het"], row["p"])
        assert r6(oracle.hardy_from_counts(c, midp=True)[2]) == row["p_midp"]
    inc = np.array([1, 0, 1, 0], dtype=np.uint8)
    assert r6(oracle.hardy_from_counts(pg.counts(0, inc))[2]) == KA["hardy_pgen_example"]["subset_0_2_rs1_p"]
    # AVG(P_HWE) = 0.857143 (plink_hardy.test:345-349)
    ps = [oracle.hardy_from_counts(pg.counts(v))[2] for v in range(4)]
    assert r6(float(np.mean(ps))) == 0.857143


def test_sex_chromosomes(oracle):
    pg = oracle.Pgen(data_path("sexchr_example.pgen"))
    pvar = oracle.load_pvar(data_path("sexchr_example.pvar"))
    psam = oracle.load_psam(data_path("sexchr_example.psam"))
    ka = KA["sexchr"]
    for v, vid in enumerate(pvar["id"]):
        ploidy = oracle.classify_ploidy(pvar["chrom"][v], pvar["pos"][v])
        if ploidy == "auto":
            c = pg.counts(v)
            af, obs = oracle.freq_from_counts(c)
            hw = oracle.hardy_from_counts(c)
            exp = ka["hardy"][vid]
            assert c[:3].tolist() == exp["counts"]
            assert (r6(hw[0]), r6(hw[1]), r6(hw[2])) == (exp["o_het"], exp["e_het"], exp["p"])
        else:
            sac = oracle.sex_aware_counts(pg.geno(v), ploidy, psam["sex"])
            af = sac["alt_allele_ct"] / sac["obs_allele_ct"]
            obs = sac["obs_allele_ct"]
            exp = ka["hardy"][vid]
            if ploidy == "x":
                assert [sac["hwe_hom_ref"], sac["hwe_het"], sac["hwe_hom_alt"]] == exp["counts"]
                male_ref = sac["geno_hom_ref"] - sac["hwe_hom_ref"]
                male_alt = sac["geno_hom_alt"] - sac["hwe_hom_alt"]
                for midp, key in ((False, "p"), (True, "p_midp")):
                    p = oracle.ln_p_to_pvalue(oracle.hwe_xchr_lnp(sac["hwe_het"], sac["hwe_hom_ref"],
                                                                 sac["hwe_hom_alt"], male_ref, male_alt, midp))
                    assert r6(p) == exp[key]
            else:
                assert [sac["geno_hom_ref"], sac["geno_het"], sac["geno_hom_alt"]] == exp["counts"]
            if vid == "y1":
                assert [sac["geno_hom_ref"], sac["geno_het"], sac["geno_hom_alt"], sac["geno_missing"]] == \
                    ka["freq_counts_y1"]
        assert [r6(af), obs] == ka["freq"][vid]
    # sexchr_xpar.pvar relabels par1 to X:1000000; build 'none' makes it non-PAR
    xp = oracle.load_pvar(data_path("sexchr_xpar.pvar"))
    v = xp["id"].index("par1")
    assert oracle.classify_ploidy(xp["chrom"][v], xp["pos"][v], "grch38") == "auto"
    assert oracle.classify_ploidy(xp["chrom"][v], xp["pos"][v], "none") == "x"
    sac = oracle.sex_aware_counts(pg.geno(v), "x", psam["sex"])
    assert [sac["alt_allele_ct"] / sac["obs_allele_ct"], sac["obs_allele_ct"]] == ka["xpar_build_none_par1"]


def test_missing_known_answers(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    ka = KA["missing_pgen_example"]
    assert [int(pg.counts(v)[3]) for v in range(4)] == ka["variant_missing_ct"]
    assert pg.missing_per_sample().tolist() == ka["sample_missing_ct"]
    assert pg.missing_per_sample(0, 2).tolist() == ka["region_1_10000_20000_sample_missing_ct"]
    inc = np.array([1, 0, 0, 1], dtype=np.uint8)
    assert pg.missing_per_sample(0, 2, inc).tolist() == [0, 1]


def test_score_known_answers(oracle):
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    ka = KA["score_pgen_example"]
    w = ka["weights"]
    s, d, ac = oracle.score(pg, range(4), w)
    assert s[:, 0].tolist() == ka["default"]["score_sum"]
    assert d.tolist() == ka["default"]["dosage_sum"]
    assert ac.tolist() == ka["default"]["allele_ct"]
    assert (s[:, 0] / ac).tolist() == ka["default"]["score_avg"]
    s, _, _ = oracle.score(pg, [0], [1.0], flip=[1])
    assert s[0, 0] == ka["flip_rs1_weight1"]["SAMPLE1"] and s[2, 0] == ka["flip_rs1_weight1"]["SAMPLE3"]
    s, d, ac = oracle.score(pg, range(4), w, mode="no_mean_imputation")
    for name, exp in ka["no_mean_imputation"].items():
        i = int(name[-1]) - 1
        assert s[i, 0] == exp["score_sum"]
        if "allele_ct" in exp:
            assert ac[i] == exp["allele_ct"] and d[i] == exp["dosage_sum"]
    s, d, ac = oracle.score(pg, [1], [1.0], mode="center")
    exp = ka["center_rs2_weight1"]
    assert np.allclose(s[:, 0], exp["score_sum"], rtol=0, atol=1e-15)
    assert set(ac.tolist()) == {exp["allele_ct"]} and set(d.tolist()) == {exp["dosage_sum"]}
    _, _, ac = oracle.score(pg, range(4), w, mode="center")
    assert ac[1] == ka["center_all_sample2_allele_ct"]
    # all-missing file: everything skipped
    am = oracle.Pgen(data_path("all_missing.pgen"))
    s, d, ac = oracle.score(am, range(2), [1.0, 0.5])
    assert not s.any() and not ac.any()
    # subset SAMPLE1+SAMPLE3: SAMPLE1 still -0.5 (plink_score.test:180-186)
    inc = np.array([1, 0, 1, 0], dtype=np.uint8)
    s, _, _ = oracle.score(pg, range(4), w, include=inc)
    assert s[0, 0] == -0.5 and s.shape[0] == 2


def test_dosage_known_answers(oracle):
    pg = oracle.Pgen(data_path("dosage_example.pgen"))
    ka = KA["dosage_example"]
    assert pg.has_dosage
    for v in range(4):
        d = pg.dosage(v)
        exp = [(-9.0 if x is None else x) for x in ka["dosages"][v]]
        assert d.tolist() == exp
        af, obs = oracle.freq_from_counts(pg.counts(v))
        assert af == ka["hardcall_freq"][v] and obs == ka["hardcall_obs"][v]
        counts, dos, r2 = pg.dcounts(v)
        daf, dobs = oracle.freq_from_dcounts(dos)
        assert daf == ka["dosage_freq"][v] and dobs == ka["dosage_obs"][v]
        assert r2 == pytest.approx(ka["imp_r2"][v], rel=1e-12)


def test_phase_known_answers(oracle):
    pg = oracle.Pgen(data_path("phased_example.pgen"))
    assert pg.has_phase
    for v in range(4):
        g, pp, pi = pg.phase(v)
        pairs = oracle.unphased_pairs(g, pp, pi)
        exp = KA["phased_example"]["pairs"][v]
        for s in range(4):
            if exp[s] is None:
                assert g[s] == -9
            else:
                assert pairs[s].tolist() == exp[s]
    # subset [0, 2] of rs1 -> [[0,0],[1,0]] (read_pgen_phased.test:84-87)
    g, pp, pi = pg.phase(0, np.array([1, 0, 1, 0], dtype=np.uint8))
    assert oracle.unphased_pairs(g, pp, pi).tolist() == [[0, 0], [1, 0]]


def test_pca_known_answers(oracle):
    pg = oracle.Pgen(data_path("pca_example.pgen"))
    ka = KA["pca_example"]
    ev, vecs, m_eff = oracle.pca(pg, ka["n_pcs"])
    assert [round(x, 10) for x in ev] == ka["eigenvalues_round10"]
    assert ev[0] == pytest.approx(ka["eigenvalue1_full"], rel=1e-9)
    assert vecs.shape == (250, 3)
    # orthonormal eigenvectors
    assert np.allclose(vecs.T @ vecs, np.eye(3), atol=1e-10)


def test_streaming_example_record_types(oracle):
    """50k-variant fixture mixing plain, LD and difflist records: totals are
    self-consistent and the threaded scan equals the scalar one."""
    pg = oracle.Pgen(data_path("streaming_example.pgen"))
    kinds = np.bincount([pg.vrtype(v) & 7 for v in range(pg.M)], minlength=8)
    assert kinds.tolist() == [49904, 0, 4, 7, 24, 0, 61, 0]
    a = pg.counts_range()
    assert (a.sum(axis=1) == 8).all()
    b = pg.scan_counts_mt(0, pg.M, 4)
    assert np.array_equal(a, b)


def test_hwe_extremes(oracle):
    # symmetric in the two homozygote counts, finite for large balanced counts
    assert oracle.hwe_lnp(100, 300, 50) == pytest.approx(oracle.hwe_lnp(100, 50, 300), rel=1e-12)
    lp = oracle.hwe_lnp(49000, 26000, 25000)
    assert -50 < lp <= 0.0
    assert oracle.hwe_lnp(0, 0, 0) == 0.0
    assert math.isinf(oracle.hwe_lnp(100000, 200000, 200000))


def test_ld_known_answers(oracle):
    """plink_ld.test:19-62, 121-137: r2 / D' / OBS_CT of the pgen_example pairs."""
    ka = KA["plink_ld"]
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    idx = {"rs1": 0, "rs2": 1, "rs3": 2, "rs4": 3}
    for a, b, r2, dp, n in ka["pairwise"]:
        got = oracle.ld_stats(pg.ld_sums(idx[a], idx[b]))
        assert got[0] == pytest.approx(r2, rel=1e-12) and got[2] == n
        if dp is not None:
            assert got[1] == pytest.approx(dp, rel=1e-12)
    for a, r2, n in ka["self"]:
        got = oracle.ld_stats(pg.ld_sums(idx[a], idx[a]))
        assert got[0] == pytest.approx(r2, rel=1e-12) and got[2] == n
    inc = np.array([1, 1, 0, 0], dtype=np.uint8)
    assert list(oracle.ld_stats(pg.ld_sums(0, 1, include=inc))) == ka["subset_s1_s2"]
    am = oracle.Pgen(data_path("all_missing.pgen"))
    assert list(oracle.ld_stats(am.ld_sums(0, 1))) == ka["all_missing"]
    big = oracle.Pgen(data_path("large_example.pgen"))
    r2s = {round(oracle.ld_stats(big.ld_sums(a, b))[0], 12) for a in range(10) for b in range(a + 1, 10)}
    assert r2s == {1.0}


def test_sample_counts_known_answers(oracle):
    """read_pfile_genotypes_counts.test:50-63, read_pfile_sample_counts_sparse.test:10-14."""
    ka = KA["read_pfile"]
    pg = oracle.Pgen(data_path("pgen_example.pgen"))
    got = pg.sample_counts()
    assert [list(map(int, r)) for r in got] == [ka["sample_counts"][f"SAMPLE{i}"] for i in range(1, 5)]
    rare = oracle.Pgen(data_path("rare_small.pgen"))
    tot = rare.sample_counts().sum(axis=0, dtype=np.int64)
    t = ka["rare_small_totals"]
    assert [int(x) for x in tot] == [t["hom_ref"], t["het"], t["hom_alt"], t["missing"]]
