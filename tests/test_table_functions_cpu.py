"""The table-function shells (C++, csrc/shell) exercised WITHOUT a GPU: what the
reference's *_negative.test files assert happens at bind time, the registration
contract, projection pushdown of metadata-only queries, and MaxThreads().
Message substrings are the ones the reference's tests require."""

import os

import pytest

from conftest import data_path

F = pytest.importorskip("plinking_duck_amd.functions")

EX = data_path("pgen_example.pgen")
ORPHAN = data_path("pgen_orphan.pgen")
W = [1.0, 0.5, -0.5, 2.0]


def test_registered_functions():
    assert sorted(F.functions()) == ["plink_freq", "plink_hardy", "plink_ld", "plink_missing", "plink_pca",
                                     "plink_score", "read_pfile", "read_pgen"]


def err(fn, *args, exc=F.InvalidInputException, **kw):
    with pytest.raises(exc) as e:
        F.query(fn, *args, **kw)
    return str(e.value)


@pytest.mark.parametrize("fn", ["plink_freq", "plink_hardy", "plink_missing", "read_pgen"])
def test_common_bind_errors(fn):
    # file not found: the function name is in the message (plink_freq_negative.test etc.)
    assert fn in err(fn, "nonexistent.pgen")
    assert "cannot find .pvar or .bim" in err(fn, data_path("pgen_no_pvar.pgen"))
    assert "variant count mismatch" in err(fn, EX, pvar=data_path("mismatched_variants.pvar"))
    assert "sample count mismatch" in err(fn, EX, psam=data_path("mismatched_samples.psam"))
    assert "not found" in err(fn, EX, samples=["NONEXISTENT"])
    assert "out of range" in err(fn, EX, samples=[999])
    assert "duplicate sample index" in err(fn, EX, samples=[0, 1, 0])
    assert "samples list must not be empty" in err(fn, EX, samples=[])
    assert "LIST(INTEGER)" in err(fn, ORPHAN, samples=["SAMPLE1"])
    with pytest.raises(F.BinderException):
        F.query(fn, EX, no_such_parameter=1)
    with pytest.raises(F.BinderException):
        F.query(fn)


@pytest.mark.parametrize("fn", ["plink_freq", "plink_hardy", "plink_missing"])
def test_region_errors(fn):
    assert "region" in err(fn, EX, region="invalid")
    assert "region" in err(fn, EX, region="1:abc-100")
    assert "region" in err(fn, EX, region=":1-100")


def test_build_parameter():
    assert "unrecognized build" in err("plink_freq", EX, build="hg17")
    assert "unrecognized build" in err("plink_hardy", EX, build="hg17")


def test_missing_mode_errors():
    assert "mode must be 'variant' or 'sample'" in err("plink_missing", EX, mode="invalid")
    assert "sample mode requires" in err("plink_missing", ORPHAN, mode="sample")


def test_score_bind_errors():
    assert "plink_score" in err("plink_score", "nonexistent.pgen", weights=[1.0, 0.5])
    assert "weights parameter is required" in err("plink_score", EX)
    assert "weights list length" in err("plink_score", EX, weights=[1.0, 0.5])
    assert "weights list is empty" in err("plink_score", EX, weights=[])
    assert "weights must not be NULL" in err("plink_score", EX, weights=None)
    assert "ID-keyed weights must be" in err("plink_score", EX, weights=[{"variant": "rs1", "a1": "G", "beta": 1.0}])
    assert "cannot find .psam or .fam" in err("plink_score", ORPHAN, weights=W)
    assert "center and no_mean_imputation cannot both be true" in err("plink_score", EX, weights=W, center=True,
                                                                     no_mean_imputation=True)
    assert "variant count mismatch" in err("plink_score", EX, weights=W, pvar=data_path("mismatched_variants.pvar"))
    assert "region" in err("plink_score", EX, weights=W, region="invalid")


def test_pca_bind_errors():
    pca = data_path("pca_example.pgen")
    assert "n_pcs must be >= 1" in err("plink_pca", pca, n_pcs=0)
    assert "n_pcs must be >= 1" in err("plink_pca", pca, n_pcs=-1)
    assert "invalid mode" in err("plink_pca", pca, mode="invalid")
    assert "cannot find .pvar" in err("plink_pca", data_path("nonexistent.pgen"))
    assert "cannot find .psam" in err("plink_pca", ORPHAN)


def test_read_pgen_bind_errors():
    assert "min (0.8) > max (0.2)" in err("read_pgen", EX, af_range={"min": 0.8, "max": 0.2})
    assert "unknown field 'minimum'" in err("read_pgen", EX, af_range={"minimum": 0.1})
    assert "out of range" in err("read_pgen", EX, af_range={"min": -0.5})
    assert "must be a STRUCT" in err("read_pgen", EX, af_range=0.5)
    assert "unknown category" in err("read_pgen", EX, include_genotypes=["homozygous"])
    assert "specify only one of" in err("read_pgen", EX, include_genotypes=["het"], genotype_range={"min": 1})
    assert "dosages and phased cannot both be true" in err("read_pgen", EX, dosages=True, phased=True)
    assert "incompatible with dosages" in err("read_pgen", EX, genotypes="counts", dosages=True)
    assert "incompatible with phased" in err("read_pgen", EX, genotypes="stats", phased=True)
    assert "invalid genotypes value" in err("read_pgen", EX, genotypes="matrix")
    assert "orient" in err("read_pgen", EX, orient="sample")
    # read_pgen_genotypes_columns_negative.test:12-14
    assert "genotypes := 'columns' requires a .psam/.fam file" in err(
        "read_pgen", data_path("pgen_orphan.pgen"), genotypes="columns")
    assert "genotypes := 'struct' requires a .psam/.fam file" in err(
        "read_pgen", data_path("pgen_orphan.pgen"), genotypes="struct")


def test_read_pgen_variants_parameter_binds():
    """read_pgen_variants.test: every accepted shape of `variants :=`, metadata-only so no device is needed."""
    ids = lambda **kw: F.query("read_pgen", EX, columns=["ID"], **kw).column("ID")
    assert ids(variants=0) == ["rs1"] and ids(variants=3) == ["rs4"]
    assert ids(variants="rs1") == ["rs1"]
    assert ids(variants=[0, 2]) == ["rs1", "rs3"]
    assert ids(variants=[2, 0]) == ["rs3", "rs1"]          # caller order is kept
    assert ids(variants=["rs1", "rs4"]) == ["rs1", "rs4"]
    assert ids(variants="1:10000") == ["rs1"] and ids(variants="1:10000:A:G") == ["rs1"]
    assert ids(variants={"start": 0, "stop": 1}) == ["rs1", "rs2"]
    assert ids(variants={"start": "rs2", "stop": "rs4"}) == ["rs2", "rs3", "rs4"]
    assert ids(variants={"chrom": "2", "pos": 15000}) == ["rs4"]
    assert ids(variants=[{"chrom": "1", "pos": 20000}, {"chrom": "2", "pos": 15000}]) == ["rs2", "rs4"]
    assert "out of range" in err("read_pgen", EX, variants=999)
    assert "not found" in err("read_pgen", EX, variants="rs999")
    assert "not found" in err("read_pgen", EX, variants="1:10000:A:T")
    assert "not found" in err("read_pgen", EX, variants={"chrom": "7", "pos": 1})
    assert "invalid CPRA format" in err("read_pgen", EX, variants="1:2:3")
    assert "duplicate variant index 0" in err("read_pgen", EX, variants=[0, 1, 0])
    assert "must not be empty" in err("read_pgen", EX, variants=[])
    assert "must not be NULL" in err("read_pgen", EX, variants=None)
    assert "ambiguous" in err("read_pgen", EX, variants={"start": 0, "chrom": "1"})
    assert "after stop" in err("read_pgen", EX, variants={"start": 2, "stop": 1})
    assert "must be an integer, string, struct, or list" in err("read_pgen", EX, variants=1.5)


def test_metadata_only_projection_needs_no_device():
    """Projection pushdown: with no genotype-derived column projected the shells
    never open the device (reference: need_frequencies, src/plink_freq.cpp:309-323)."""
    r = F.query("plink_freq", EX, columns=["CHROM", "POS", "ID", "REF", "ALT"])
    assert r.sorted("CHROM", "POS") == [("1", 10000, "rs1", "A", "G"), ("1", 20000, "rs2", "C", "T"),
                                        ("1", 30000, "rs3", "G", "A"), ("2", 15000, "rs4", "T", "C")]
    assert r.types == ["VARCHAR", "INTEGER", "VARCHAR", "VARCHAR", "VARCHAR"]
    r = F.query("plink_hardy", EX, columns=["ID", "A1"])
    assert sorted(r.rows) == [("rs1", "G"), ("rs2", "T"), ("rs3", "A"), ("rs4", "C")]
    r = F.query("plink_missing", EX, columns=["ID"], region="1:10000-20000")
    assert sorted(r.column("ID")) == ["rs1", "rs2"]
    assert len(F.query("plink_freq", EX, columns=["ID"], region="99:1-100")) == 0
    assert len(F.query("plink_freq", EX, columns=["ID"], region="1:1-9999")) == 0
    r = F.query("plink_missing", EX, mode="sample", columns=["FID", "IID"])
    assert sorted(r.rows, key=lambda t: t[1]) == [(None, f"SAMPLE{i}") for i in range(1, 5)]
    r = F.query("read_pgen", EX, columns=["ID", "POS"])
    assert len(r) == 4
    # .bim companion, ALT '.' / empty ID -> NULL handling lives in the same column fill
    r = F.query("plink_freq", EX, pvar=data_path("pgen_example.bim"), columns=["ID", "REF", "ALT"])
    assert sorted(r.rows)[0] == ("rs1", "A", "G")


def test_output_schemas():
    r = F.query("plink_freq", EX, columns=["ID"], counts=True, dosage=True)
    assert r.all_names == ["CHROM", "POS", "ID", "REF", "ALT", "ALT_FREQ", "OBS_CT", "HOM_REF_CT", "HET_CT",
                           "HOM_ALT_CT", "MISSING_CT", "IMP_R2"]
    assert F.query("plink_freq", EX, columns=["ID"], dosage=True).all_names[-1] == "IMP_R2"
    assert F.query("plink_hardy", EX, columns=["ID"]).all_names == [
        "CHROM", "POS", "ID", "REF", "ALT", "A1", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "O_HET", "E_HET", "P_HWE"]
    assert F.query("plink_missing", EX, columns=["ID"]).all_names[-3:] == ["MISSING_CT", "OBS_CT", "F_MISS"]
    assert F.query("plink_missing", EX, mode="sample", columns=["IID"]).all_names == [
        "FID", "IID", "MISSING_CT", "OBS_CT", "F_MISS"]
    assert F.query("plink_score", EX, weights=W, columns=["IID"]).all_names == [
        "FID", "IID", "ALLELE_CT", "DENOM", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM", "SCORE_AVG"]


def test_max_threads_follow_the_reference_formulas():
    """MaxThreads(): freq/hardy min(range/500 + 1, cap), cap = plinking_max_threads or 16
    (src/plink_freq.cpp:84-87, plink_common.cpp:1906-1911); metadata-only scan here."""
    big = data_path("streaming_example.pgen")  # 50,000 variants
    assert F.query("plink_freq", big, columns=["ID"], threads=64).threads == 16
    assert F.query("plink_freq", big, columns=["ID"], threads=64, settings={"plinking_max_threads": 2}).threads == 2
    assert F.query("plink_freq", big, columns=["ID"], threads=3).threads == 3
    assert F.query("plink_freq", EX, columns=["ID"], threads=8).threads == 1
    r = F.query("plink_freq", big, columns=["ID"], threads=8)
    ids = r.column("ID")
    assert len(ids) == 50000 and len(set(ids)) == 50000  # no duplicate rows from thread races
    assert F.query("read_pgen", big, columns=["ID"], threads=64).threads == 16
    assert F.query("plink_missing", big, columns=["ID"], threads=5).threads == 5


def test_plink_ld_bind_errors():
    """plink_ld_negative.test"""
    assert "plink_ld" in err("plink_ld", "nonexistent.pgen", variant1="rs1", variant2="rs2")
    assert "not found in .pvar" in err("plink_ld", EX, variant1="NOSUCHVARIANT", variant2="rs2")
    assert "not found in .pvar" in err("plink_ld", EX, variant1="rs1", variant2="NOSUCHVARIANT")
    assert "both variant1 and variant2" in err("plink_ld", EX, variant1="rs1")
    assert "both variant1 and variant2" in err("plink_ld", EX, variant2="rs2")
    assert "r2_threshold" in err("plink_ld", EX, r2_threshold=-0.1)
    assert "r2_threshold" in err("plink_ld", EX, r2_threshold=1.5)
    assert "window_kb" in err("plink_ld", EX, window_kb=-1)
    assert "not found" in err("plink_ld", EX, variant1="rs1", variant2="rs2", samples=["NOSUCHSAMPLE"])
    assert "region" in err("plink_ld", EX, region="invalid_region", variant1="rs1", variant2="rs2")
    # a region with fewer than two variants has no pairs and never touches the device
    assert len(F.query("plink_ld", EX, region="2:15000-15000", r2_threshold=0.0)) == 0
    assert len(F.query("plink_ld", EX, region="99:1-100", r2_threshold=0.0)) == 0
    r = F.query("plink_ld", EX, region="99:1-100")
    assert r.all_names == ["CHROM_A", "POS_A", "ID_A", "CHROM_B", "POS_B", "ID_B", "R2", "D_PRIME", "OBS_CT"]


def test_read_pfile_file_lists_bind():
    """read_pfile_list.test, read_pfile_list_shards.test sections 1, 5, 7 (variants), 8: metadata only, no device."""
    PFX = data_path("pgen_example")
    assert len(F.query("read_pfile", [PFX, PFX], columns=["ID"])) == 8
    assert len(set(F.query("read_pfile", [PFX, PFX], columns=["POS"]).column("POS"))) == 4
    assert len(F.query("read_pfile", [PFX], columns=["ID"])) == 4
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    whole = data_path("large_example")
    for threads in (1, 4):
        got = F.query("read_pfile", shards, columns=["CHROM", "POS", "ID"], threads=threads)
        assert sorted(got.rows) == sorted(F.query("read_pfile", whole, columns=["CHROM", "POS", "ID"]).rows)
    got = F.query("read_pfile", shards, region="1:5000-50000", columns=["CHROM", "ID"])
    assert len(got) == 451 and set(got.column("CHROM")) == {"1"}
    assert sorted(got.column("ID")) == sorted(F.query("read_pfile", whole, region="1:5000-50000", columns=["ID"]).column("ID"))
    r = F.query("read_pfile", [whole, data_path("streaming_example")], columns=["CHROM"])
    assert len(r) == 53000 and len(set(r.column("CHROM"))) == 3
    assert len(F.query("read_pfile", shards[0], variants=[0, 1], columns=["ID"])) == 2
    assert len(F.query("read_pfile", shards[:1], columns=["ID"])) == 1000
    # a psam override applies to every shard
    assert len(F.query("read_pfile", shards, psam=shards[0] + ".psam", columns=["ID"])) == 3000
    assert "sample count mismatch across files" in err("read_pfile", [PFX, whole])
    assert "sample count mismatch across files" in err("read_pfile", [shards[0], PFX], orient="sample", genotypes="counts")
    assert "overrides cannot be combined with a multi-file list" in err("read_pfile", [PFX, PFX], pvar=PFX + ".pvar")
    assert "overrides cannot be combined with a multi-file list" in err("read_pfile", [PFX, PFX], pgen=PFX + ".pgen")
    assert "variants := [...] with a multi-file list is not yet supported" in err("read_pfile", shards[:2], variants=["var1"])
    assert "not yet supported" in err("read_pfile", shards[:2], variants=[0, 1])
    assert "empty file list provided" in err("read_pfile", [])
    assert "cannot find .pgen file for prefix '%s'" % data_path("does_not_exist") in err(
        "read_pfile", [shards[0], data_path("does_not_exist")])


def test_read_pfile_sample_orient_schemas():
    """read_pfile_orient.test:55-90, :180-215, read_pfile_genotypes_columns.test:81-90, :142-159: the per-element
    modes of orient := 'sample' bind from the metadata alone, and a metadata-only projection needs no device."""
    P = data_path("pfile_example")
    r = F.query("read_pfile", P, orient="sample", columns=["FID", "IID", "SEX"])
    assert r.all_names == ["FID", "IID", "SEX", "genotypes"] and r.all_types[-1] == "TINYINT[4]"
    assert sorted(r.rows) == [("FAM001", "SAMPLE1", 1), ("FAM001", "SAMPLE2", 2), ("FAM002", "SAMPLE3", None),
                              ("FAM002", "SAMPLE4", 1)]
    assert F.query("read_pfile", P, orient="sample", genotypes="list", columns=["IID"]).all_types[-1] == "TINYINT[]"
    assert F.query("read_pfile", P, orient="sample", dosages=True, columns=["IID"]).all_types[-1] == "DOUBLE[4]"
    assert F.query("read_pfile", P, orient="sample", variants=["rs1", "rs2"], columns=["IID"]).all_types[-1] == "TINYINT[2]"
    assert F.query("read_pfile", P, orient="sample", region="1:10000-20000", columns=["IID"]).all_types[-1] == "TINYINT[2]"
    assert F.query("read_pfile", P, orient="sample", region="1:10000-30000", variants=["rs1", "rs3"],
                   samples=["SAMPLE1"], columns=["IID"]).rows == [("SAMPLE1",)]
    c = F.query("read_pfile", P, orient="sample", genotypes="columns", columns=["FID", "IID"])
    assert c.all_names == ["FID", "IID", "SEX", "rs1", "rs2", "rs3", "rs4"] and c.all_types[3:] == ["TINYINT"] * 4
    assert sorted(c.rows) == [("FAM001", "SAMPLE1"), ("FAM001", "SAMPLE2"), ("FAM002", "SAMPLE3"), ("FAM002", "SAMPLE4")]
    st = F.query("read_pfile", P, orient="sample", genotypes="struct", region="1:10000-20000", columns=["IID"])
    assert st.all_types[-1] == "STRUCT(rs1 TINYINT, rs2 TINYINT)"
    msg = err("read_pfile", P, orient="sample", settings={"plinking_max_matrix_elements": 15})
    assert "would require 16 genotype values (4 variants x 4 samples, limit: 15)" in msg
    assert len(F.query("read_pfile", P, orient="sample", settings={"plinking_max_matrix_elements": 16}, columns=["IID"])) == 4
    assert "include_genotypes is incompatible with dosages" in err("read_pfile", P, orient="sample", dosages=True,
                                                                   include_genotypes=["het"])
    # orient := 'genotype' (read_pfile_genotype_orient.test): variant columns, psam columns, the scalar genotype
    g = F.query("read_pfile", P, orient="genotype", columns=["ID", "IID", "SEX"])
    assert g.all_names == ["CHROM", "POS", "ID", "REF", "ALT", "FID", "IID", "SEX", "genotype"]
    assert g.all_types[-1] == "TINYINT" and len(g) == 16
    assert sorted(r for r in g.rows if r[0] == "rs1") == [("rs1", "SAMPLE1", 1), ("rs1", "SAMPLE2", 2),
                                                          ("rs1", "SAMPLE3", None), ("rs1", "SAMPLE4", 1)]
    assert len(F.query("read_pfile", P, orient="genotype", samples=[2, 0], columns=["IID"])) == 8
    assert F.query("read_pfile", P, orient="genotype", dosages=True, columns=["IID"]).all_types[-1] == "DOUBLE"
    for mode in ("columns", "struct"):
        assert "genotype mode already produces scalar output" in err("read_pfile", P, orient="genotype", genotypes=mode)


def test_read_pfile_sample_multifile_bind():
    """read_pfile_sample_multifile.test: dimensions, combine_samples modes, psam override, negatives (bind only)."""
    shards = [data_path("shard%d" % i) for i in (1, 2, 3)]
    q = lambda files, **kw: F.query("read_pfile", files, orient="sample", columns=["IID"], **kw)
    assert q(shards).all_types[-1] == "TINYINT[3000]" and len(q(shards)) == 8
    assert q(shards[:2], combine_samples="identical").all_types[-1] == "TINYINT[2000]"
    assert len(q(shards[:2], combine_samples="implicit")) == 8
    r = q(shards, samples=["SAMP1", "SAMP2"])
    assert len(r) == 2 and r.all_types[-1] == "TINYINT[3000]"
    c = q(shards, genotypes="columns")
    assert {"var1", "var2", "var3", "var3000"} <= set(c.all_names) and len(c.all_names) == 2 + 3000
    r = q(shards[:2], psam=shards[0] + ".psam")
    assert len(r) == 8 and r.all_types[-1] == "TINYINT[2000]"
    assert len(q(shards[:2], psam=shards[0] + ".psam", combine_samples="identical")) == 8
    for mode in ("union", "intersect", "concatenate"):
        assert "not yet implemented" in err("read_pfile", shards[:2], orient="sample", combine_samples=mode)
    assert "unknown combine_samples" in err("read_pfile", shards[:1], orient="sample", combine_samples="nonsense")
    # read_pfile_orient_negative.test: read_pgen has the variant orientation only
    for orient in ("sample", "genotype"):
        assert f"orient := '{orient}' is not supported" in err("read_pgen", data_path("pgen_example.pgen"), orient=orient)
    assert "invalid orient value 'invalid'" in err("read_pfile", data_path("pfile_example"), orient="invalid")
    assert "sample count mismatch" in err("read_pfile", shards[:2], psam=data_path("pgen_example.psam"))
    for kw in ({"pgen": shards[0] + ".pgen"}, {"pvar": shards[0] + ".pvar"}):
        assert "pgen/pvar overrides cannot be combined with a multi-file list" in err("read_pfile", shards[:2], **kw)
    # 'identical' compares the shards' IIDs; 'implicit' trusts the positions (large_example: SAMP*, streaming: SAMPLE*)
    two = [data_path("large_example"), data_path("streaming_example")]
    for orient in ("variant", "sample"):
        msg = err("read_pfile", two, orient=orient, combine_samples="identical")
        assert "combine_samples := 'identical' but sample 0 differs: 'SAMPLE1'" in msg and "vs 'SAMP1'" in msg
    assert len(F.query("read_pfile", two, combine_samples="implicit", columns=["ID"])) == 53000
    assert len(F.query("read_pfile", two, combine_samples="identical", psam=two[0] + ".psam", columns=["ID"])) == 53000


def test_read_pfile_bind():
    PFX = data_path("pgen_example")
    r = F.query("read_pfile", PFX, columns=["ID", "POS"])
    assert sorted(r.rows) == [("rs1", 10000), ("rs2", 20000), ("rs3", 30000), ("rs4", 15000)]
    assert r.all_names == ["CHROM", "POS", "ID", "REF", "ALT", "genotypes"]
    # explicit paths instead of a prefix; a full .pgen path as the prefix
    assert len(F.query("read_pfile", EX, columns=["ID"])) == 4
    assert len(F.query("read_pfile", None, pgen=EX, columns=["ID"])) == 4
    assert F.query("read_pfile", PFX, region="1:10000-20000", columns=["ID"]).column("ID") == ["rs1", "rs2"]
    assert F.query("read_pfile", PFX, region="1:10000-20000", variants=["rs2", "rs4"], columns=["ID"]).column("ID") == ["rs2"]
    s = F.query("read_pfile", PFX, orient="sample", genotypes="counts", columns=["IID", "SEX"])
    assert s.all_names == ["IID", "SEX", "genotypes"]
    assert sorted(s.rows) == [(f"SAMPLE{i}", None) for i in range(1, 5)]
    assert F.query("read_pfile", PFX, orient="sample", genotypes="stats", columns=["IID"]).all_names[-1] == "genotypes"
    assert "cannot find .pgen file for prefix" in err("read_pfile", data_path("no_such_prefix"))
    assert "no .pgen file path provided" in err("read_pfile", "")
    assert "empty file list provided" in err("read_pfile", None)  # null_list_params.test:30-33
    assert "invalid orient value" in err("read_pfile", PFX, orient="diagonal")
    assert "not compatible with orient := 'genotype'" in err("read_pfile", PFX, orient="genotype", genotypes="counts")
    assert "incompatible with phased" in err("read_pfile", PFX, orient="sample", genotypes="counts", phased=True)
    assert "incompatible with dosages" in err("read_pfile", PFX, orient="sample", genotypes="stats", dosages=True)
    assert "dosages and phased cannot both be true" in err("read_pfile", PFX, dosages=True, phased=True)
    assert "read_pfile: invalid genotypes value" in err("read_pfile", PFX, genotypes="matrix")
    assert "Invalid named parameter" in err("read_pgen", EX, region="1:1-2", exc=F.BinderException)
    # read_pfile_negative.test:113-131 and the open forms of its region grammar
    assert "invalid region" in err("read_pfile", PFX, region="invalid:abc-def")
    assert "empty chromosome" in err("read_pfile", PFX, region=":100-200")
    assert "region start (30000) > end (10000)" in err("read_pfile", PFX, region="1:30000-10000")
    assert F.query("read_pfile", PFX, region="2", columns=["ID"]).column("ID") == ["rs4"]
    assert F.query("read_pfile", PFX, region="1:20000-", columns=["ID"]).column("ID") == ["rs2", "rs3"]
    assert len(F.query("read_pfile", PFX, region="99:1-100", columns=["ID"])) == 0


def test_null_list_params_test_mirror():
    """null_list_params.test: NULL / typed-NULL list parameters."""
    PFX = data_path("pgen_example")
    assert "samples list must not be empty" in err("read_pfile", PFX, samples=None)
    assert "samples list must not be empty" in err("read_pfile", PFX, samples=[])
    assert "samples list must not be empty" in err("plink_freq", EX, samples=None)
    assert "variants must not be NULL" in err("read_pfile", PFX, variants=None)
    assert len(F.query("read_pfile", PFX, include_genotypes=None, columns=["ID"])) == 4
    assert "weights must not be NULL" in err("plink_score", EX, weights=None)


def test_pvar_and_bim_text_forms(tmp_path):
    """The single-pass .pvar/.bim parser (csrc/shell/plink_common.cpp, replacing src/plink_common.cpp:171-375):
    CRLF line ends, a last line without a newline, extra and reordered columns, '.' for ID and ALT, blank runs in a
    .bim, empty lines, a POS with leading zeros longer than the small buffer -- metadata only, no device."""
    pv = tmp_path / "a.pvar"
    pv.write_bytes(b"##fileformat=x\r\n##more\r\n#CHROM\tID\tPOS\tALT\tREF\tQUAL\tINFO\r\n"
                   b"1\trs1\t10000\tG\tA\t.\tx=1\r\n"
                   b"\r\n"
                   b"1\t.\t" + b"0" * 40 + b"20000\t.\tC\t9\ty\r\n"
                   b"1\trs3\t30000\tA,T\tG\t.\t.\r\n"
                   b"2\trs4\t15000\tC\tT\t.\t.")
    r = F.query("plink_freq", EX, pvar=str(pv), columns=["CHROM", "POS", "ID", "REF", "ALT"])
    assert r.sorted("CHROM", "POS") == [("1", 10000, "rs1", "A", "G"), ("1", 20000, None, "C", None),
                                        ("1", 30000, "rs3", "G", "A,T"), ("2", 15000, "rs4", "T", "C")]
    bim = tmp_path / "a.bim"
    bim.write_text("1  rs1\t0   10000 G A\n1\trs2 0 20000\tT  C\n1 rs3 0 30000 A G\n2 . 0 15000 . T\n")
    r = F.query("plink_freq", EX, pvar=str(bim), columns=["CHROM", "POS", "ID", "REF", "ALT"])
    assert r.sorted("CHROM", "POS") == [("1", 10000, "rs1", "A", "G"), ("1", 20000, "rs2", "C", "T"),
                                        ("1", 30000, "rs3", "G", "A"), ("2", 15000, None, "T", None)]
    assert sorted(F.query("plink_freq", EX, pvar=str(bim), columns=["ID"], region="1:15000-30000").column("ID")) == [
        "rs2", "rs3"]


def test_pvar_parse_errors(tmp_path):
    """The reference's messages for malformed variant metadata, with the line numbers of the FILE (comment, header
    and empty lines counted)."""
    def pvar(text):
        p = tmp_path / "bad.pvar"
        p.write_text(text)
        return str(p)

    head = "##c\n#CHROM\tPOS\tID\tREF\tALT\n"
    assert "is empty" in err("plink_freq", EX, pvar=pvar(""))
    assert "contains no header or data" in err("plink_freq", EX, pvar=pvar("##only\n\n##comments\n"))
    assert "missing required columns" in err("plink_freq", EX, pvar=pvar("#CHROM\tPOS\tID\tREF\n1\t1\ta\tA\n"))
    msg = err("plink_freq", EX, pvar=pvar(head + "1\t10\ta\tA\tG\n\n1\t12x\tb\tA\tG\n"))
    assert "invalid POS value '12x' at line 5" in msg
    assert "invalid POS value '' at line 3" in err("plink_freq", EX, pvar=pvar(head + "1\t\ta\tA\tG\n"))
    assert "invalid POS value" in err("plink_freq", EX, pvar=pvar(head + "1\t99999999999999999999\ta\tA\tG\n"))
    assert "missing required fields (line 4)" in err("plink_freq", EX, pvar=pvar(head + "1\t10\ta\tA\tG\n1\t20\tb\tA\n"))
    assert "missing required fields (line 1)" in err("plink_freq", EX, pvar=pvar("1 rs1 0 10000 G\n"))
    assert "non-contiguous" in err("plink_freq", EX, pvar=pvar(head + "1\t1\ta\tA\tG\n2\t1\tb\tA\tG\n1\t2\tc\tA\tG\n2\t2\td\tA\tG\n"))


def test_pvar_side_cache_serves_a_fresh_process(tmp_path):
    """The binary side-cache of the parsed .pvar columns (csrc/shell/plink_common.cpp; the cost it removes from a
    new process's first bind is the reference's LoadVariantMetadata, src/plink_common.cpp:171-375): written by the
    first bind, read by the next PROCESS, ignored when the text changed, survived when it is garbage."""
    import glob
    import struct
    import subprocess
    import sys

    m, n = 30_000, 8
    prefix = tmp_path / "big"
    # a fixed-width (mode 0x02) .pgen: 12-byte header, one 2-byte record per variant
    with open(str(prefix) + ".pgen", "wb") as f:
        f.write(bytes([0x6c, 0x1b, 0x02]) + struct.pack("<II", m, n) + b"\x40" + b"\x00\x00" * m)
    with open(str(prefix) + ".psam", "w") as f:
        f.write("#IID\n" + "".join(f"S{i}\n" for i in range(n)))

    def write_pvar(tag):
        with open(str(prefix) + ".pvar", "w") as f:
            f.write("##x\n#CHROM\tPOS\tID\tREF\tALT\n")
            f.write("".join(f"{1 + i * 22 // m}\t{100 + i}\t{tag}{i}\t{'ACGT'[i % 4]}\t{'.' if i % 97 == 0 else 'T'}\n"
                            for i in range(m)))

    cache = tmp_path / "cache"
    cache.mkdir()
    code = ("import sys, json; sys.path.insert(0, %r); import plinking_duck_amd.functions as F; "
            "r = F.query('plink_freq', %r, columns=['CHROM', 'POS', 'ID', 'REF', 'ALT'], region='3:1-100000000'); "
            "print(json.dumps({'n': len(r), 'first': r.sorted('POS')[0], 'last': r.sorted('POS')[-1], "
            "'nulls': sum(1 for x in r.rows if x[4] is None), 'bind': r.timing_ms['bind']}))"
            % (str(__import__("conftest").ROOT), str(prefix) + ".pgen"))

    def run(extra_env=None):
        env = dict(os.environ, PLINKING_PVAR_CACHE="1", PLINKING_PVAR_CACHE_DIR=str(cache))
        env.update(extra_env or {})
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        return __import__("json").loads(out.stdout.strip().splitlines()[-1])

    write_pvar("rs")
    assert os.path.getsize(str(prefix) + ".pvar") > 256 << 10
    first = run()
    files = glob.glob(str(cache / "*.pvarc"))
    assert len(files) == 1 and first["n"] > 1000 and first["first"][2].startswith("rs")
    second = run()                                   # a new process: served from the side-cache
    assert {k: second[k] for k in ("n", "first", "last", "nulls")} == {k: first[k] for k in ("n", "first", "last", "nulls")}
    off = run({"PLINKING_PVAR_CACHE": "0"})          # and the text parse says the same
    assert {k: off[k] for k in ("n", "first", "last", "nulls")} == {k: first[k] for k in ("n", "first", "last", "nulls")}
    stamp = os.path.getmtime(files[0])
    # garbage in the cache file: parsed again, file rewritten
    with open(files[0], "r+b") as f:
        f.seek(40)
        f.write(b"\xff" * 64)
    assert run()["first"] == first["first"]
    # a new version of the text: never the old columns
    import time
    time.sleep(0.02)
    write_pvar("qv")
    renamed = run()
    assert renamed["first"][2].startswith("qv") and renamed["n"] == first["n"]
    assert glob.glob(str(cache / "*.pvarc")) == files and os.path.getmtime(files[0]) >= stamp
    assert not glob.glob(str(cache / "*.tmp"))


def negative_cases():
    import json

    with open(os.path.join(os.path.dirname(__file__), "golden", "negative_cases.json")) as f:
        return json.load(f)["cases"]


def _localise(v):
    """test/data/x -> this repo's copy of the fixture (paths that do not exist stay nonexistent)."""
    if isinstance(v, str) and v.startswith("test/data/"):
        return data_path(v[len("test/data/"):])
    if isinstance(v, list):
        return [_localise(x) for x in v]
    return v


def run_negative_case(case, have_device):
    args = [_localise(a) for a in case["args"]]
    named = {k: (_localise(v) if k in ("pvar", "psam", "pgen", "sex_file") else v) for k, v in case["named"].items()}
    with pytest.raises((F.InvalidInputException, F.BinderException, F.IOException)) as e:
        F.query(case["function"], *args, columns=case.get("columns"), settings=case.get("settings"), **named)
    if not have_device and "no ROCm-capable device" in str(e.value):
        pytest.skip("this check sits behind the open of the file on the device; the -m gpu twin runs it")
    assert case["error_contains"] in str(e.value)


@pytest.mark.parametrize("case", negative_cases(), ids=lambda c: c["source"].split("/")[-1])
def test_reference_negative_cases(case):
    """Every `statement error` of the reference's *_negative.test files for the eight table functions
    (tests/golden/negative_cases.json, extracted by tests/golden/make_negative_cases.py): the call must fail, with the
    substring the reference's test requires in the message.  Most fail before any device call; the few that sit behind
    the open of the file are skipped here and run by test_table_functions_gpu.py's twin of this test."""
    run_negative_case(case, have_device=False)
