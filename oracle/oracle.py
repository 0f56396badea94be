"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes wrapper over oracle/libpgen_oracle.so (pgen_oracle.c) plus numpy
restatements of the per-variant / per-sample arithmetic of the reference's table
functions, each citing the reference lines it follows.  Used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.

Parity is PINNED: tests/test_oracle_golden.py checks everything here against the
known-answer values of the reference's own sqllogictests.
"""

from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpgen_oracle.so")


def build(force: bool = False):
    """Compile the C restatement (gcc/g++ via oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("pgen_oracle.c", "g1_seed.cpp", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def _load():
    build()
    L = C.CDLL(_SO)
    vp, u32 = C.c_void_p, C.c_uint32
    L.pgo_open.restype = vp
    L.pgo_open.argtypes = [C.c_char_p]
    L.pgo_open_mem.restype = vp
    L.pgo_open_mem.argtypes = [vp, C.c_size_t]
    L.pgo_close.argtypes = [vp]
    L.pgo_last_open_error.restype = C.c_char_p
    for f in ("pgo_variant_ct", "pgo_sample_ct", "pgo_has_dosage", "pgo_has_phase"):
        getattr(L, f).argtypes = [vp]
    L.pgo_vrtype.argtypes = [vp, u32]
    L.pgo_get_geno.argtypes = [vp, u32, vp, vp]
    L.pgo_get_raw.argtypes = [vp, u32, vp]
    L.pgo_ld_sums.argtypes = [vp, u32, u32, vp, vp]
    L.pgo_get_counts.argtypes = [vp, u32, vp, vp]
    L.pgo_get_missingness.argtypes = [vp, u32, vp, vp]
    L.pgo_get_phase.argtypes = [vp, u32, vp, vp, vp, vp]
    L.pgo_get_dosage.argtypes = [vp, u32, vp, vp]
    L.pgo_get_dcounts.argtypes = [vp, u32, vp, vp, vp, vp]
    L.pgo_counts_range.argtypes = [vp, u32, u32, vp, vp]
    L.pgo_missing_per_sample.argtypes = [vp, u32, u32, vp, vp]
    L.pgo_scan_counts_mt.argtypes = [vp, u32, u32, u32, vp]
    L.pgo_hwe_lnp.restype = C.c_double
    L.pgo_hwe_lnp.argtypes = [C.c_int32] * 3 + [u32]
    L.pgo_hwe_xchr_lnp.restype = C.c_double
    L.pgo_hwe_xchr_lnp.argtypes = [C.c_int32] * 5 + [u32]
    L.pgo_fill_g1.argtypes = [vp, C.c_size_t]
    return L


_L = _load()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def hwe_lnp(hets, hom1, hom2, midp=False):
    return _L.pgo_hwe_lnp(int(hets), int(hom1), int(hom2), 1 if midp else 0)


def hwe_xchr_lnp(fhets, fhom1, fhom2, male1, male2, midp=False):
    return _L.pgo_hwe_xchr_lnp(int(fhets), int(fhom1), int(fhom2), int(male1), int(male2), 1 if midp else 0)


def fill_g1(n_rows, n_cols):
    """src/plink_pca.cpp:517-523 -- mt19937_64(12345) + normal_distribution, row-major."""
    g = np.empty((n_rows, n_cols), dtype=np.float64)
    _L.pgo_fill_g1(_p(g), g.size)
    return g


class Pgen:
    """One .pgen opened by the C oracle."""

    def __init__(self, path=None, mem: np.ndarray | None = None):
        if path is not None:
            self._h = _L.pgo_open(path.encode())
        else:
            self._mem = np.ascontiguousarray(mem, dtype=np.uint8)  # keep alive
            self._h = _L.pgo_open_mem(_p(self._mem), self._mem.size)
        if not self._h:
            raise IOError(_L.pgo_last_open_error().decode())
        self.M = _L.pgo_variant_ct(self._h)
        self.N = _L.pgo_sample_ct(self._h)
        self.has_dosage = bool(_L.pgo_has_dosage(self._h))
        self.has_phase = bool(_L.pgo_has_phase(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _L.pgo_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _inc(include):
        return None if include is None else np.ascontiguousarray(include, dtype=np.uint8)

    def _n_out(self, include):
        return self.N if include is None else int(np.count_nonzero(include))

    def vrtype(self, v):
        return _L.pgo_vrtype(self._h, v)

    def geno(self, v, include=None):
        inc = self._inc(include)
        out = np.zeros(self._n_out(include), dtype=np.int8)
        rc = _L.pgo_get_geno(self._h, v, _p(inc), _p(out))
        if rc < 0:
            raise IOError(f"oracle decode failed for variant {v}")
        return out

    def raw(self, v):
        out = np.zeros(self.N, dtype=np.uint8)
        if _L.pgo_get_raw(self._h, v, _p(out)) != 0:
            raise IOError(f"oracle decode failed for variant {v}")
        return out

    def sample_counts(self, vidx=None, include=None):
        """read_pfile's sample-orient aggregate (src/pfile_reader.cpp:3330-3355): per included sample
        {hom_ref, het, hom_alt, missing} over the listed variants (default: all)."""
        vidx = range(self.M) if vidx is None else vidx
        n = self._n_out(include)
        out = np.zeros((n, 4), dtype=np.uint32)
        for v in vidx:
            g = self.geno(v, include)
            out[:, 1] += g == 1
            out[:, 2] += g == 2
            out[:, 3] += g == -9
        out[:, 0] = len(vidx) - out[:, 1] - out[:, 2] - out[:, 3]  # hom_ref is derived at emit time
        return out

    def ld_sums(self, va, vb, include=None):
        """{n, sum_a, sum_b, sum_ab, sum_a2, sum_b2} of plink_ld's sample loop."""
        inc = self._inc(include)
        out = np.zeros(6, dtype=np.uint64)
        if _L.pgo_ld_sums(self._h, va, vb, _p(inc), _p(out)) != 0:
            raise IOError(f"oracle decode failed for variants {va}, {vb}")
        return out

    def counts(self, v, include=None):
        inc = self._inc(include)
        out = np.zeros(4, dtype=np.uint32)
        if _L.pgo_get_counts(self._h, v, _p(inc), _p(out)) != 0:
            raise IOError(f"oracle decode failed for variant {v}")
        return out

    def counts_range(self, v0=0, v1=None, include=None):
        v1 = self.M if v1 is None else v1
        inc = self._inc(include)
        out = np.zeros((max(0, v1 - v0), 4), dtype=np.uint32)
        if _L.pgo_counts_range(self._h, v0, v1, _p(inc), _p(out)) != 0:
            raise IOError("oracle decode failed")
        return out

    def missingness(self, v, include=None):
        inc = self._inc(include)
        out = np.zeros(self._n_out(include), dtype=np.uint8)
        if _L.pgo_get_missingness(self._h, v, _p(inc), _p(out)) < 0:
            raise IOError("oracle decode failed")
        return out

    def missing_per_sample(self, v0=0, v1=None, include=None):
        v1 = self.M if v1 is None else v1
        inc = self._inc(include)
        out = np.zeros(self._n_out(include), dtype=np.uint32)
        if _L.pgo_missing_per_sample(self._h, v0, v1, _p(inc), _p(out)) != 0:
            raise IOError("oracle decode failed")
        return out

    def phase(self, v, include=None):
        inc = self._inc(include)
        n = self._n_out(include)
        g = np.zeros(n, dtype=np.int8)
        pp = np.zeros(n, dtype=np.uint8)
        pi = np.zeros(n, dtype=np.uint8)
        if _L.pgo_get_phase(self._h, v, _p(inc), _p(g), _p(pp), _p(pi)) < 0:
            raise IOError("oracle decode failed")
        return g, pp, pi

    def dosage(self, v, include=None):
        inc = self._inc(include)
        out = np.zeros(self._n_out(include), dtype=np.float64)
        if _L.pgo_get_dosage(self._h, v, _p(inc), _p(out)) < 0:
            raise IOError("oracle decode failed")
        return out

    def dcounts(self, v, include=None):
        inc = self._inc(include)
        counts = np.zeros(4, dtype=np.uint32)
        dos = np.zeros(2, dtype=np.uint64)
        r2 = C.c_double(0.0)
        if _L.pgo_get_dcounts(self._h, v, _p(inc), _p(counts), _p(dos), C.byref(r2)) != 0:
            raise IOError("oracle decode failed")
        return counts, dos, r2.value

    def scan_counts_mt(self, v0, v1, threads):
        """Reference scan structure: T threads, fetch_add(128) claims (src/plink_freq.cpp:434-488)."""
        out = np.zeros((max(0, v1 - v0), 4), dtype=np.uint32)
        _L.pgo_scan_counts_mt(self._h, v0, v1, threads, _p(out))
        return out


# ---------------------------------------------------------------------------
# table-function arithmetic
# ---------------------------------------------------------------------------

def unphased_pairs(geno, phasepresent, phaseinfo):
    """UnpackPhasedGenotypes (src/plink_common.cpp:1549-1584): [N][2] allele pairs."""
    out = np.zeros((len(geno), 2), dtype=np.int8)
    for s, g in enumerate(geno):
        if g == -9:
            out[s] = (-9, -9)
        elif g == 0:
            out[s] = (0, 0)
        elif g == 2:
            out[s] = (1, 1)
        elif g == 1:
            out[s] = (1, 0) if (phasepresent[s] and phaseinfo[s]) else (0, 1)
        else:
            out[s] = (-9, -9)
    return out


def freq_from_counts(c):
    """src/plink_freq.cpp:495-544 -> (alt_freq or None, obs_ct)."""
    obs = int(c[0]) + int(c[1]) + int(c[2])
    if obs == 0:
        return None, 0
    return (float(c[1]) + 2.0 * float(c[2])) / (2.0 * float(obs)), 2 * obs


def freq_from_dcounts(dos):
    """src/plink_freq.cpp:523-535 -> (alt_freq or None, obs_ct)."""
    total = int(dos[0]) + int(dos[1])
    if total == 0:
        return None, 0
    return float(dos[1]) / float(total), total // 16384


def ln_p_to_pvalue(ln_p):
    """src/plink_hardy.cpp:52-64."""
    if math.isnan(ln_p):
        return 1.0
    p = math.exp(ln_p)
    return min(1.0, max(0.0, p))


def hardy_from_counts(c, midp=False):
    """src/plink_hardy.cpp:572-589 -> (o_het, e_het, p_hwe) or None when no observation."""
    hom_ref, het, hom_alt = int(c[0]), int(c[1]), int(c[2])
    obs = hom_ref + het + hom_alt
    if obs == 0:
        return None
    o_het = het / obs
    p = (2.0 * hom_ref + het) / (2.0 * obs)
    e_het = 2.0 * p * (1.0 - p)
    p_hwe = ln_p_to_pvalue(hwe_lnp(het, hom_ref, hom_alt, midp))
    return o_het, e_het, p_hwe


def normalize_chrom(chrom):
    c = chrom.lower()
    return c[3:] if c.startswith("chr") else c


PAR_BOUNDS = {
    "grch38": (2781479, 155701383, 156030895),
    "grch37": (2699520, 154931044, 155260560),
}


def classify_ploidy(chrom, pos, build="grch38"):
    """src/plink_common.cpp:1960-1979 -> 'auto' | 'x' | 'y' | 'mt'."""
    c = normalize_chrom(chrom)
    if c in ("par1", "par2", "xy", "25"):
        return "auto"
    if c in ("y", "24"):
        return "y"
    if c in ("mt", "m", "26", "chrm"):
        return "mt"
    if c in ("x", "23"):
        if build in PAR_BOUNDS:
            p1, p2s, p2e = PAR_BOUNDS[build]
            if (0 < pos <= p1) or (p2s <= pos <= p2e):
                return "auto"
        return "x"
    return "auto"


def sex_aware_counts(geno, ploidy, sex):
    """ComputeSexAwareCounts (src/plink_common.cpp:1996-2108).  sex: None or uint8[N] (1 male, 2 female)."""
    r = dict(obs_allele_ct=0, alt_allele_ct=0, geno_hom_ref=0, geno_het=0, geno_hom_alt=0, geno_missing=0,
             hwe_hom_ref=0, hwe_het=0, hwe_hom_alt=0, sex_unavailable=False, hwe_defined=False)
    if ploidy in ("x", "y") and sex is None:
        r["sex_unavailable"] = True
        return r
    for i, g in enumerate(geno):
        s = 0 if sex is None else int(sex[i])
        haploid = (ploidy == "mt") or (ploidy == "y" and s == 1) or (ploidy == "x" and s == 1)
        if ploidy == "y" and s != 1:
            r["geno_missing"] += 1
        elif ploidy == "x" and s not in (1, 2):
            r["geno_missing"] += 1
        elif haploid:
            if g == 0:
                r["obs_allele_ct"] += 1
                r["geno_hom_ref"] += 1
            elif g == 2:
                r["obs_allele_ct"] += 1
                r["alt_allele_ct"] += 1
                r["geno_hom_alt"] += 1
            else:
                r["geno_missing"] += 1
        else:  # diploid stratum (chrX females)
            if g == -9:
                r["geno_missing"] += 1
            else:
                r["obs_allele_ct"] += 2
                r["alt_allele_ct"] += int(g)
                key = ("hom_ref", "het", "hom_alt")[int(g)]
                r["hwe_" + key] += 1
                r["geno_" + key] += 1
    r["hwe_defined"] = ploidy in ("x", "auto")
    return r


def score(pg: Pgen, vidx, weights, flip=None, mode="default", include=None):
    """PlinkScoreScan phase 1 (src/plink_score.cpp:575-654), one weight column per call column."""
    weights = np.asarray(weights, dtype=np.float64)
    if weights.ndim == 1:
        weights = weights.reshape(-1, 1)
    n = pg.N if include is None else int(np.count_nonzero(include))
    ncol = weights.shape[1]
    score_sum = np.zeros((n, ncol))
    dosage_sum = np.zeros(n)
    allele_ct = np.zeros(n, dtype=np.uint32)
    for i, v in enumerate(vidx):
        d = pg.dosage(int(v), include)
        nm = d != -9.0
        non_missing = int(np.count_nonzero(nm))
        if non_missing == 0:
            continue
        sum_alt = float(np.sum(d[nm]))
        fl = bool(flip[i]) if flip is not None else False
        w = weights[i]
        if mode == "center":
            mean_alt = sum_alt / non_missing
            freq = mean_alt / 2.0
            sd = math.sqrt(2.0 * freq * (1.0 - freq))
            if sd == 0.0:
                continue
            mean_scored = (2.0 - mean_alt) if fl else mean_alt
            scored = (2.0 - d) if fl else d
            std = (scored - mean_scored) / sd
            score_sum[nm] += np.outer(std[nm], w)
            allele_ct[nm] += 2
        elif mode == "no_mean_imputation":
            scored = (2.0 - d) if fl else d
            score_sum[nm] += np.outer(scored[nm], w)
            dosage_sum[nm] += scored[nm]
            allele_ct[nm] += 2
        else:
            mean_alt = sum_alt / non_missing
            alt = np.where(nm, d, mean_alt)
            scored = (2.0 - alt) if fl else alt
            score_sum += np.outer(scored, w)
            dosage_sum += scored
            allele_ct += 2
    return score_sum, dosage_sum, allele_ct


def ld_stats(sums):
    """ComputeLdStats after its sample loop (src/plink_ld.cpp:86-134) -> (r2, d_prime, obs_ct), r2/d_prime
    None when the pair has < 2 observations or a monomorphic side."""
    n = int(sums[0])
    if n < 2:
        return None, None, n
    sum_a, sum_b, sum_ab, sum_a2, sum_b2 = (float(x) for x in sums[1:6])
    dn = float(n)
    mean_a, mean_b = sum_a / dn, sum_b / dn
    cov_ab = sum_ab / dn - mean_a * mean_b
    var_a = sum_a2 / dn - mean_a * mean_a
    var_b = sum_b2 / dn - mean_b * mean_b
    if var_a < 1e-15 or var_b < 1e-15:
        return None, None, n
    r2 = (cov_ab * cov_ab) / (var_a * var_b)
    d = cov_ab / 4.0
    p_a, p_b = sum_a / (2.0 * dn), sum_b / (2.0 * dn)
    if d >= 0:
        d_max = min(p_a * (1.0 - p_b), (1.0 - p_a) * p_b)
    else:
        d_max = max(-p_a * p_b, -(1.0 - p_a) * (1.0 - p_b))
    d_prime = 0.0 if abs(d_max) < 1e-15 else d / d_max
    return r2, d_prime, n


def variant_norm(alt_freq):
    """ComputeVariantNorm (src/plink_common.cpp:1521-1533) -> (center, inv_stdev) or None if skipped."""
    if alt_freq <= 0.0 or alt_freq >= 1.0:
        return None
    return 2.0 * alt_freq, 1.0 / math.sqrt(2.0 * alt_freq * (1.0 - alt_freq))


def pca(pg: Pgen, n_pcs, include=None, v0=0, v1=None):
    """plink_pca (src/plink_pca.cpp:392-416, 517-523, 630-724, 881-959).

    numpy.linalg.svd stands in for Eigen::BDCSVD (thin U); eigenvectors are
    therefore defined up to sign.  Returns (eigenvalues[k], eigenvectors[N][k], M_eff)."""
    v1 = pg.M if v1 is None else v1
    k2 = 2 * n_pcs
    qq = (n_pcs + 1) * k2
    eff = []
    for v in range(v0, v1):
        c = pg.counts(v, include)
        obs = int(c[0]) + int(c[1]) + int(c[2])
        if obs == 0:
            continue
        af = (float(c[1]) + 2.0 * float(c[2])) / (2.0 * obs)
        nrm = variant_norm(af)
        if nrm is None:
            continue
        eff.append((v, nrm))
    M = len(eff)
    N = pg.N if include is None else int(np.count_nonzero(include))
    X = np.zeros((M, N))
    for i, (v, (center, inv_sd)) in enumerate(eff):
        g = pg.geno(v, include).astype(np.float64)
        X[i] = np.where(g == -9, 0.0, (g - center) * inv_sd)  # NormalizeGenotypes, plink_common.cpp:1535-1543
    G1 = fill_g1(N, k2)
    QQ = np.zeros((M, qq))
    for p in range(n_pcs + 1):
        Y = X @ G1                    # Step A, plink_pca.cpp:632-645
        QQ[:, p * k2:(p + 1) * k2] = Y
        if p < n_pcs:
            G1 = (X.T @ Y) / M        # Step B + MergePass, plink_pca.cpp:649-661, 940-954
    U, _, _ = np.linalg.svd(QQ, full_matrices=False)   # RunKrylovSVD, plink_pca.cpp:683-697
    BB = X.T @ U                      # Phase 3, plink_pca.cpp:664-676
    U2, S, _ = np.linalg.svd(BB, full_matrices=False)  # RunFinalSVD, plink_pca.cpp:700-720
    return (S[:n_pcs] ** 2) / M, U2[:, :n_pcs], M


# ---------------------------------------------------------------------------
# minimal companion-file readers for the tests
# ---------------------------------------------------------------------------

def load_pvar(path):
    """#CHROM POS ID REF ALT (or headerless .bim: CHROM ID CM POS ALT REF)."""
    chrom, pos, ids, ref, alt = [], [], [], [], []
    header = None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("##") or not line:
                continue
            parts = line.split("\t") if "\t" in line else line.split()
            if line.startswith("#"):
                header = [p.lstrip("#") for p in parts]
                continue
            if header is None:  # .bim
                chrom.append(parts[0]); ids.append(parts[1]); pos.append(int(parts[3]))
                alt.append(parts[4]); ref.append(parts[5])
            else:
                rec = dict(zip(header, parts))
                chrom.append(rec["CHROM"]); pos.append(int(rec["POS"])); ids.append(rec["ID"])
                ref.append(rec["REF"]); alt.append(rec["ALT"])
    return dict(chrom=chrom, pos=pos, id=ids, ref=ref, alt=alt)


def load_psam(path):
    iid, fid, sex = [], [], []
    header = None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("##") or not line:
                continue
            parts = line.split("\t") if "\t" in line else line.split()
            if line.startswith("#"):
                header = [p.lstrip("#") for p in parts]
                continue
            if header is None:  # .fam
                fid.append(parts[0]); iid.append(parts[1]); sex.append(parts[4])
            else:
                rec = dict(zip(header, parts))
                iid.append(rec["IID"]); fid.append(rec.get("FID")); sex.append(rec.get("SEX", "NA"))
    sexes = np.array([int(s) if s in ("1", "2") else 0 for s in sex], dtype=np.uint8)
    return dict(iid=iid, fid=fid, sex=sexes)
