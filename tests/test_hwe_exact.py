"""The product's Hardy-Weinberg exact test (hwe_core.hpp: ratio recurrence from the mode, tables within 2^-30
relative of the observed one count as ties) against an EXACT enumeration in rational arithmetic -- the adjudicator
neither the product nor the oracle (both floating point) can be for each other.

Definitions (src/plink_hardy.cpp:52-95 -> plink2::HweLnP; Wigginton, Cutler & Abecasis 2005): with n individuals and
a copies of the rarer allele, P(k hets) is proportional to  n! / (hom_r! k! hom_c!) * 2^k ; the two-sided p is the sum
over tables no likelier than the observed one; mid-p subtracts half the probability of the tables EQUAL to it.

Exactly equal tables exist (e.g. n = 3, a = 3: P(1 het) = P(3 hets)), and they are where a floating-point rule can land
on the wrong side: every such case up to n = 40 is enumerated, larger n are sampled up to 2,000, and the smallest
relative gap between two DIFFERENT table probabilities is checked to be far outside the 2^-30 band."""

import math
from fractions import Fraction

import numpy as np
import pytest


def exact_tables(n, a):
    """{k: Fraction} relative probabilities of the het counts k = a, a-2, ... (a = copies of one allele, any)."""
    b = 2 * n - a
    rare, common = min(a, b), max(a, b)
    k = rare & 1
    p = Fraction(1)
    out = {k: p}
    while k + 2 <= rare:
        # P(k+2) / P(k) = 4 hr hc / ((k+2)(k+1)),  hr = (rare-k)/2, hc = (common-k)/2
        p = p * Fraction(4 * ((rare - k) // 2) * ((common - k) // 2), (k + 2) * (k + 1))
        k += 2
        out[k] = p
    return out


def exact_p(n_hom1, n_het, n_hom2, midp):
    n = n_hom1 + n_het + n_hom2
    a = 2 * n_hom1 + n_het
    t = exact_tables(n, a)
    obs = t[n_het]
    total = sum(t.values())
    tail = sum(p for p in t.values() if p <= obs)
    if midp:
        tail -= sum(p for p in t.values() if p == obs) / 2
    return tail / total


def product_p(lib, hom1, het, hom2, midp):
    return math.exp(lib.hwe_lnp(het, hom1, hom2, midp))


def test_every_table_up_to_forty_individuals(lib):
    ties = 0
    for n in range(1, 41):
        for hom1 in range(n + 1):
            for het in range(n - hom1 + 1):
                hom2 = n - hom1 - het
                t = exact_tables(n, 2 * hom1 + het)
                ties += sum(1 for k, p in t.items() if k != het and p == t[het])
                for midp in (False, True):
                    want = exact_p(hom1, het, hom2, midp)
                    got = product_p(lib, hom1, het, hom2, midp)
                    assert got == pytest.approx(float(want), rel=1e-12, abs=1e-300), (hom1, het, hom2, midp)
    assert ties >= 40  # exactly tied tables do occur at small n (56 of them up to 40): the rule was exercised


@pytest.mark.parametrize("n", [100, 257, 500, 1000, 2000])
def test_sampled_tables_and_constructed_ties(lib, oracle, n):
    rng = np.random.default_rng(n)
    cases = set()
    for a in sorted(set(rng.integers(1, n + 1, size=14).tolist()) | {1, 2, 3, n - 1, n}):
        t = exact_tables(n, a)
        ks = sorted(t)
        mode = max(ks, key=lambda k: t[k])
        picks = {ks[0], ks[-1], mode} | {k for k in (mode - 2, mode + 2, mode - 20, mode + 20) if k in t}
        picks |= set(rng.choice(ks, size=min(len(ks), 4), replace=False).tolist())
        # exactly tied tables, if this (n, a) has any: both members become observed tables
        by_p = {}
        for k, p in t.items():
            by_p.setdefault(p, []).append(k)
        for group in by_p.values():
            if len(group) > 1:
                picks |= set(group)
        for k in picks:
            cases.add((a, int(k)))
    for a, het in sorted(cases):
        hom1 = (a - het) // 2
        hom2 = n - het - hom1
        if hom2 < 0:
            continue
        for midp in (False, True):
            want = float(exact_p(hom1, het, hom2, midp))
            got = product_p(lib, hom1, het, hom2, midp)
            assert got == pytest.approx(want, rel=1e-9, abs=1e-300), (n, hom1, het, hom2, midp)
            # the oracle's lgamma form is held to the same answer at its own tolerance (1e-6 in ln p)
            assert oracle.hwe_lnp(het, hom1, hom2, midp) == pytest.approx(math.log(want) if want > 0 else -math.inf, abs=1e-6)


def test_no_two_different_tables_come_within_the_tie_band():
    """2^-30 is safe as long as two tables with DIFFERENT exact probabilities never sit that close: the smallest
    relative gap found for n <= 160 (every a) and for sampled (n, a) up to 2,000."""
    band = 2.0 ** -30
    worst = (1.0, None)
    rng = np.random.default_rng(7)
    todo = [(n, a) for n in range(2, 161) for a in range(1, n + 1)]
    todo += [(int(n), int(a)) for n in (400, 1000, 2000) for a in rng.integers(2, n + 1, size=12)]
    for n, a in todo:
        t = exact_tables(n, a)
        # floats order the candidates; exact arithmetic decides the close ones
        logs = sorted((math.log(p.numerator) - math.log(p.denominator), k) for k, p in t.items())
        for (l0, k0), (l1, k1) in zip(logs, logs[1:]):
            if l1 - l0 < 1e-6 and t[k0] != t[k1]:
                gap = abs(float((t[k1] - t[k0]) / t[k0]))
                if gap < worst[0]:
                    worst = (gap, (n, a, k0, k1))
    assert worst[0] > 1e3 * band, worst


@pytest.mark.gpu
def test_device_exact_tests_against_the_rational_enumeration(gpu_lib):
    """k_hwe_batch (the same walk with v_rcp_f64 + two Newton steps for the quotients) on every table of n <= 24 and
    on sampled tables of 2,000 individuals."""
    L = gpu_lib
    rows, want = [], {False: [], True: []}
    for n in list(range(1, 25)) + [2000]:
        rng = np.random.default_rng(n)
        if n <= 24:
            combos = [(h1, het, n - h1 - het) for h1 in range(n + 1) for het in range(n - h1 + 1)]
        else:
            combos = []
            for a in rng.integers(1, n + 1, size=10):
                ks = sorted(exact_tables(n, int(a)))
                for het in rng.choice(ks, size=min(len(ks), 6), replace=False):
                    h1 = (int(a) - int(het)) // 2
                    if n - int(het) - h1 >= 0:
                        combos.append((h1, int(het), n - int(het) - h1))
        for h1, het, h2 in combos:
            rows.append((h1, het, h2, 0))
            for midp in (False, True):
                want[midp].append(float(exact_p(h1, het, h2, midp)))
    counts = np.array(rows, dtype=np.uint32)
    for midp in (False, True):
        got = np.exp(L.hwe_lnp_batch(counts, midp))
        assert np.allclose(got, np.array(want[midp]), rtol=1e-9, atol=1e-300)
