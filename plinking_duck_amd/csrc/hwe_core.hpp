// hwe_core.hpp -- Hardy-Weinberg exact tests, host + device.
//
// Replaces plink2::HweLnP / HweXchrLnP (plink2_stats.cc, absent from the
// reference tree; call sites src/plink_hardy.cpp:78,94).  Definitions:
//   autosomal  Wigginton, Cutler & Abecasis (2005): two-sided p = sum of P(k hets)
//              over all tables no likelier than the observed one;
//   chrX       Graffelman & Weir (2016): same rule over the joint distribution of
//              (A-allele males, female hets);
//   mid-p      subtract half the probability of the tables tied with the observed.
// Probabilities are carried relative to the modal table and advanced with the
// exact ratios P(k+2)/P(k), so 500k-sample counts neither overflow nor need
// lgamma; work is O(distance to the mode + width of the distribution).
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define PGH_HD __host__ __device__
#else
#define PGH_HD
#endif

namespace pgh {

// tables whose probability equals the observed one up to rounding count as ties
constexpr double kHweTieEps = 9.313225746154785e-10; // 2^-30

// num / den inside the walks below.  On the host the IEEE quotient; on the device v_rcp_f64 refined by two Newton
// steps and one multiply (~7 instructions instead of the ~25 of a correctly rounded FP64 division, which was most
// of k_hwe_batch): the quotient can differ from the IEEE one in its last bit, the walk's sums by ~1e-13 relative
// after its few thousand steps -- four orders below the tie band (2^-30) and the tests' 1e-12.
PGH_HD inline double HweRatio(double num, double den) {
#if defined(__HIP_DEVICE_COMPILE__)
	double x = __builtin_amdgcn_rcp(den);
	x = fma(fma(-den, x, 1.0), x, x);
	x = fma(fma(-den, x, 1.0), x, x);
	return num * x;
#else
	return num / den;
#endif
}

// P(k+2 hets) / P(k hets) for rare-allele count `rare`, common-allele count `common`
PGH_HD inline double HweStepUp(int64_t rare, int64_t common, int64_t k) {
	return 4.0 * static_cast<double>((rare - k) >> 1) * static_cast<double>((common - k) >> 1) /
	       (static_cast<double>(k + 2) * static_cast<double>(k + 1));
}

PGH_HD inline double HweLnP(int32_t obs_hets, int32_t obs_hom1, int32_t obs_hom2, uint32_t midp) {
	const int64_t n = static_cast<int64_t>(obs_hets) + obs_hom1 + obs_hom2;
	if (n <= 0) {
		return 0.0;
	}
	const int64_t hom_rare = obs_hom1 < obs_hom2 ? obs_hom1 : obs_hom2;
	const int64_t rare = 2 * hom_rare + obs_hets;
	const int64_t common = 2 * n - rare;
	// start near the expected het count, fix parity, then climb to the true mode
	int64_t mode = static_cast<int64_t>(static_cast<double>(rare) * static_cast<double>(common) /
	                                    static_cast<double>(2 * n));
	if ((mode ^ rare) & 1) {
		mode++;
	}
	if (mode > rare) {
		mode -= 2;
	}
	if (mode < 0) {
		mode += 2;
	}
	while (mode + 2 <= rare && HweStepUp(rare, common, mode) > 1.0) {
		mode += 2;
	}
	while (mode >= 2 && HweStepUp(rare, common, mode - 2) < 1.0) {
		mode -= 2;
	}
	// The walks below carry (k, homozygote-pair counts) as doubles -- all exact
	// integers < 2^53 -- so the inner loops are pure FP64: the ratio of step k,
	//   P(k+2)/P(k) = 4 * hr * hc / ((k+2)(k+1)),  hr = (rare-k)/2, hc = (common-k)/2,
	// does not depend on the running product and pipelines ahead of it.
	const double kd0 = static_cast<double>(mode);
	const double hr0 = static_cast<double>((rare - mode) >> 1);
	const double hc0 = static_cast<double>((common - mode) >> 1);
	// observed table relative to the mode
	double p_obs = 1.0;
	if (obs_hets > mode) {
		double k = kd0, hr = hr0, hc = hc0;
		for (int64_t it = (obs_hets - mode) >> 1; it > 0 && p_obs > 0.0; it--) {
			p_obs *= HweRatio(4.0 * hr * hc, (k + 2.0) * (k + 1.0));
			k += 2.0;
			hr -= 1.0;
			hc -= 1.0;
		}
	} else {
		double k = kd0, hr = hr0, hc = hc0;
		for (int64_t it = (mode - obs_hets) >> 1; it > 0 && p_obs > 0.0; it--) {
			p_obs *= HweRatio(k * (k - 1.0), 4.0 * (hr + 1.0) * (hc + 1.0));
			k -= 2.0;
			hr += 1.0;
			hc += 1.0;
		}
	}
	if (!(p_obs > 0.0)) {
		return -INFINITY; // p underflows double: exp(lnP) is 0 either way
	}
	const double hi = p_obs * (1.0 + kHweTieEps);
	const double lo = p_obs * (1.0 - kHweTieEps);
	// a term 2^-64 of the sum it joins cannot change that sum's double: the walks stop there (they ran to 1e-30
	// of the mode before: a third more steps for nothing a double can hold)
	const double negligible = lo * 0x1p-64;
	double total = 1.0;
	double tail = 0.0;
	double ties = 0.0;
	if (1.0 <= hi) {
		tail = 1.0;
		ties = 1.0 >= lo ? 1.0 : 0.0;
	}
	// The two long walks advance numerator and denominator of the step ratio by their (exact, integer-valued)
	// differences instead of recomputing the products: up,  A = 4 hr hc -> A - S, S = 4 (hr + hc) - 4 -> S - 8 and
	// B = (k+2)(k+1) -> B + T, T = 4 k + 10 -> T + 8; down, C = k (k-1) -> C - U, U = 4 k - 6 -> U - 8 and
	// D = 4 (hr+1)(hc+1) -> D + V, V = 4 (hr + hc) + 12 -> V + 8.  Four additions per step where there were three
	// updates, three multiplications and two additions; every value stays an integer below 2^53.
	{
		double p = 1.0;
		double a = 4.0 * hr0 * hc0, s_ = 4.0 * (hr0 + hc0) - 4.0;
		double b = (kd0 + 2.0) * (kd0 + 1.0), t_ = 4.0 * kd0 + 10.0;
		for (int64_t it = (rare - mode) >> 1; it > 0; it--) {
			p *= HweRatio(a, b);
			a -= s_;
			s_ -= 8.0;
			b += t_;
			t_ += 8.0;
			total += p;
			if (p <= hi) {
				tail += p;
				if (p >= lo) {
					ties += p;
				}
				if (p < negligible && p < total * 0x1p-64) {
					break;
				}
			}
		}
	}
	{
		double p = 1.0;
		double c = kd0 * (kd0 - 1.0), u_ = 4.0 * kd0 - 6.0;
		double d = 4.0 * (hr0 + 1.0) * (hc0 + 1.0), v_ = 4.0 * (hr0 + hc0) + 12.0;
		for (int64_t it = mode >> 1; it > 0; it--) {
			p *= HweRatio(c, d);
			c -= u_;
			u_ -= 8.0;
			d += v_;
			v_ += 8.0;
			total += p;
			if (p <= hi) {
				tail += p;
				if (p >= lo) {
					ties += p;
				}
				if (p < negligible && p < total * 0x1p-64) {
					break;
				}
			}
		}
	}
	if (midp) {
		tail -= 0.5 * ties;
	}
	double pv = tail / total;
	if (pv > 1.0) {
		pv = 1.0;
	}
	return log(pv);
}

// Het-count distribution of `n` diploid individuals carrying `a` copies of one
// allele, relative to its modal table: calls fn(k, rel) for every het count k
// whose relative probability is not negligible, and returns the sum of rel.
template <class Fn>
PGH_HD inline double HweWalk(int64_t n, int64_t a, Fn &&fn) {
	const int64_t b = 2 * n - a;
	const int64_t rare = a < b ? a : b;
	const int64_t common = 2 * n - rare;
	int64_t mode = n > 0 ? static_cast<int64_t>(static_cast<double>(rare) * static_cast<double>(common) /
	                                            static_cast<double>(2 * n))
	                     : 0;
	if ((mode ^ rare) & 1) {
		mode++;
	}
	if (mode > rare) {
		mode -= 2;
	}
	if (mode < 0) {
		mode += 2;
	}
	while (mode + 2 <= rare && HweStepUp(rare, common, mode) > 1.0) {
		mode += 2;
	}
	while (mode >= 2 && HweStepUp(rare, common, mode - 2) < 1.0) {
		mode -= 2;
	}
	double total = 1.0;
	fn(mode, 1.0);
	double p = 1.0;
	for (int64_t k = mode; k + 2 <= rare && p > 1e-300; k += 2) {
		p *= HweStepUp(rare, common, k);
		total += p;
		fn(k + 2, p);
	}
	p = 1.0;
	for (int64_t k = mode; k >= 2 && p > 1e-300; k -= 2) {
		p /= HweStepUp(rare, common, k - 2);
		total += p;
		fn(k - 2, p);
	}
	return total;
}

// chrX exact test.  A table is (mA = males carrying allele A, k = female hets);
// its probability factors into a hypergeometric term for the split of the A
// alleles between the sexes and the autosomal het distribution of the females:
//   P(mA, k) = C(nm, mA) C(2 nf, nA - mA) / C(nt, nA) * P_hwe(k | nf, nA - mA).
// Host only (one call per chrX variant).
inline double HweXchrLnP(int32_t female_hets, int32_t female_hom1, int32_t female_hom2, int32_t male1, int32_t male2,
                         uint32_t midp) {
	const int64_t nf = static_cast<int64_t>(female_hets) + female_hom1 + female_hom2;
	const int64_t nm = static_cast<int64_t>(male1) + male2;
	if (nf + nm <= 0) {
		return 0.0;
	}
	const int64_t nA = 2 * static_cast<int64_t>(female_hom1) + female_hets + male1;
	const int64_t m_lo = nA - 2 * nf > 0 ? nA - 2 * nf : 0;
	const int64_t m_hi = nA < nm ? nA : nm;
	// hypergeometric weight of each mA relative to mA = m_lo, in log space
	// (ratio H(m+1)/H(m) = (nm-m)(nA-m) / ((m+1)(2nf-nA+m+1)))
	const int64_t span = m_hi - m_lo + 1;
	double *lh = new double[span];
	lh[0] = 0.0;
	double lh_max = 0.0;
	for (int64_t m = m_lo; m < m_hi; m++) {
		const double r = static_cast<double>(nm - m) * static_cast<double>(nA - m) /
		                 (static_cast<double>(m + 1) * static_cast<double>(2 * nf - nA + m + 1));
		lh[m - m_lo + 1] = lh[m - m_lo] + log(r);
		if (lh[m - m_lo + 1] > lh_max) {
			lh_max = lh[m - m_lo + 1];
		}
	}
	// observed table
	double w_obs = 0.0;
	const double t_obs = HweWalk(nf, nA - male1, [&](int64_t k, double rel) {
		if (k == female_hets) {
			w_obs = rel;
		}
	});
	const double p_obs = exp(lh[male1 - m_lo] - lh_max) * w_obs / t_obs;
	double result;
	if (!(p_obs > 0.0)) {
		result = -INFINITY;
	} else {
		const double hi = p_obs * (1.0 + kHweTieEps);
		const double lo = p_obs * (1.0 - kHweTieEps);
		double total = 0.0, tail = 0.0, ties = 0.0;
		for (int64_t m = m_lo; m <= m_hi; m++) {
			const double h = exp(lh[m - m_lo] - lh_max);
			total += h;
			if (h <= lo) {
				tail += h; // every table of this column is less likely than the observed one
				continue;
			}
			const double t = HweWalk(nf, nA - m, [](int64_t, double) {});
			const double scale = h / t;
			HweWalk(nf, nA - m, [&](int64_t, double rel) {
				const double joint = rel * scale;
				if (joint <= hi) {
					tail += joint;
					if (joint >= lo) {
						ties += joint;
					}
				}
			});
		}
		if (midp) {
			tail -= 0.5 * ties;
		}
		double pv = tail / total;
		if (pv > 1.0) {
			pv = 1.0;
		}
		result = log(pv);
	}
	delete[] lh;
	return result;
}

// ---- chrX, one variant per workgroup (device) ---------------------------------------------
// The same rule as HweXchrLnP above with the table columns (mA) spread over the lanes.  The
// hypergeometric column weights come from lgamma instead of a running log-ratio, so no per-call
// array is needed: ln H(m) = ln C(nm, m) + ln C(2 nf, nA - m)  (the common denominator cancels
// against the column of the maximum).

PGH_HD inline double LnChoose(double n, double k) {
	return lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
}

struct XchrShape {
	int64_t nf, nm, nA, m_lo, m_hi;
	double lh_max;
};

PGH_HD inline double XchrColumnLog(const XchrShape &x, int64_t m) {
	return LnChoose(static_cast<double>(x.nm), static_cast<double>(m)) +
	       LnChoose(static_cast<double>(2 * x.nf), static_cast<double>(x.nA - m));
}

PGH_HD inline XchrShape MakeXchrShape(int32_t female_hets, int32_t female_hom1, int32_t female_hom2, int32_t male1,
                                      int32_t male2) {
	XchrShape x;
	x.nf = static_cast<int64_t>(female_hets) + female_hom1 + female_hom2;
	x.nm = static_cast<int64_t>(male1) + male2;
	x.nA = 2 * static_cast<int64_t>(female_hom1) + female_hets + male1;
	x.m_lo = x.nA - 2 * x.nf > 0 ? x.nA - 2 * x.nf : 0;
	x.m_hi = x.nA < x.nm ? x.nA : x.nm;
	// mode of the hypergeometric split, then its neighbours (the closed form can be off by one)
	const double nt = static_cast<double>(x.nm + 2 * x.nf);
	int64_t mode = static_cast<int64_t>((static_cast<double>(x.nA) + 1.0) * (static_cast<double>(x.nm) + 1.0) / (nt + 2.0));
	mode = mode < x.m_lo ? x.m_lo : (mode > x.m_hi ? x.m_hi : mode);
	x.lh_max = XchrColumnLog(x, mode);
	for (int64_t d = -2; d <= 2; d++) {
		const int64_t m = mode + d;
		if (m >= x.m_lo && m <= x.m_hi) {
			const double v = XchrColumnLog(x, m);
			x.lh_max = v > x.lh_max ? v : x.lh_max;
		}
	}
	return x;
}

// probability of the observed table relative to the column of the maximum (0 if it underflows)
PGH_HD inline double XchrObserved(const XchrShape &x, int32_t female_hets, int32_t male1) {
	double w_obs = 0.0;
	const double t_obs = HweWalk(x.nf, x.nA - male1, [&](int64_t k, double rel) {
		if (k == female_hets) {
			w_obs = rel;
		}
	});
	return exp(XchrColumnLog(x, male1) - x.lh_max) * w_obs / t_obs;
}

// one column's share of {total, tail, ties}
PGH_HD inline void XchrColumn(const XchrShape &x, int64_t m, double lo, double hi, double &total, double &tail,
                              double &ties) {
	const double h = exp(XchrColumnLog(x, m) - x.lh_max);
	total += h;
	if (h <= lo) {
		tail += h; // every table of this column is less likely than the observed one
		return;
	}
	const double t = HweWalk(x.nf, x.nA - m, [](int64_t, double) {});
	const double scale = h / t;
	HweWalk(x.nf, x.nA - m, [&](int64_t, double rel) {
		const double joint = rel * scale;
		if (joint <= hi) {
			tail += joint;
			if (joint >= lo) {
				ties += joint;
			}
		}
	});
}

} // namespace pgh
