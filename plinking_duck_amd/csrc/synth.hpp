// synth.hpp -- seeded synthetic genotype generator shared by host and device.
//
// BASELINE.md section 3 / SURVEY.md section 8d: per-variant ALT frequency
// p_v ~ U(0.01, 0.5), genotype ~ Binomial(2, p_v), missing with a fixed
// probability.  Counter-based (keyed by seed, variant, sample) and integer-only,
// so the host twin and the HIP kernel produce bit-identical records.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define PGH_HD __host__ __device__
#else
#define PGH_HD
#endif

namespace pgh {

PGH_HD inline uint64_t Mix64(uint64_t x) {
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

struct SynthVariant {
	uint64_t key;       // per-variant hash key
	uint32_t t_hom_ref; // u <  t_hom_ref          -> 0
	uint32_t t_hom_alt; // u >= 2^32 - t_hom_alt   -> 2, otherwise 1
};

PGH_HD inline SynthVariant SynthVariantParams(uint64_t seed, uint32_t v) {
	const uint64_t s2 = Mix64(seed);
	const uint64_t h = Mix64(s2 ^ Mix64(static_cast<uint64_t>(v) + 0x9e3779b97f4a7c15ULL));
	// p in units of 2^-32: 0.01 + 0.49 * U[0,1)
	const uint64_t p_lo = 42949673ULL;   // round(0.01 * 2^32)
	const uint64_t p_span = 2104533975ULL; // round(0.49 * 2^32)
	const uint64_t pf = p_lo + (((h & 0xffffffffULL) * p_span) >> 32);
	const uint64_t qf = (1ULL << 32) - pf;
	SynthVariant r;
	r.key = s2 ^ (static_cast<uint64_t>(v) << 32);
	r.t_hom_ref = static_cast<uint32_t>((qf * qf) >> 32);
	r.t_hom_alt = static_cast<uint32_t>((pf * pf) >> 32);
	return r;
}

// 2-bit code (0 hom-ref, 1 het, 2 hom-alt, 3 missing) of sample s.
PGH_HD inline uint32_t SynthGenotype(const SynthVariant &sv, uint32_t s, uint32_t miss_threshold) {
	const uint64_t h = Mix64(sv.key ^ static_cast<uint64_t>(s) ^ 0xd1b54a32d192ed03ULL);
	const uint32_t u_miss = static_cast<uint32_t>(h >> 32);
	const uint32_t u = static_cast<uint32_t>(h);
	if (u_miss < miss_threshold) {
		return 3;
	}
	if (u < sv.t_hom_ref) {
		return 0;
	}
	if (u >= 0u - sv.t_hom_alt && sv.t_hom_alt != 0) {
		return 2;
	}
	return 1;
}

inline uint32_t SynthMissThreshold(double missing_rate) {
	if (missing_rate <= 0.0) {
		return 0;
	}
	if (missing_rate >= 1.0) {
		return 0xffffffffu;
	}
	return static_cast<uint32_t>(missing_rate * 4294967296.0);
}

} // namespace pgh
