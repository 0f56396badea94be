// oracle/g1_seed.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// The reference seeds plink_pca's N x 2k start matrix with libstdc++'s
// std::normal_distribution<double> over std::mt19937_64(12345), filled row-major
// (src/plink_pca.cpp:517-523).  The golden eigenvalues depend on that exact
// stream, so the oracle draws it from the same library calls.
#include <cstddef>
#include <random>

extern "C" void pgo_fill_g1(double *out, size_t n) {
	std::mt19937_64 rng(12345);
	std::normal_distribution<double> dist(0.0, 1.0);
	for (size_t i = 0; i < n; i++) {
		out[i] = dist(rng);
	}
}
