// linalg.cpp -- see linalg.hpp.
#include "linalg.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace pgh {

namespace {

// One-sided Jacobi (Hestenes) SVD of the n x n row-major matrix r:
// r <- U_r (n x n), s <- singular values, both sorted descending.
void JacobiSvd(std::vector<double> &r, size_t n, std::vector<double> &s) {
	// work on columns: rotate pairs until all are mutually orthogonal
	const double eps = 1e-15;
	for (int sweep = 0; sweep < 60; sweep++) {
		double off = 0.0;
		for (size_t p = 0; p + 1 < n; p++) {
			for (size_t q = p + 1; q < n; q++) {
				double alpha = 0.0, beta = 0.0, gamma = 0.0;
				for (size_t i = 0; i < n; i++) {
					const double x = r[i * n + p], y = r[i * n + q];
					alpha += x * x;
					beta += y * y;
					gamma += x * y;
				}
				if (gamma == 0.0 || std::fabs(gamma) <= eps * std::sqrt(alpha * beta)) {
					continue;
				}
				off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
				const double zeta = (beta - alpha) / (2.0 * gamma);
				const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
				const double c = 1.0 / std::sqrt(1.0 + t * t);
				const double sn = c * t;
				for (size_t i = 0; i < n; i++) {
					const double x = r[i * n + p], y = r[i * n + q];
					r[i * n + p] = c * x - sn * y;
					r[i * n + q] = sn * x + c * y;
				}
			}
		}
		if (off <= eps) {
			break;
		}
	}
	s.assign(n, 0.0);
	for (size_t j = 0; j < n; j++) {
		double nrm = 0.0;
		for (size_t i = 0; i < n; i++) {
			nrm += r[i * n + j] * r[i * n + j];
		}
		s[j] = std::sqrt(nrm);
	}
	std::vector<size_t> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return s[a] > s[b]; });
	std::vector<double> u(n * n, 0.0), s2(n);
	for (size_t k = 0; k < n; k++) {
		const size_t j = order[k];
		s2[k] = s[j];
		if (s[j] > 0.0) {
			for (size_t i = 0; i < n; i++) {
				u[i * n + k] = r[i * n + j] / s[j];
			}
		}
	}
	r.swap(u);
	s.swap(s2);
}

} // namespace

void SymmetricEigen(const std::vector<double> &g, size_t n, std::vector<double> &eigenvalues,
                    std::vector<double> &v) {
	std::vector<double> a(g);
	v.assign(n * n, 0.0);
	for (size_t i = 0; i < n; i++) {
		v[i * n + i] = 1.0;
	}
	for (int sweep = 0; sweep < 100; sweep++) {
		double off = 0.0, diag = 0.0;
		for (size_t i = 0; i < n; i++) {
			diag += a[i * n + i] * a[i * n + i];
			for (size_t j = i + 1; j < n; j++) {
				off += a[i * n + j] * a[i * n + j];
			}
		}
		if (off <= 1e-32 * diag || off == 0.0) {
			break;
		}
		for (size_t p = 0; p + 1 < n; p++) {
			for (size_t q = p + 1; q < n; q++) {
				const double apq = a[p * n + q];
				if (apq == 0.0) {
					continue;
				}
				const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
				const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
				for (size_t k = 0; k < n; k++) { // rotate columns p, q
					const double x = a[k * n + p], y = a[k * n + q];
					a[k * n + p] = c * x - sn * y;
					a[k * n + q] = sn * x + c * y;
				}
				for (size_t k = 0; k < n; k++) { // rotate rows p, q
					const double x = a[p * n + k], y = a[q * n + k];
					a[p * n + k] = c * x - sn * y;
					a[q * n + k] = sn * x + c * y;
				}
				for (size_t k = 0; k < n; k++) {
					const double x = v[k * n + p], y = v[k * n + q];
					v[k * n + p] = c * x - sn * y;
					v[k * n + q] = sn * x + c * y;
				}
			}
		}
	}
	std::vector<size_t> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return a[x * n + x] > a[y * n + y]; });
	eigenvalues.resize(n);
	std::vector<double> vs(n * n);
	for (size_t k = 0; k < n; k++) {
		eigenvalues[k] = a[order[k] * n + order[k]];
		for (size_t i = 0; i < n; i++) {
			vs[i * n + k] = v[i * n + order[k]];
		}
	}
	v.swap(vs);
}

void ThinSvdInPlace(double *a, size_t m, size_t n, std::vector<double> &s) {
	if (n == 0 || m == 0) {
		s.clear();
		return;
	}
	// Householder QR, reflectors stored below the diagonal of a (v_k[k] kept in vk0)
	std::vector<double> tau(n, 0.0), vk0(n, 0.0), w(n);
	for (size_t k = 0; k < n; k++) {
		double nrm2 = 0.0;
		for (size_t i = k; i < m; i++) {
			nrm2 += a[i * n + k] * a[i * n + k];
		}
		const double nrm = std::sqrt(nrm2);
		if (nrm == 0.0) {
			continue;
		}
		const double akk = a[k * n + k];
		const double alpha = akk >= 0 ? -nrm : nrm;
		const double v0 = akk - alpha;
		const double vtv = nrm2 - akk * akk + v0 * v0; // v = column with its head replaced by v0
		if (vtv == 0.0) {
			continue;
		}
		tau[k] = 2.0 / vtv;
		vk0[k] = v0;
		// w_j = v^T a[:, j] for the trailing columns
		std::fill(w.begin(), w.end(), 0.0);
		for (size_t i = k; i < m; i++) {
			const double vi = i == k ? v0 : a[i * n + k];
			const double *row = a + i * n;
			for (size_t j = k + 1; j < n; j++) {
				w[j] += vi * row[j];
			}
		}
		for (size_t i = k; i < m; i++) {
			const double f = tau[k] * (i == k ? v0 : a[i * n + k]);
			double *row = a + i * n;
			for (size_t j = k + 1; j < n; j++) {
				row[j] -= f * w[j];
			}
		}
		a[k * n + k] = alpha; // R's diagonal; the sub-diagonal keeps v's tail
	}
	// R (upper triangle) -> SVD
	std::vector<double> r(n * n, 0.0);
	for (size_t i = 0; i < n; i++) {
		for (size_t j = i; j < n; j++) {
			r[i * n + j] = a[i * n + j];
		}
	}
	JacobiSvd(r, n, s); // r = U_r
	// U = Q * [U_r; 0]: start from the padded block and apply the reflectors in reverse
	std::vector<double> u(m * n, 0.0);
	for (size_t i = 0; i < n; i++) {
		std::copy(r.begin() + static_cast<std::ptrdiff_t>(i * n), r.begin() + static_cast<std::ptrdiff_t>((i + 1) * n),
		          u.begin() + static_cast<std::ptrdiff_t>(i * n));
	}
	for (size_t kk = n; kk-- > 0;) {
		if (tau[kk] == 0.0) {
			continue;
		}
		std::fill(w.begin(), w.end(), 0.0);
		for (size_t i = kk; i < m; i++) {
			const double vi = i == kk ? vk0[kk] : a[i * n + kk];
			const double *row = u.data() + i * n;
			for (size_t j = 0; j < n; j++) {
				w[j] += vi * row[j];
			}
		}
		for (size_t i = kk; i < m; i++) {
			const double f = tau[kk] * (i == kk ? vk0[kk] : a[i * n + kk]);
			double *row = u.data() + i * n;
			for (size_t j = 0; j < n; j++) {
				row[j] -= f * w[j];
			}
		}
	}
	std::copy(u.begin(), u.end(), a);
}

} // namespace pgh
