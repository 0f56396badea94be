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

namespace {

// Householder tridiagonalisation + implicit-shift QL, for the larger matrices (plink_pca's final 220 x 220 Gram
// matrix: ~6 n^3 contiguous flops instead of Jacobi's ~10 sweeps x 3 n^3).  qt holds the accumulated orthogonal
// factor as ROWS (row j = j-th basis vector) so that both the reflector updates and the QL rotations run over
// contiguous memory.  On return d = eigenvalues (unordered), row j of qt = eigenvector of d[j].
bool TridiagonalEigen(std::vector<double> &a, size_t n, std::vector<double> &d, std::vector<double> &qt) {
	std::vector<double> e(n, 0.0), v(n), pv(n);
	qt.assign(n * n, 0.0);
	for (size_t i = 0; i < n; i++) {
		qt[i * n + i] = 1.0;
	}
	// A <- H_k A H_k for k = 0 .. n-3, H_k = I - beta v v^T zeroing column k below the subdiagonal (the matrix is
	// kept full and symmetric: row k of the trailing block is read instead of column k)
	for (size_t k = 0; k + 2 < n; k++) {
		const size_t m = n - k - 1; // order of the trailing block, rows/cols k+1 ..
		double norm2 = 0.0;
		for (size_t i = 0; i < m; i++) {
			v[i] = a[k * n + k + 1 + i];
			norm2 += v[i] * v[i];
		}
		const double tail2 = norm2 - v[0] * v[0];
		if (tail2 == 0.0) {
			continue; // already tridiagonal in this column
		}
		const double alpha = v[0] >= 0.0 ? -std::sqrt(norm2) : std::sqrt(norm2);
		v[0] -= alpha;
		const double beta = 2.0 / (tail2 + v[0] * v[0]);
		// p = beta A22 v; w = p - (beta/2)(v^T p) v; A22 -= v w^T + w v^T
		double vp = 0.0;
		for (size_t i = 0; i < m; i++) {
			const double *row = &a[(k + 1 + i) * n + k + 1];
			double sum = 0.0;
			for (size_t j = 0; j < m; j++) {
				sum += row[j] * v[j];
			}
			pv[i] = beta * sum;
			vp += v[i] * pv[i];
		}
		const double half = 0.5 * beta * vp;
		for (size_t i = 0; i < m; i++) {
			pv[i] -= half * v[i];
		}
		for (size_t i = 0; i < m; i++) {
			double *row = &a[(k + 1 + i) * n + k + 1];
			const double vi = v[i], wi = pv[i];
			for (size_t j = 0; j < m; j++) {
				row[j] -= vi * pv[j] + wi * v[j];
			}
		}
		a[k * n + k + 1] = a[(k + 1) * n + k] = alpha;
		for (size_t i = 1; i < m; i++) {
			a[k * n + k + 1 + i] = a[(k + 1 + i) * n + k] = 0.0;
		}
		// Q <- Q H_k: with the basis vectors as rows of qt, rows k+1 .. of qt mix: qt_rows -= beta v (v^T qt_rows)
		std::fill(pv.begin(), pv.end(), 0.0);
		std::vector<double> &acc = pv; // reused as the length-n accumulator v^T qt[k+1.., :]
		acc.assign(n, 0.0);
		for (size_t i = 0; i < m; i++) {
			const double *row = &qt[(k + 1 + i) * n];
			const double vi = v[i];
			for (size_t j = 0; j < n; j++) {
				acc[j] += vi * row[j];
			}
		}
		for (size_t i = 0; i < m; i++) {
			double *row = &qt[(k + 1 + i) * n];
			const double bv = beta * v[i];
			for (size_t j = 0; j < n; j++) {
				row[j] -= bv * acc[j];
			}
		}
	}
	d.resize(n);
	for (size_t i = 0; i < n; i++) {
		d[i] = a[i * n + i];
		e[i] = i + 1 < n ? a[i * n + i + 1] : 0.0;
	}
	// implicit-shift QL on (d, e); every plane rotation is applied to rows i, i+1 of qt
	for (size_t l = 0; l < n; l++) {
		for (int iter = 0;; iter++) {
			size_t m = l;
			for (; m + 1 < n; m++) {
				const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
				if (std::fabs(e[m]) <= 2.3e-16 * dd) {
					break;
				}
			}
			if (m == l) {
				break;
			}
			if (iter == 60) {
				return false;
			}
			double gg = (d[l + 1] - d[l]) / (2.0 * e[l]);
			double r = std::hypot(gg, 1.0);
			gg = d[m] - d[l] + e[l] / (gg + (gg >= 0.0 ? r : -r));
			double sn = 1.0, cs = 1.0, p = 0.0;
			bool underflow = false;
			for (size_t i = m; i-- > l;) {
				double f = sn * e[i];
				const double b = cs * e[i];
				r = std::hypot(f, gg);
				e[i + 1] = r;
				if (r == 0.0) {
					d[i + 1] -= p;
					e[m] = 0.0;
					underflow = true;
					break;
				}
				sn = f / r;
				cs = gg / r;
				gg = d[i + 1] - p;
				r = (d[i] - gg) * sn + 2.0 * cs * b;
				p = sn * r;
				d[i + 1] = gg + p;
				gg = cs * r - b;
				double *lo = &qt[i * n], *hi = &qt[(i + 1) * n];
				for (size_t k = 0; k < n; k++) {
					f = hi[k];
					hi[k] = sn * lo[k] + cs * f;
					lo[k] = cs * lo[k] - sn * f;
				}
			}
			if (underflow) {
				continue;
			}
			d[l] -= p;
			e[l] = gg;
			e[m] = 0.0;
		}
	}
	return true;
}

} // namespace

void SymmetricEigen(const std::vector<double> &g, size_t n, std::vector<double> &eigenvalues,
                    std::vector<double> &v) {
	if (n > 32) {
		std::vector<double> work(g), d, qt;
		if (TridiagonalEigen(work, n, d, qt)) {
			std::vector<size_t> order(n);
			std::iota(order.begin(), order.end(), 0);
			std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return d[x] > d[y]; });
			eigenvalues.resize(n);
			v.assign(n * n, 0.0);
			for (size_t k = 0; k < n; k++) {
				eigenvalues[k] = d[order[k]];
				for (size_t i = 0; i < n; i++) {
					v[i * n + k] = qt[order[k] * n + i];
				}
			}
			return;
		}
		// (no convergence in 60 QL steps for some eigenvalue: fall through to Jacobi, which cannot fail)
	}
	// Cyclic Jacobi on contiguous rows.  A rotation in the (p, q) plane is A <- J^T A J: rows p and q are
	// rotated over their whole length (two contiguous, vectorisable passes), and because A stays symmetric the
	// new columns p and q are those rows again -- copied, not recomputed; the 2 x 2 block gets its closed form.
	// The eigenvectors are kept as ROWS of vt for the same reason.  (The first form rotated columns in place,
	// 4 n strided read-modify-writes per rotation: 90 ms for plink_pca's 220 x 220 Gram matrix, now ~20.)
	std::vector<double> a(g), vt(n * n, 0.0);
	for (size_t i = 0; i < n; i++) {
		vt[i * n + i] = 1.0;
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
				const double app = a[p * n + p], aqq = a[q * n + q];
				const double theta = (aqq - app) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
				const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
				double *rp = &a[p * n], *rq = &a[q * n];
				for (size_t k = 0; k < n; k++) {
					const double x = rp[k], y = rq[k];
					rp[k] = c * x - sn * y;
					rq[k] = sn * x + c * y;
				}
				rp[p] = app - t * apq;
				rq[q] = aqq + t * apq;
				rp[q] = rq[p] = 0.0;
				for (size_t k = 0; k < n; k++) {
					if (k != p && k != q) {
						a[k * n + p] = rp[k];
						a[k * n + q] = rq[k];
					}
				}
				double *vp = &vt[p * n], *vq = &vt[q * n];
				for (size_t k = 0; k < n; k++) {
					const double x = vp[k], y = vq[k];
					vp[k] = c * x - sn * y;
					vq[k] = sn * x + c * y;
				}
			}
		}
	}
	std::vector<size_t> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return a[x * n + x] > a[y * n + y]; });
	eigenvalues.resize(n);
	v.assign(n * n, 0.0);
	for (size_t k = 0; k < n; k++) {
		eigenvalues[k] = a[order[k] * n + order[k]];
		for (size_t i = 0; i < n; i++) {
			v[i * n + k] = vt[order[k] * n + i];
		}
	}
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
