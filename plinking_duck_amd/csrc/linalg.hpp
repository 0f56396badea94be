// linalg.hpp -- small dense host linear algebra for plink_pca's two thin SVDs
// (the reference calls Eigen::BDCSVD on the host, src/plink_pca.cpp:683-724;
// Eigen is not in this image).
#pragma once

#include <cstddef>
#include <vector>

namespace pgh {

// Thin SVD of the row-major m x n matrix a (m >= n): on return a holds U (m x n,
// orthonormal columns, same storage) and s the singular values in descending
// order.  Householder QR, then one-sided Jacobi on the n x n factor.
void ThinSvdInPlace(double *a, size_t m, size_t n, std::vector<double> &s);

// Eigen-decomposition of the symmetric n x n matrix g (row-major): eigenvalues in
// descending order, eigenvectors as the columns of v (n x n, row-major).  Cyclic Jacobi.
void SymmetricEigen(const std::vector<double> &g, size_t n, std::vector<double> &eigenvalues,
                    std::vector<double> &v);

} // namespace pgh
