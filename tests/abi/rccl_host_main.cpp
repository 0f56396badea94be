// Runs tests/abi/rccl_host.cpp's host on ONE GPU: an RCCL communicator of one rank (ncclCommInitAll), a synthetic
// dataset, pgh_pca (no collective) against pgh_pca_sharded with the RCCL all-reduce callback on the library's
// stream.  With one rank the all-reduce adds nothing, so the two must agree to the last bit of the eigenvalues; what
// the run proves is that the callback's stream handle, RCCL's launch on it and the library's ordering around it
// work together (tests/test_gpu_parity.py builds and runs this on the GPU box).
#include "pgenhip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

int RunShardedPca(const char *pgen_path, uint32_t v_begin, uint32_t v_end, ncclComm_t comm, uint32_t n_pcs,
                  const std::vector<uint32_t> &vidx, const std::vector<double> &center,
                  const std::vector<double> &inv_stdev, uint64_t n_var_total, const std::vector<double> &g1,
                  std::vector<double> &eigenvalues, std::vector<double> &eigenvectors);

int main(int argc, char **argv) {
	if (argc < 2) {
		std::fprintf(stderr, "usage: rccl_host <prefix for the synthetic files>\n");
		return 2;
	}
	const std::string prefix = argv[1];
	const uint32_t m = 3000, n = 2501, n_pcs = 4;
	char err[PGH_ERRBUF_LEN] = {0};
	if (pgh_synth_write_files(prefix.c_str(), m, n, 77, 0.02, err) != PGH_OK) {
		std::fprintf(stderr, "synth: %s\n", err);
		return 1;
	}
	const std::string path = prefix + ".pgen";
	pgh_dataset *ds = nullptr;
	if (pgh_open(path.c_str(), nullptr, 0, UINT32_MAX, &ds, err) != PGH_OK) {
		std::fprintf(stderr, "open: %s\n", err);
		return 1;
	}
	// the bind's prepass: allele frequency -> centre and inverse standard deviation, monomorphic variants dropped
	std::vector<uint32_t> counts(4ull * m);
	if (pgh_counts_range(ds, nullptr, 0, m, reinterpret_cast<uint32_t(*)[4]>(counts.data()), err) != PGH_OK) {
		std::fprintf(stderr, "counts: %s\n", err);
		return 1;
	}
	std::vector<uint32_t> vidx;
	std::vector<double> center, inv_stdev;
	for (uint32_t v = 0; v < m; v++) {
		const double obs = counts[4 * v] + counts[4 * v + 1] + counts[4 * v + 2];
		if (obs == 0) {
			continue;
		}
		const double af = (counts[4 * v + 1] + 2.0 * counts[4 * v + 2]) / (2.0 * obs);
		if (af <= 0.0 || af >= 1.0) {
			continue;
		}
		vidx.push_back(v);
		center.push_back(2.0 * af);
		inv_stdev.push_back(1.0 / std::sqrt(2.0 * af * (1.0 - af)));
	}
	std::vector<double> g1(static_cast<size_t>(n) * 2 * n_pcs);
	uint64_t state = 12345;
	for (double &x : g1) { // any full-rank start matrix serves both calls alike
		state = state * 6364136223846793005ull + 1442695040888963407ull;
		x = static_cast<double>(static_cast<int64_t>(state >> 11)) / 9007199254740992.0 - 0.5;
	}
	std::vector<double> ev_plain(n_pcs), vec_plain(static_cast<size_t>(n) * n_pcs);
	if (pgh_pca(ds, nullptr, static_cast<uint32_t>(vidx.size()), vidx.data(), center.data(), inv_stdev.data(), n_pcs,
	            g1.data(), ev_plain.data(), vec_plain.data(), err) != PGH_OK) {
		std::fprintf(stderr, "pgh_pca: %s\n", err);
		return 1;
	}
	pgh_close(ds);
	ncclComm_t comm;
	int dev = 0;
	if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) {
		std::fprintf(stderr, "ncclCommInitAll failed\n");
		return 1;
	}
	std::vector<double> ev_rccl(n_pcs), vec_rccl(static_cast<size_t>(n) * n_pcs);
	const int rc = RunShardedPca(path.c_str(), 0, UINT32_MAX, comm, n_pcs, vidx, center, inv_stdev, vidx.size(), g1, ev_rccl,
	                             vec_rccl);
	ncclCommDestroy(comm);
	if (rc != PGH_OK) {
		std::fprintf(stderr, "pgh_pca_sharded over RCCL failed: %d\n", rc);
		return 1;
	}
	double worst = 0.0;
	for (uint32_t k = 0; k < n_pcs; k++) {
		worst = std::fmax(worst, std::fabs(ev_rccl[k] - ev_plain[k]) / ev_plain[k]);
		std::printf("eigenvalue %u: %.12g (plain) %.12g (RCCL all-reduce, 1 rank)\n", k, ev_plain[k], ev_rccl[k]);
	}
	std::printf("worst relative difference %.3g over %zu effective variants\n", worst, vidx.size());
	// ... and the in-library form of the same exchange: a shard group on distinct devices (one here) all-reduces
	// through communicators the library makes itself (pgh_open_sharded -> ncclCommInitAll at the first collective)
	pgh_dataset *group = nullptr;
	if (pgh_open_sharded(path.c_str(), nullptr, 0, UINT32_MAX, &dev, 1, &group, err) != PGH_OK) {
		std::fprintf(stderr, "pgh_open_sharded: %s\n", err);
		return 1;
	}
	if (!pgh_group_uses_rccl(group)) {
		std::fprintf(stderr, "the group did not get RCCL communicators\n");
		return 1;
	}
	std::vector<double> ev_group(n_pcs), vec_group(static_cast<size_t>(n) * n_pcs);
	if (pgh_pca(group, nullptr, static_cast<uint32_t>(vidx.size()), vidx.data(), center.data(), inv_stdev.data(), n_pcs,
	            g1.data(), ev_group.data(), vec_group.data(), err) != PGH_OK) {
		std::fprintf(stderr, "pgh_pca over the group: %s\n", err);
		return 1;
	}
	pgh_close(group);
	for (uint32_t k = 0; k < n_pcs; k++) {
		worst = std::fmax(worst, std::fabs(ev_group[k] - ev_plain[k]) / ev_plain[k]);
		std::printf("eigenvalue %u: %.12g (in-library RCCL all-reduce, shard group of 1)\n", k, ev_group[k]);
	}
	return worst < 1e-12 ? 0 : 1;
}
