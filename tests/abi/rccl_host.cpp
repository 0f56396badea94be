// The C++ host INTEGRATION.md section 4 describes: one process per GPU, variants sharded, and the
// all-reduce that pgh_pca_sharded asks for done by RCCL on the library's own stream.  Compiled (not
// run) by tests/test_abi.py: it pins the callback signature and the order of the arguments; built together with
// rccl_host_main.cpp and RUN with a one-rank communicator by tests/test_gpu_parity.py on the GPU box.
#include "pgenhip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <vector>

static int AllSum(void *ctx, void *d_buf, uint64_t count, void *stream) {
	ncclComm_t comm = static_cast<ncclComm_t>(ctx);
	return ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, comm, static_cast<hipStream_t>(stream)) == ncclSuccess
	           ? 0
	           : 1;
}

int RunShardedPca(const char *pgen_path, uint32_t v_begin, uint32_t v_end, ncclComm_t comm, uint32_t n_pcs,
                  const std::vector<uint32_t> &vidx, const std::vector<double> &center,
                  const std::vector<double> &inv_stdev, uint64_t n_var_total, const std::vector<double> &g1,
                  std::vector<double> &eigenvalues, std::vector<double> &eigenvectors) {
	char err[PGH_ERRBUF_LEN];
	pgh_dataset *ds = nullptr;
	int rc = pgh_open(pgen_path, nullptr, v_begin, v_end, &ds, err);
	if (rc != PGH_OK) {
		return rc;
	}
	rc = pgh_pca_sharded(ds, nullptr, static_cast<uint32_t>(vidx.size()), vidx.data(), center.data(),
	                     inv_stdev.data(), n_var_total, n_pcs, g1.data(), AllSum, comm, eigenvalues.data(),
	                     eigenvectors.data(), err);
	pgh_close(ds);
	return rc;
}
