// ld.hpp -- pairwise genotype-correlation sums on the packed rows (definitions in ld.hip).
#pragma once

#include "kernels.hpp"

namespace pgh {

// One unit of work: an anchor row against 1..4 CONSECUTIVE partner rows (rows relative to
// view.rows); results go to out[out_base .. out_base + n_b).
struct LdTask {
	uint32_t a_row;
	uint32_t b_row;
	uint32_t n_b;
	uint32_t out_base;
};

// out[p] = {n, sum_a, sum_b, sum_ab, sum_a2, sum_b2} over the samples at which BOTH variants are
// non-missing (and which mask2 keeps, if given) -- the five sums and the count of the reference's
// ComputeLdStats (src/plink_ld.cpp:52-84), as exact integers.
// n_rows = resident rows behind view.rows.  *bad_task (device, preset to UINT32_MAX by the caller) receives
// 1 + the index of the first task that is not well-formed (n_b outside 1..4, a row >= n_rows); such a task
// reads and writes nothing.
hipError_t LaunchLdPairs(const RowView &view, uint32_t n_rows, const LdTask *tasks, uint32_t n_tasks,
                         const uint8_t *mask2, uint32_t (*out)[6], uint32_t *bad_task, hipStream_t stream);

} // namespace pgh
