// ld.hip -- plink_ld's per-pair sums as popcount algebra on two 2-bit rows (gfx950).
//
// The reference decodes both variants and walks the samples one by one with doubles
// (src/plink_ld.cpp:52-84).  Every sum it forms is an integer:
//   with lo/hi the bit-planes of a code (0 = 00, 1 = 01, 2 = 10, missing = 11) and
//   nm = "neither call is missing", restricted to nm
//     n      = popc(nm)
//     sum_a  = popc(a_lo) + 2 popc(a_hi)          sum_a2 = popc(a_lo) + 4 popc(a_hi)
//     sum_ab = popc(a_lo & b_lo) + 2 popc(a_lo & b_hi) + 2 popc(a_hi & b_lo) + 4 popc(a_hi & b_hi)
// (a non-missing call never has both planes set).  Nine popcounts per 16 sample pairs; the
// shell then applies the reference's double arithmetic to the exact sums.
//
// A workgroup (rows >= 4 KiB) or a wave (shorter rows) takes one anchor row against up to four
// consecutive partner rows, so the anchor's words are loaded once for the four.
#include "ld.hpp"

namespace pgh {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kLdPartners = 4;
constexpr uint32_t kEven = 0x55555555u;

struct PairAcc {
	uint32_t n = 0, a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0, ll = 0, lh = 0, hl = 0, hh = 0;
};

__device__ inline void Tally(PairAcc &acc, uint32_t a, uint32_t a_ok, uint32_t b, uint32_t keep) {
	// a_ok: even bits set where the anchor call is present (and the sample is kept)
	const uint32_t nm = a_ok & ~(b & (b >> 1)) & keep;
	const uint32_t alo = a & nm, ahi = (a >> 1) & nm;
	const uint32_t blo = b & nm, bhi = (b >> 1) & nm;
	acc.n += __popc(nm);
	acc.a_lo += __popc(alo);
	acc.a_hi += __popc(ahi);
	acc.b_lo += __popc(blo);
	acc.b_hi += __popc(bhi);
	acc.ll += __popc(alo & blo);
	acc.lh += __popc(alo & bhi);
	acc.hl += __popc(ahi & blo);
	acc.hh += __popc(ahi & bhi);
}

__device__ inline uint32_t WaveSum32(uint32_t v) {
	for (int d = 32; d > 0; d >>= 1) {
		v += __shfl_xor(v, d, 64);
	}
	return v;
}

// LANES = 256: one workgroup per task; LANES = 64: one wave per task, four tasks per workgroup
template <uint32_t LANES, bool MASKED>
__global__ __launch_bounds__(256) void k_ld_pairs(const uint8_t *__restrict__ rows, uint64_t pitch, uint32_t sample_ct,
                                                  uint32_t n_rows, const LdTask *__restrict__ tasks, uint32_t n_tasks,
                                                  const uint8_t *__restrict__ mask2, uint32_t (*out)[6],
                                                  uint32_t *__restrict__ bad_task) {
	constexpr uint32_t kPerBlock = 256 / LANES;
	__shared__ uint32_t s_part[4][kLdPartners][9];
	const uint32_t lane_in_unit = threadIdx.x % LANES;
	const uint32_t unit = threadIdx.x / LANES;
	const uint32_t t = blockIdx.x * kPerBlock + unit;
	bool live = t < n_tasks; // uniform per unit
	LdTask task {0, 0, 0, 0};
	if (live) {
		task = tasks[t];
		// A task the host could not have built (no partners, too many, rows outside the resident matrix) means
		// the task list did not reach the device intact.  It reads nothing, and it is reported: the smallest
		// offending task index + 1 lands in *bad_task and the host fails the call (PGH_ERR_DEVICE).
		live = task.n_b != 0 && task.n_b <= kLdPartners && task.a_row < n_rows && task.b_row < n_rows &&
		       task.n_b <= n_rows - task.b_row;
		if (!live && lane_in_unit == 0) {
			atomicMin(bad_task, t + 1);
		}
	}
	const uint32_t n_vec = ((sample_ct + 3) / 4 + 15) / 16; // 16-byte vectors that hold calls
	const u32x4 *a_ptr = reinterpret_cast<const u32x4 *>(rows + static_cast<uint64_t>(task.a_row) * pitch);
	const u32x4 *b_ptr = reinterpret_cast<const u32x4 *>(rows + static_cast<uint64_t>(task.b_row) * pitch);
	const u32x4 *m_ptr = reinterpret_cast<const u32x4 *>(mask2);
	const uint64_t pitch_vec = pitch / 16;
	PairAcc acc[kLdPartners];
	if (live) {
		for (uint32_t i = lane_in_unit; i < n_vec; i += LANES) {
			const u32x4 a = a_ptr[i];
			u32x4 keep = {kEven, kEven, kEven, kEven};
			if (MASKED) {
				keep = m_ptr[i];
			}
			u32x4 b[kLdPartners];
#pragma unroll
			for (uint32_t p = 0; p < kLdPartners; p++) {
				// partners past n_b re-read the last real one (their sums are dropped)
				const uint32_t q = p < task.n_b ? p : task.n_b - 1;
				b[p] = b_ptr[q * pitch_vec + i];
			}
#pragma unroll
			for (int w = 0; w < 4; w++) {
				const uint32_t a_ok = ~(a[w] & (a[w] >> 1)) & kEven;
#pragma unroll
				for (uint32_t p = 0; p < kLdPartners; p++) {
					Tally(acc[p], a[w], a_ok, b[p][w], keep[w]);
				}
			}
		}
	}
	// zero pad past N decodes as two present hom-ref calls: it only ever reaches n
	const uint32_t pad = MASKED ? 0u : n_vec * 64u - sample_ct;
#pragma unroll
	for (uint32_t p = 0; p < kLdPartners; p++) {
		uint32_t v[9] = {acc[p].n,    acc[p].a_lo, acc[p].a_hi, acc[p].b_lo, acc[p].b_hi,
		                 acc[p].ll,   acc[p].lh,   acc[p].hl,   acc[p].hh};
#pragma unroll
		for (int k = 0; k < 9; k++) {
			v[k] = WaveSum32(v[k]);
		}
		if (LANES == 256) {
			if ((threadIdx.x & 63u) == 0) {
#pragma unroll
				for (int k = 0; k < 9; k++) {
					s_part[threadIdx.x >> 6][p][k] = v[k];
				}
			}
		} else if (lane_in_unit == 0 && live && p < task.n_b) {
			uint32_t *o = out[task.out_base + p];
			o[0] = v[0] - pad;
			o[1] = v[1] + 2u * v[2];
			o[2] = v[3] + 2u * v[4];
			o[3] = v[5] + 2u * (v[6] + v[7]) + 4u * v[8];
			o[4] = v[1] + 4u * v[2];
			o[5] = v[3] + 4u * v[4];
		}
	}
	if (LANES == 256) {
		__syncthreads();
		if (threadIdx.x < kLdPartners && live && threadIdx.x < task.n_b) {
			const uint32_t p = threadIdx.x;
			uint32_t v[9];
#pragma unroll
			for (int k = 0; k < 9; k++) {
				v[k] = s_part[0][p][k] + s_part[1][p][k] + s_part[2][p][k] + s_part[3][p][k];
			}
			uint32_t *o = out[task.out_base + p];
			o[0] = v[0] - pad;
			o[1] = v[1] + 2u * v[2];
			o[2] = v[3] + 2u * v[4];
			o[3] = v[5] + 2u * (v[6] + v[7]) + 4u * v[8];
			o[4] = v[1] + 4u * v[2];
			o[5] = v[3] + 4u * v[4];
		}
	}
}

} // namespace

hipError_t LaunchLdPairs(const RowView &view, uint32_t n_rows, const LdTask *tasks, uint32_t n_tasks,
                         const uint8_t *mask2, uint32_t (*out)[6], uint32_t *bad_task, hipStream_t stream) {
	if (n_tasks == 0) {
		return hipSuccess;
	}
	const bool wide = view.record_bytes >= 4096;
#define PGH_LD(LANES, MASKED, BLOCKS)                                                                                  \
	hipLaunchKernelGGL((k_ld_pairs<LANES, MASKED>), dim3(BLOCKS), dim3(256), 0, stream, view.rows, view.pitch,         \
	                   view.sample_ct, n_rows, tasks, n_tasks, mask2, out, bad_task)
	if (wide) {
		if (mask2) {
			PGH_LD(256, true, n_tasks);
		} else {
			PGH_LD(256, false, n_tasks);
		}
	} else {
		const uint32_t blocks = (n_tasks + 3) / 4;
		if (mask2) {
			PGH_LD(64, true, blocks);
		} else {
			PGH_LD(64, false, blocks);
		}
	}
#undef PGH_LD
	return hipGetLastError();
}

} // namespace pgh
