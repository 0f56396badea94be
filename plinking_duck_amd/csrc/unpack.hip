// unpack.hip -- 2-bit -> int8 unpack with validity words (PgrGet + GenoarrToBytesMinus9), gfx950.
//
// Data layout: the genotype matrix is variant-major; row v holds ceil(N/4)
// bytes of packed 2-bit calls (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4) followed by zero bytes up to `pitch`
// (a multiple of 16, so every row can be streamed as whole 16-byte lanes and
// the pad decodes as hom-ref, which every kernel cancels against N).
#include "device_utils.hpp"

#include <algorithm>
#include "kernels.hpp"

#include <cstdlib>

namespace pgh {

namespace {

// ---------------------------------------------------------------------------
// 2-bit -> int8 unpack
// ---------------------------------------------------------------------------

// 8 bits (4 calls) -> 4 bytes, one call per byte
__device__ __forceinline__ uint32_t Spread4(uint32_t x) {
	uint32_t t = (x | (x << 12)) & 0x000f000fu;
	return (t | (t << 6)) & 0x03030303u;
}

__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ rows, uint64_t pitch, uint32_t sample_ct,
                                                uint32_t v_first, uint32_t v_count, int8_t *__restrict__ out,
                                                uint64_t out_pitch, uint64_t *__restrict__ validity,
                                                uint32_t fill4) {
	const uint32_t dwords = (sample_ct + 15) / 16;           // input dwords holding data
	const uint32_t val_words16 = ((sample_ct + 63) / 64) * 4; // uint16 slots per validity row
	const uint32_t d = blockIdx.x * 256u + threadIdx.x;
	if (d >= val_words16) {
		return;
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint32_t valid16 = 0;
		if (d < dwords) {
			const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(row) + d);
			uint4 o;
			uint32_t vbits = 0;
			uint32_t *op = &o.x;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t t = Spread4((w >> (8 * k)) & 0xffu);
				const uint32_t miss = t & (t >> 1) & 0x01010101u; // 1 in each missing byte
				const uint32_t mm = miss * 0xffu;                 // 0xff in each missing byte
				op[k] = (t & ~mm) | (fill4 & mm);
				vbits |= (((miss * 0x01020408u) >> 24) & 0xfu) << (4 * k);
			}
			valid16 = ~vbits & 0xffffu;
			const uint32_t left = sample_ct - d * 16u;
			if (left < 16u) {
				valid16 &= (1u << left) - 1u;
			}
			if (out) {
				StoreStream(reinterpret_cast<uint4 *>(out + static_cast<uint64_t>(i) * out_pitch) + d, o);
			}
		}
		if (validity) {
			uint16_t *vrow = reinterpret_cast<uint16_t *>(validity + static_cast<uint64_t>(i) * (val_words16 / 4));
			vrow[d] = static_cast<uint16_t>(valid16);
		}
	}
}

// Wide form for long rows: a lane takes 16 bytes (64 calls), expands them to 64 output
// bytes + one 64-bit validity word, and the wave's 4 KiB of output goes through LDS so
// that every global store instruction writes 1 KiB contiguous (lane-major -> piece-major).
template <bool NT_STORE>
__global__ __launch_bounds__(256) void k_unpack_wide(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t sample_ct, uint32_t v_first, uint32_t v_count,
                                                     int8_t *__restrict__ out, uint64_t out_pitch,
                                                     uint64_t *__restrict__ validity, uint32_t fill4) {
	__shared__ uint4 s_tile[4][256]; // per wave: 64 lanes x 4 pieces of 16 bytes
	const uint32_t chunks = (sample_ct + 63) / 64; // 16-byte input chunks == validity words per row
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	const uint32_t wave_col0 = blockIdx.x * 256u + wave * 64u;
	if (wave_col0 >= chunks) {
		return; // whole wave past the row (no barriers in this kernel)
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint4 w = make_uint4(0, 0, 0, 0);
		if (col < chunks) {
			w = LoadStream(reinterpret_cast<const uint4 *>(row) + col);
		}
		const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
		uint64_t vbits = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			uint4 o;
			uint32_t *op = &o.x;
			uint32_t miss16 = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t t = Spread4((ws[j] >> (8 * k)) & 0xffu);
				const uint32_t miss = t & (t >> 1) & 0x01010101u;
				const uint32_t mm = miss * 0xffu;
				op[k] = (t & ~mm) | (fill4 & mm);
				miss16 |= (((miss * 0x01020408u) >> 24) & 0xfu) << (4 * k);
			}
			vbits |= static_cast<uint64_t>(~miss16 & 0xffffu) << (16 * j);
			s_tile[wave][lane * 4u + j] = o;
		}
		// same-wave LDS exchange: LDS serves one wave's instructions in order, so a
		// wavefront-scope fence (ordering for the compiler, a waitcnt for the hardware) is
		// all the tile needs -- no workgroup barrier
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		if (out) {
			uint8_t *orow = reinterpret_cast<uint8_t *>(out) + static_cast<uint64_t>(i) * out_pitch +
			                static_cast<uint64_t>(wave_col0) * 64u;
			const uint64_t row_left = out_pitch - static_cast<uint64_t>(wave_col0) * 64u;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t piece = k * 64u + lane;
				const uint4 o = s_tile[wave][piece];
				if (static_cast<uint64_t>(piece) * 16u + 16u <= row_left) {
					if (NT_STORE) {
						StoreStream(reinterpret_cast<uint4 *>(orow) + piece, o);
					} else {
						reinterpret_cast<uint4 *>(orow)[piece] = o;
					}
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // reads done before the next row's writes
		if (validity && col < chunks) {
			const uint32_t left = sample_ct - col * 64u;
			if (left < 64u) {
				vbits &= (1ull << left) - 1ull;
			}
			validity[static_cast<uint64_t>(i) * chunks + col] = vbits;
		}
	}
}

__global__ __launch_bounds__(256) void k_unpack_subset(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                       uint32_t v_first, uint32_t v_count,
                                                       const uint32_t *__restrict__ sel, uint32_t n_out,
                                                       int8_t *__restrict__ out, uint64_t out_pitch,
                                                       uint64_t *__restrict__ validity, int32_t fill) {
	// one lane per 16 output samples: gathers their 2-bit calls from the raw row
	const uint32_t val_words16 = ((n_out + 63) / 64) * 4;
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	if (g >= val_words16) {
		return;
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint32_t valid16 = 0;
		const uint32_t k0 = g * 16u;
		if (k0 < n_out) {
			uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
			for (uint32_t j = 0; j < 16; j++) {
				const uint32_t k = k0 + j;
				uint32_t byte = 0;
				if (k < n_out) {
					const uint32_t s = sel[k];
					const uint32_t code = (row[s >> 2] >> (2 * (s & 3))) & 3u;
					if (code == 3u) {
						byte = static_cast<uint32_t>(fill) & 0xffu;
					} else {
						byte = code;
						valid16 |= 1u << j;
					}
				}
				o[j >> 2] |= byte << (8 * (j & 3));
			}
			if (out) {
				uint4 q = {o[0], o[1], o[2], o[3]};
				reinterpret_cast<uint4 *>(out + static_cast<uint64_t>(i) * out_pitch)[g] = q;
			}
		}
		if (validity) {
			uint16_t *vrow = reinterpret_cast<uint16_t *>(validity + static_cast<uint64_t>(i) * (val_words16 / 4));
			vrow[g] = static_cast<uint16_t>(valid16);
		}
	}
}

// Sample-major unpack (read_pfile orient := 'sample', src/pfile_reader.cpp:1560-1720: the reference pre-reads
// every effective variant into a variants x samples matrix and emits one row per sample).  out[k][j] = call
// of output sample k at listed variant j, missing -> fill.  A workgroup moves a 64-variant x 64-sample tile
// through LDS: reads run along the samples of a row (one dword = 16 calls per lane), writes along the variants
// of a sample (16 bytes per lane).
template <bool SUBSET>
__global__ __launch_bounds__(256) void k_unpack_transposed(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                           uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                           uint32_t n_var, const uint32_t *__restrict__ sel,
                                                           uint32_t n_out, int8_t *__restrict__ out,
                                                           uint64_t out_stride, int32_t fill) {
	__shared__ uint8_t tile[64][64 + 4]; // [sample][variant]
	const uint32_t k0 = blockIdx.x * 64u, j0 = blockIdx.y * 64u;
	{
		const uint32_t vr = threadIdx.x >> 2, dw = threadIdx.x & 3u;
		const uint32_t j = j0 + vr;
		if (j < n_var) {
			const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(vlist[j]) * pitch);
			uint32_t word = 0;
			if (!SUBSET && k0 + 16u * dw < sample_ct) {
				word = row32[(k0 >> 4) + dw];
			}
#pragma unroll
			for (uint32_t i = 0; i < 16; i++) {
				uint32_t code;
				if (SUBSET) {
					const uint32_t k = k0 + 16u * dw + i;
					const uint32_t s = k < n_out ? sel[k] : 0u;
					code = (row32[s >> 4] >> (2u * (s & 15u))) & 3u;
				} else {
					code = (word >> (2u * i)) & 3u;
				}
				tile[16u * dw + i][vr] = static_cast<uint8_t>(code == 3u ? fill : static_cast<int32_t>(code));
			}
		}
	}
	__syncthreads();
	const uint32_t samp = threadIdx.x >> 2, chunk = (threadIdx.x & 3u) * 16u;
	const uint32_t k = k0 + samp;
	if (k >= n_out || j0 + chunk >= n_var) {
		return;
	}
	int8_t *dst = out + static_cast<uint64_t>(k) * out_stride + j0 + chunk;
	const uint32_t left = n_var - (j0 + chunk);
	for (uint32_t i = 0; i < 16u && i < left; i++) {
		dst[i] = static_cast<int8_t>(tile[samp][chunk + i]);
	}
}

} // namespace

// A bare kernel with the unpack's traffic shape and none of its arithmetic: a lane reads 16 bytes and writes
// 4 x 16 bytes (1 KiB per wave-instruction) + 8 bytes.  What it reaches is the ceiling k_unpack_wide is measured
// against in the same run (bench.py: roofline.store_ceiling); tools/store_peak.hip holds the other store forms.
__global__ __launch_bounds__(256) void k_unpack_shape_probe(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst,
                                                            unsigned long long *__restrict__ val, size_t n_vec) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n_vec; i += stride) {
		const u32x4 w = __builtin_nontemporal_load(src + i);
		const size_t wave0 = (i & ~static_cast<size_t>(63)) * 4, lane = i & 63;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const u32x4 o = {w.x + k, w.y, w.z, w.w};
			__builtin_nontemporal_store(o, dst + wave0 + 64 * k + lane);
		}
		val[i] = w.x;
	}
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

hipError_t LaunchUnpackShapeProbe(const void *src, size_t n_vec, void *dst, void *val, hipStream_t stream) {
	if (n_vec == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_unpack_shape_probe, dim3(16384), dim3(256), 0, stream, static_cast<const u32x4 *>(src),
	                   static_cast<u32x4 *>(dst), static_cast<unsigned long long *>(val), n_vec);
	return hipGetLastError();
}

hipError_t LaunchUnpack(const RowView &view, uint32_t v_first, uint32_t v_count, int8_t *out, uint64_t out_pitch,
                        uint64_t *validity, int8_t fill, hipStream_t stream) {
	if (v_count == 0) {
		return hipSuccess;
	}
	const uint32_t val_words16 = ((view.sample_ct + 63) / 64) * 4;
	const uint32_t f = static_cast<uint8_t>(fill);
	const uint32_t fill4 = f * 0x01010101u;
	if (view.sample_ct >= 4096) {
		const uint32_t chunks = (view.sample_ct + 63) / 64;
		dim3 grid_w((chunks + 255) / 256, v_count < 65535u ? v_count : 65535u);
		static const int variant = [] {
			const char *e = getenv("PGH_UNPACK_VARIANT"); // tuning knob: 0 = non-temporal stores, 1 = plain stores
			return e ? atoi(e) : 0;
		}();
		if (variant == 1) {
			hipLaunchKernelGGL(k_unpack_wide<false>, grid_w, dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, v_first, v_count, out, out_pitch, validity, fill4);
		} else if (variant == 2) {
			dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
			hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct, v_first,
			                   v_count, out, out_pitch, validity, fill4);
		} else {
			hipLaunchKernelGGL(k_unpack_wide<true>, grid_w, dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, v_first, v_count, out, out_pitch, validity, fill4);
		}
		return hipGetLastError();
	}
	dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
	hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct, v_first, v_count,
	                   out, out_pitch, validity, fill4);
	return hipGetLastError();
}

hipError_t LaunchUnpackSubset(const RowView &view, uint32_t v_first, uint32_t v_count, const uint32_t *sel,
                              uint32_t n_out, int8_t *out, uint64_t out_pitch, uint64_t *validity, int8_t fill,
                              hipStream_t stream) {
	if (v_count == 0 || n_out == 0) {
		return hipSuccess;
	}
	const uint32_t val_words16 = ((n_out + 63) / 64) * 4;
	dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
	hipLaunchKernelGGL(k_unpack_subset, grid, dim3(256), 0, stream, view.rows, view.pitch, v_first, v_count, sel,
	                   n_out, out, out_pitch, validity, static_cast<int32_t>(fill));
	return hipGetLastError();
}

hipError_t LaunchUnpackTransposed(const RowView &view, const uint32_t *vlist, uint32_t n_var, const uint32_t *sel,
                                  uint32_t k_first, uint32_t k_count, int8_t *out, uint64_t out_stride, int8_t fill,
                                  hipStream_t stream) {
	if (n_var == 0 || k_count == 0) {
		return hipSuccess;
	}
	// samples k_first .. k_first + k_count of the output order; k_first is a multiple of 64
	for (uint32_t j_done = 0; j_done < n_var; j_done += 65535u * 64u) { // grid.y limit
		const uint32_t n = std::min<uint64_t>(65535ull * 64ull, n_var - j_done);
		dim3 grid((k_count + 63) / 64, (n + 63) / 64);
		if (sel) {
			hipLaunchKernelGGL(k_unpack_transposed<true>, grid, dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct,
			                   vlist + j_done, n, sel + k_first, k_count, out + j_done, out_stride, static_cast<int32_t>(fill));
		} else {
			RowView shifted = view;
			shifted.rows = view.rows + (k_first / 4);
			hipLaunchKernelGGL(k_unpack_transposed<false>, grid, dim3(256), 0, stream, shifted.rows, view.pitch,
			                   view.sample_ct - k_first, vlist + j_done, n, nullptr, k_count, out + j_done, out_stride,
			                   static_cast<int32_t>(fill));
		}
	}
	return hipGetLastError();
}

} // namespace pgh
