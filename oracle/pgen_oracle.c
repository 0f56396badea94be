/*
 * oracle/pgen_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one variant at a time) of the decode and
 * tally arithmetic that the reference obtains from plink-ng's pgenlib
 * (github.com/chrchang/plink-ng @ 4ce97faa08bc370bedb30dcc82b4eeeef1c7c1f4,
 * 2.0/include/pgenlib_read.cc, plink2_stats.cc).  That third-party source is an
 * empty submodule in /root/reference, so the functions below restate the
 * published PLINK 2 .pgen format rules and are anchored on the reference's own
 * call sites:
 *
 *   pgo_get_geno        <-> PgrGet + GenoarrToBytesMinus9   src/pgen_reader.cpp:727-733,
 *                                                           src/plink_freq.cpp:463-469
 *   pgo_get_counts      <-> PgrGetCounts                    src/plink_freq.cpp:482
 *   pgo_get_missingness <-> PgrGetMissingness               src/plink_missing.cpp:479
 *   pgo_get_dosage      <-> PgrGetD + Dosage16ToDoublesMinus9  src/plink_score.cpp:586-596
 *   pgo_get_dcounts     <-> PgrGetDCounts                   src/plink_freq.cpp:475
 *   pgo_get_phase       <-> PgrGetP                         src/pgen_reader.cpp:715
 *   pgo_hwe_lnp         <-> plink2::HweLnP                  src/plink_hardy.cpp:78
 *   pgo_hwe_xchr_lnp    <-> plink2::HweXchrLnP              src/plink_hardy.cpp:94
 *   pgo_scan_counts_mt  <-> the scan-thread structure       src/plink_freq.cpp:434-488
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here
 * against the known-answer values of the reference's own sqllogictests
 *  over the reference's committed fixture files.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (plinking_duck_amd/) never does.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	uint8_t *body;     /* bytes of the .pgen file */
	size_t body_len;
	uint8_t *index;    /* bytes holding the header tables (== body for mode 0x10/0x02, .pgi for 0x20) */
	size_t index_len;
	int owns_body;
	uint32_t M, N;
	uint8_t mode, ctrl;
	uint8_t *vrtype;   /* M entries, 8-bit (4-bit entries widened) */
	uint64_t *fpos;    /* M+1 record offsets into body */
	uint32_t sid_bytes;
	int has_dosage, has_phase, has_multiallelic;
	uint32_t *allele_ct; /* REF + ALTs per variant when the header carries ALT allele counts, else NULL */
	char err[256];
} pgo_file;

static char g_open_err[256];

const char *pgo_last_open_error(void) {
	return g_open_err;
}

static uint8_t *read_whole(const char *path, size_t *len_out) {
	FILE *f = fopen(path, "rb");
	if (!f) {
		return NULL;
	}
	fseeko(f, 0, SEEK_END);
	off_t n = ftello(f);
	fseeko(f, 0, SEEK_SET);
	uint8_t *buf = (uint8_t *)malloc((size_t)n + 16);
	if (!buf) {
		fclose(f);
		return NULL;
	}
	if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) {
		free(buf);
		fclose(f);
		return NULL;
	}
	memset(buf + n, 0, 16);
	fclose(f);
	*len_out = (size_t)n;
	return buf;
}

static uint32_t rd_u32(const uint8_t *p) {
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd_u64(const uint8_t *p) {
	return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32);
}

void pgo_close(pgo_file *h) {
	if (!h) {
		return;
	}
	if (h->index != h->body) {
		free(h->index);
	}
	if (h->owns_body) {
		free(h->body);
	}
	free(h->vrtype);
	free(h->fpos);
	free(h->allele_ct);
	free(h);
}

/* Parse the header + per-variant tables.  Layout (PLINK 2 .pgen spec):
 *   6c 1b <mode> u32 M, u32 N, ctrl byte,
 *   u64 body offset per 65536-variant block,
 *   then per block: vrtypes (4- or 8-bit), record byte lengths (1-4 B),
 *   [alt allele counts], [nonref flag bits]. */
static uint32_t rd_sid(const uint8_t *p, uint32_t nbytes);

static pgo_file *pgo_parse(uint8_t *body, size_t body_len, uint8_t *index, size_t index_len, int owns_body) {
	pgo_file *h = (pgo_file *)calloc(1, sizeof(pgo_file));
	h->body = body;
	h->body_len = body_len;
	h->index = index;
	h->index_len = index_len;
	h->owns_body = owns_body;
	if (index_len < 3 || index[0] != 0x6c || index[1] != 0x1b) {
		snprintf(g_open_err, sizeof g_open_err, "not a .pgen file (bad magic)");
		goto fail;
	}
	h->mode = index[2];
	if (h->mode == 0x02) {
		/* fixed-width 2-bit: 12-byte header, every record ceil(N/4) bytes, vrtype 0 */
		if (index_len < 12) {
			snprintf(g_open_err, sizeof g_open_err, "truncated fixed-width header");
			goto fail;
		}
		h->M = rd_u32(index + 3);
		h->N = rd_u32(index + 7);
		h->ctrl = index[11];
		h->vrtype = (uint8_t *)calloc(h->M ? h->M : 1, 1);
		h->fpos = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)h->M + 1));
		uint64_t w = ((uint64_t)h->N + 3) / 4;
		for (uint64_t v = 0; v <= h->M; v++) {
			h->fpos[v] = 12 + v * w;
		}
	} else if (h->mode == 0x10 || h->mode == 0x30) {
		if (index_len < 12) {
			snprintf(g_open_err, sizeof g_open_err, "truncated header");
			goto fail;
		}
		h->M = rd_u32(index + 3);
		h->N = rd_u32(index + 7);
		h->ctrl = index[11];
		uint32_t vr_bits = (h->ctrl & 0x0f) < 4 ? 4 : 8;
		uint32_t len_bytes = (h->ctrl & 3) + 1;
		if ((h->ctrl & 0x0f) >= 8) {
			snprintf(g_open_err, sizeof g_open_err, "unsupported header ctrl 0x%02x", h->ctrl);
			goto fail;
		}
		uint32_t ac_bytes = (h->ctrl >> 4) & 3;
		uint32_t nonref_mode = (h->ctrl >> 6) & 3;
		uint32_t block_ct = (h->M + 65535) / 65536;
		size_t pos = 12;
		size_t offs_pos = pos;
		pos += (size_t)block_ct * 8;
		h->vrtype = (uint8_t *)calloc(h->M ? h->M : 1, 1);
		h->fpos = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)h->M + 1));
		for (uint32_t b = 0; b < block_ct; b++) {
			uint32_t v0 = b * 65536u;
			uint32_t cnt = h->M - v0 < 65536u ? h->M - v0 : 65536u;
			uint64_t fp = rd_u64(index + offs_pos + 8 * (size_t)b);
			size_t vr_len = vr_bits == 4 ? (cnt + 1) / 2 : cnt;
			if (pos + vr_len + (size_t)cnt * len_bytes > index_len) {
				snprintf(g_open_err, sizeof g_open_err, "truncated variant-record tables");
				goto fail;
			}
			for (uint32_t i = 0; i < cnt; i++) {
				uint8_t t;
				if (vr_bits == 4) {
					t = (index[pos + i / 2] >> (4 * (i & 1))) & 0x0f;
				} else {
					t = index[pos + i];
				}
				h->vrtype[v0 + i] = t;
			}
			pos += vr_len;
			for (uint32_t i = 0; i < cnt; i++) {
				uint32_t l = 0;
				for (uint32_t k = 0; k < len_bytes; k++) {
					l |= (uint32_t)index[pos + (size_t)i * len_bytes + k] << (8 * k);
				}
				h->fpos[v0 + i] = fp;
				fp += l;
			}
			h->fpos[v0 + cnt] = fp;
			pos += (size_t)cnt * len_bytes;
			if (ac_bytes) { /* ALT allele counts */
				if (!h->allele_ct) {
					h->allele_ct = (uint32_t *)malloc(sizeof(uint32_t) * (h->M ? h->M : 1));
					for (uint32_t i = 0; i < h->M; i++) {
						h->allele_ct[i] = 2;
					}
				}
				for (uint32_t i = 0; i < cnt && pos + (size_t)(i + 1) * ac_bytes <= index_len; i++) {
					h->allele_ct[v0 + i] = 1 + rd_sid(index + pos + (size_t)i * ac_bytes, ac_bytes);
				}
			}
			pos += (size_t)cnt * ac_bytes;
			if (nonref_mode == 3) {
				pos += (cnt + 7) / 8;
			}
		}
		if (h->M == 0) {
			h->fpos[0] = 0;
		}
	} else {
		snprintf(g_open_err, sizeof g_open_err, "unsupported storage mode 0x%02x", h->mode);
		goto fail;
	}
	if (h->M && h->fpos[h->M] > body_len) {
		snprintf(g_open_err, sizeof g_open_err, "variant records run past end of file");
		goto fail;
	}
	/* sample-id width in difflists = bytes needed to represent N itself */
	h->sid_bytes = h->N < 0x100 ? 1 : (h->N < 0x10000 ? 2 : (h->N < 0x1000000 ? 3 : 4));
	for (uint32_t v = 0; v < h->M; v++) {
		uint8_t t = h->vrtype[v];
		if (t & 0x60) {
			h->has_dosage = 1;
		}
		if (t & 0x10) {
			h->has_phase = 1;
		}
		if (t & 0x08) {
			h->has_multiallelic = 1;
		}
	}
	return h;
fail:
	h->owns_body = 0; /* the caller keeps ownership of both buffers on failure */
	h->index = h->body;
	pgo_close(h);
	return NULL;
}

pgo_file *pgo_open(const char *path) {
	size_t len = 0;
	uint8_t *buf = read_whole(path, &len);
	if (!buf) {
		snprintf(g_open_err, sizeof g_open_err, "cannot read '%.200s'", path);
		return NULL;
	}
	if (len >= 3 && buf[0] == 0x6c && buf[1] == 0x1b && buf[2] == 0x20) {
		/* body-only .pgen with external index <path>.pgi */
		char pgi[4096];
		snprintf(pgi, sizeof pgi, "%s.pgi", path);
		size_t ilen = 0;
		uint8_t *ibuf = read_whole(pgi, &ilen);
		if (!ibuf) {
			snprintf(g_open_err, sizeof g_open_err, "cannot read index '%.200s'", pgi);
			free(buf);
			return NULL;
		}
		pgo_file *h = pgo_parse(buf, len, ibuf, ilen, 1);
		if (!h) {
			free(ibuf);
			free(buf);
		}
		return h;
	}
	pgo_file *h = pgo_parse(buf, len, buf, len, 1);
	if (!h) {
		free(buf);
	}
	return h;
}

/* In-memory open over caller-owned bytes (bench cpu_baseline: synthetic records
 * generated in RAM). */
pgo_file *pgo_open_mem(uint8_t *bytes, size_t len) {
	return pgo_parse(bytes, len, bytes, len, 0);
}

uint32_t pgo_variant_ct(const pgo_file *h) {
	return h->M;
}
uint32_t pgo_sample_ct(const pgo_file *h) {
	return h->N;
}
int pgo_has_dosage(const pgo_file *h) {
	return h->has_dosage;
}
int pgo_has_phase(const pgo_file *h) {
	return h->has_phase;
}
uint32_t pgo_vrtype(const pgo_file *h, uint32_t v) {
	return h->vrtype[v];
}
uint64_t pgo_record_offset(const pgo_file *h, uint32_t v) {
	return h->fpos[v];
}

/* ------------------------------------------------------------------------- */
/* Record decode                                                             */
/* ------------------------------------------------------------------------- */

static uint32_t rd_varint(const uint8_t **pp, const uint8_t *end) {
	uint32_t v = 0, shift = 0;
	const uint8_t *p = *pp;
	while (p < end) {
		uint8_t b = *p++;
		v |= (uint32_t)(b & 0x7f) << shift;
		if (!(b & 0x80)) {
			break;
		}
		shift += 7;
	}
	*pp = p;
	return v;
}

static uint32_t rd_sid(const uint8_t *p, uint32_t nbytes) {
	uint32_t v = 0;
	for (uint32_t k = 0; k < nbytes; k++) {
		v |= (uint32_t)p[k] << (8 * k);
	}
	return v;
}

/* Apply a difflist at *pp to the per-sample genotype byte array g[N]
 * (values 0..3).  with_geno = 0 parses an id-only list into ids_out. */
static int apply_difflist(const pgo_file *h, const uint8_t **pp, const uint8_t *end, uint8_t *g, uint32_t *ids_out,
                          uint32_t *len_out) {
	const uint8_t *p = *pp;
	uint32_t len = rd_varint(&p, end);
	if (len_out) {
		*len_out = len;
	}
	if (len == 0) {
		*pp = p;
		return 0;
	}
	if (len > h->N) {
		return -1;
	}
	uint32_t group_ct = (len + 63) / 64;
	const uint8_t *first_ids = p;
	p += (size_t)group_ct * h->sid_bytes;
	p += group_ct - 1; /* per-group delta byte lengths: only needed for random access */
	const uint8_t *vals = NULL;
	if (g) {
		vals = p;
		p += (len + 3) / 4;
	}
	if (p > end) {
		return -1;
	}
	uint32_t k = 0;
	for (uint32_t grp = 0; grp < group_ct; grp++) {
		uint32_t sid = rd_sid(first_ids + (size_t)grp * h->sid_bytes, h->sid_bytes);
		uint32_t in_grp = len - grp * 64 < 64 ? len - grp * 64 : 64;
		for (uint32_t j = 0; j < in_grp; j++, k++) {
			if (j) {
				sid += rd_varint(&p, end);
			}
			if (sid >= h->N) {
				return -1;
			}
			if (g) {
				g[sid] = (vals[k / 4] >> (2 * (k & 3))) & 3;
			}
			if (ids_out) {
				ids_out[k] = sid;
			}
		}
	}
	*pp = p;
	return 0;
}

/* Decode the main (hardcall) track of variant v into g[N] (0,1,2,3=missing),
 * returning a pointer just past it (start of aux tracks) in *aux_out. */
static int decode_main(const pgo_file *h, uint32_t v, uint8_t *g, const uint8_t **aux_out) {
	const uint32_t N = h->N;
	const uint8_t t = h->vrtype[v];
	const uint8_t *p = h->body + h->fpos[v];
	const uint8_t *end = h->body + h->fpos[v + 1];
	switch (t & 7) {
	case 0: {
		size_t nb = ((size_t)N + 3) / 4;
		if (p + nb > end) {
			return -1;
		}
		for (uint32_t s = 0; s < N; s++) {
			g[s] = (p[s / 4] >> (2 * (s & 3))) & 3;
		}
		p += nb;
		break;
	}
	case 1: {
		/* 1-bit: byte c names the two values (low = c/4, high = low + (c&3)) */
		uint8_t c = *p++;
		uint8_t lo = c >> 2, hi = (uint8_t)(lo + (c & 3));
		size_t nb = ((size_t)N + 7) / 8;
		if (p + nb > end) {
			return -1;
		}
		for (uint32_t s = 0; s < N; s++) {
			g[s] = ((p[s / 8] >> (s & 7)) & 1) ? hi : lo;
		}
		p += nb;
		if (apply_difflist(h, &p, end, g, NULL, NULL)) {
			return -1;
		}
		break;
	}
	case 2:
	case 3: {
		/* LD-compressed against the most recent non-LD record, then a difflist;
		 * type 3 additionally swaps 0 <-> 2 AFTER the difflist is applied. */
		uint32_t b = v;
		while (b > 0) {
			b--;
			uint32_t bt = h->vrtype[b] & 7;
			if (bt != 2 && bt != 3) {
				break;
			}
		}
		if ((h->vrtype[b] & 6) == 2) {
			return -1; /* no base */
		}
		const uint8_t *dummy;
		if (decode_main(h, b, g, &dummy)) {
			return -1;
		}
		if (apply_difflist(h, &p, end, g, NULL, NULL)) {
			return -1;
		}
		if ((t & 7) == 3) {
			for (uint32_t s = 0; s < N; s++) {
				if (g[s] == 0) {
					g[s] = 2;
				} else if (g[s] == 2) {
					g[s] = 0;
				}
			}
		}
		break;
	}
	case 4:
	case 6:
	case 7: {
		uint8_t base = (t & 7) == 4 ? 0 : ((t & 7) == 6 ? 2 : 3);
		memset(g, base, N);
		if (apply_difflist(h, &p, end, g, NULL, NULL)) {
			return -1;
		}
		break;
	}
	default:
		return -1;
	}
	if (aux_out) {
		*aux_out = p;
	}
	return 0;
}

static uint8_t *scratch_g(const pgo_file *h) {
	return (uint8_t *)malloc(h->N ? h->N : 1);
}

/* hardcalls as int8 {0,1,2,-9}, compacted to the included samples in ascending
 * file order (pgenlib subset semantics).  include: N bytes (0/1) or NULL. */
int pgo_get_geno(const pgo_file *h, uint32_t v, const uint8_t *include, int8_t *out) {
	if (v >= h->M) {
		return -1;
	}
	uint8_t *g = scratch_g(h);
	int rc = decode_main(h, v, g, NULL);
	if (!rc) {
		uint32_t k = 0;
		for (uint32_t s = 0; s < h->N; s++) {
			if (include && !include[s]) {
				continue;
			}
			out[k++] = g[s] == 3 ? -9 : (int8_t)g[s];
		}
		rc = (int)k;
	}
	free(g);
	return rc;
}

/* plink_ld's sample loop (src/plink_ld.cpp:52-84): over the included samples at which
 * neither call is missing, out = {n, sum_a, sum_b, sum_ab, sum_a2, sum_b2}.  The reference
 * accumulates these in doubles; every one is an integer below 2^53, so uint64 is the same. */
int pgo_ld_sums(const pgo_file *h, uint32_t va, uint32_t vb, const uint8_t *include, uint64_t out[6]) {
	if (va >= h->M || vb >= h->M) {
		return -1;
	}
	uint8_t *ga = scratch_g(h);
	uint8_t *gb = scratch_g(h);
	int rc = decode_main(h, va, ga, NULL);
	if (!rc) {
		rc = decode_main(h, vb, gb, NULL);
	}
	if (!rc) {
		memset(out, 0, 6 * sizeof(uint64_t));
		for (uint32_t s = 0; s < h->N; s++) {
			if ((include && !include[s]) || ga[s] == 3 || gb[s] == 3) {
				continue;
			}
			uint64_t a = ga[s], b = gb[s];
			out[0]++;
			out[1] += a;
			out[2] += b;
			out[3] += a * b;
			out[4] += a * a;
			out[5] += b * b;
		}
	}
	free(ga);
	free(gb);
	return rc;
}

/* raw 2-bit codes (0..3), one byte per raw sample, no subsetting */
int pgo_get_raw(const pgo_file *h, uint32_t v, uint8_t *out) {
	if (v >= h->M) {
		return -1;
	}
	return decode_main(h, v, out, NULL);
}

int pgo_get_counts(const pgo_file *h, uint32_t v, const uint8_t *include, uint32_t out[4]) {
	if (v >= h->M) {
		return -1;
	}
	uint8_t *g = scratch_g(h);
	int rc = decode_main(h, v, g, NULL);
	out[0] = out[1] = out[2] = out[3] = 0;
	if (!rc) {
		for (uint32_t s = 0; s < h->N; s++) {
			if (include && !include[s]) {
				continue;
			}
			out[g[s]]++;
		}
	}
	free(g);
	return rc;
}

/* one byte per included sample: 1 if hardcall missing */
int pgo_get_missingness(const pgo_file *h, uint32_t v, const uint8_t *include, uint8_t *out) {
	if (v >= h->M) {
		return -1;
	}
	uint8_t *g = scratch_g(h);
	int rc = decode_main(h, v, g, NULL);
	if (!rc) {
		uint32_t k = 0;
		for (uint32_t s = 0; s < h->N; s++) {
			if (include && !include[s]) {
				continue;
			}
			out[k++] = g[s] == 3;
		}
		rc = (int)k;
	}
	free(g);
	return rc;
}

/* Aux track 1 (vrtype bit 0x08, multiallelic patches) sits between the main track and the phase / dosage
 * tracks.  PgrGet / PgrGetCounts / PgrGetD collapse the ALT alleles (the main track as stored), so it is only
 * measured here.  Layout from the PLINK 2 .pgen specification -- NO reference fixture holds such a record, so
 * this is parity unpinned: 1 byte of modes (low nibble part a: patches of genotype-1 calls, high nibble part b:
 * patches of genotype-2 calls; 0 = a bit per such call, 1 = id list in difflist layout, 15 = none); part a's
 * patched calls carry one allele code each (0 / 1 / 2 / 4 / 8 bits for 3 / 4 / 5-6 / 7-18 / more alleles), part
 * b's two (1 bit per call for 3 alleles, then 2+2 / 4+4 / 8+8 bits for 4-5 / 6-17 / more).  Returns NULL when
 * malformed. */
static const uint8_t *skip_aux1(const pgo_file *h, uint32_t v, const uint8_t *g, const uint8_t *p, const uint8_t *end) {
	uint32_t alleles = h->allele_ct ? h->allele_ct[v] : 2;
	if (alleles < 3 || p >= end) {
		return NULL;
	}
	uint32_t calls[2] = {0, 0};
	for (uint32_t s = 0; s < h->N; s++) {
		calls[0] += g[s] == 1;
		calls[1] += g[s] == 2;
	}
	uint8_t modes = *p++;
	for (int part = 0; part < 2; part++) {
		uint32_t mode = part == 0 ? (modes & 15u) : (modes >> 4), n = 0;
		if (mode == 0) {
			uint32_t nb = (calls[part] + 7) / 8;
			if (p + nb > end) {
				return NULL;
			}
			for (uint32_t i = 0; i < calls[part]; i++) {
				n += (p[i / 8] >> (i & 7)) & 1;
			}
			p += nb;
		} else if (mode == 1) {
			uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (h->N ? h->N : 1));
			int rc = apply_difflist(h, &p, end, NULL, ids, &n);
			free(ids);
			if (rc) {
				return NULL;
			}
		} else if (mode != 15) {
			return NULL;
		}
		uint32_t bits;
		if (part == 0) {
			bits = alleles == 3 ? 0 : alleles == 4 ? 1 : alleles <= 6 ? 2 : alleles <= 18 ? 4 : 8;
		} else {
			bits = alleles == 3 ? 1 : alleles <= 5 ? 4 : alleles <= 17 ? 8 : 16;
		}
		p += ((uint64_t)n * bits + 7) / 8;
		if (p > end) {
			return NULL;
		}
	}
	return p;
}

/* Phase track (vrtype bit 0x10).  Outputs per included sample. */
int pgo_get_phase(const pgo_file *h, uint32_t v, const uint8_t *include, int8_t *geno_out, uint8_t *phasepresent_out,
                  uint8_t *phaseinfo_out) {
	if (v >= h->M) {
		return -1;
	}
	const uint32_t N = h->N;
	uint8_t *g = scratch_g(h);
	const uint8_t *p;
	int rc = decode_main(h, v, g, &p);
	if (rc) {
		free(g);
		return rc;
	}
	uint8_t t = h->vrtype[v];
	if (t & 0x08) {
		p = skip_aux1(h, v, g, p, h->body + h->fpos[v + 1]);
		if (!p) {
			free(g);
			return -2;
		}
	}
	uint8_t *pp = (uint8_t *)calloc(N ? N : 1, 1);
	uint8_t *pi = (uint8_t *)calloc(N ? N : 1, 1);
	if (t & 0x10) {
		uint32_t het_ct = 0;
		for (uint32_t s = 0; s < N; s++) {
			het_ct += g[s] == 1;
		}
		const uint8_t *first = p;
		uint32_t first_bytes = (1 + het_ct + 7) / 8;
		int explicit_pp = first[0] & 1;
		const uint8_t *info = first + first_bytes;
		uint32_t hi = 0, phased_i = 0;
		for (uint32_t s = 0; s < N; s++) {
			if (g[s] != 1) {
				continue;
			}
			uint32_t bit = 1 + hi;
			int b = (first[bit / 8] >> (bit & 7)) & 1;
			if (!explicit_pp) {
				pp[s] = 1;
				pi[s] = (uint8_t)b;
			} else if (b) {
				pp[s] = 1;
				pi[s] = (info[phased_i / 8] >> (phased_i & 7)) & 1;
				phased_i++;
			}
			hi++;
		}
	}
	uint32_t k = 0;
	for (uint32_t s = 0; s < N; s++) {
		if (include && !include[s]) {
			continue;
		}
		geno_out[k] = g[s] == 3 ? -9 : (int8_t)g[s];
		phasepresent_out[k] = pp[s];
		phaseinfo_out[k] = pi[s];
		k++;
	}
	free(pp);
	free(pi);
	free(g);
	return (int)k;
}

/* Locate and decode the dosage track: dos[s] = u16 dosage or 0xffff if the
 * sample has no explicit dosage. */
static int decode_dosage16(const pgo_file *h, uint32_t v, uint8_t *g, uint16_t *dos) {
	const uint32_t N = h->N;
	const uint8_t *p;
	if (decode_main(h, v, g, &p)) {
		return -1;
	}
	const uint8_t *end = h->body + h->fpos[v + 1];
	uint8_t t = h->vrtype[v];
	for (uint32_t s = 0; s < N; s++) {
		dos[s] = 0xffff;
	}
	if (t & 0x08) {
		p = skip_aux1(h, v, g, p, end);
		if (!p) {
			return -2;
		}
	}
	/* (a phased-dosage track, 0x80, lies behind the dosage track: PgrGetD does not read it, nor does this) */
	if (t & 0x10) {
		uint32_t het_ct = 0;
		for (uint32_t s = 0; s < N; s++) {
			het_ct += g[s] == 1;
		}
		uint32_t first_bytes = (1 + het_ct + 7) / 8;
		if (p[0] & 1) {
			uint32_t phased = 0;
			for (uint32_t i = 0; i < het_ct; i++) {
				phased += (p[(1 + i) / 8] >> ((1 + i) & 7)) & 1;
			}
			p += first_bytes + (phased + 7) / 8;
		} else {
			p += first_bytes;
		}
	}
	switch (t & 0x60) {
	case 0:
		break;
	case 0x20: {
		uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
		uint32_t len = 0;
		if (apply_difflist(h, &p, end, NULL, ids, &len)) {
			free(ids);
			return -1;
		}
		for (uint32_t i = 0; i < len; i++) {
			dos[ids[i]] = (uint16_t)(p[2 * i] | (p[2 * i + 1] << 8));
		}
		free(ids);
		break;
	}
	case 0x40:
		for (uint32_t s = 0; s < N; s++) {
			dos[s] = (uint16_t)(p[2 * s] | (p[2 * s + 1] << 8));
		}
		break;
	case 0x60: {
		const uint8_t *bits = p;
		p += (N + 7) / 8;
		uint32_t k = 0;
		for (uint32_t s = 0; s < N; s++) {
			if ((bits[s / 8] >> (s & 7)) & 1) {
				dos[s] = (uint16_t)(p[2 * k] | (p[2 * k + 1] << 8));
				k++;
			}
		}
		break;
	}
	}
	return 0;
}

/* ALT dosage in [0,2] as double, -9.0 when missing; explicit dosage wins over
 * the hardcall, u16/16384 (Dosage16ToDoublesMinus9). */
int pgo_get_dosage(const pgo_file *h, uint32_t v, const uint8_t *include, double *out) {
	if (v >= h->M) {
		return -1;
	}
	uint8_t *g = scratch_g(h);
	uint16_t *dos = (uint16_t *)malloc(sizeof(uint16_t) * (h->N ? h->N : 1));
	int rc = decode_dosage16(h, v, g, dos);
	if (!rc) {
		uint32_t k = 0;
		for (uint32_t s = 0; s < h->N; s++) {
			if (include && !include[s]) {
				continue;
			}
			if (dos[s] != 0xffff) {
				out[k++] = (double)dos[s] / 16384.0;
			} else {
				out[k++] = g[s] == 3 ? -9.0 : (double)g[s];
			}
		}
		rc = (int)k;
	}
	free(dos);
	free(g);
	return rc;
}

/* PgrGetDCounts: hardcall counts, dosage sums scaled by 16384 per allele, and
 * the MaCH imputation r2 (variance of dosage / 2p(1-p)). */
int pgo_get_dcounts(const pgo_file *h, uint32_t v, const uint8_t *include, uint32_t counts[4], uint64_t all_dosages[2],
                    double *imp_r2) {
	if (v >= h->M) {
		return -1;
	}
	uint8_t *g = scratch_g(h);
	uint16_t *dos = (uint16_t *)malloc(sizeof(uint16_t) * (h->N ? h->N : 1));
	int rc = decode_dosage16(h, v, g, dos);
	counts[0] = counts[1] = counts[2] = counts[3] = 0;
	all_dosages[0] = all_dosages[1] = 0;
	*imp_r2 = 0.0;
	if (!rc) {
		uint64_t sum = 0, ssq = 0;
		uint32_t nm = 0;
		for (uint32_t s = 0; s < h->N; s++) {
			if (include && !include[s]) {
				continue;
			}
			counts[g[s]]++;
			uint64_t d;
			if (dos[s] != 0xffff) {
				d = dos[s];
			} else if (g[s] != 3) {
				d = (uint64_t)g[s] * 16384u;
			} else {
				continue;
			}
			nm++;
			sum += d;
			ssq += d * d;
		}
		all_dosages[1] = sum;
		all_dosages[0] = (uint64_t)nm * 32768u - sum;
		if (nm) {
			double sumd = (double)sum;
			double avg = sumd / (double)nm;
			double var = (double)ssq - sumd * avg;
			double denom = sumd * (32768.0 - avg);
			*imp_r2 = denom != 0.0 ? 2.0 * var / denom : NAN;
		}
	}
	free(dos);
	free(g);
	return rc;
}

/* ------------------------------------------------------------------------- */
/* Range helpers used by the tests (loops over the per-variant functions)     */
/* ------------------------------------------------------------------------- */

int pgo_counts_range(const pgo_file *h, uint32_t v0, uint32_t v1, const uint8_t *include, uint32_t *out) {
	for (uint32_t v = v0; v < v1; v++) {
		int rc = pgo_get_counts(h, v, include, out + 4 * (size_t)(v - v0));
		if (rc) {
			return rc;
		}
	}
	return 0;
}

/* per-sample missing tallies over [v0,v1): plink_missing.cpp:585-619 */
int pgo_missing_per_sample(const pgo_file *h, uint32_t v0, uint32_t v1, const uint8_t *include, uint32_t *out) {
	uint32_t n_out = 0;
	for (uint32_t s = 0; s < h->N; s++) {
		n_out += !include || include[s];
	}
	memset(out, 0, sizeof(uint32_t) * n_out);
	uint8_t *m = (uint8_t *)malloc(h->N ? h->N : 1);
	for (uint32_t v = v0; v < v1; v++) {
		int rc = pgo_get_missingness(h, v, include, m);
		if (rc < 0) {
			free(m);
			return rc;
		}
		for (uint32_t k = 0; k < n_out; k++) {
			out[k] += m[k];
		}
	}
	free(m);
	return 0;
}

/* ------------------------------------------------------------------------- */
/* HWE exact tests                                                           */
/* ------------------------------------------------------------------------- */

/* Autosomal exact test (Wigginton, Cutler, Abecasis 2005), two-sided: sum of
 * the probabilities of all het counts no likelier than the observed one;
 * mid-p subtracts half the probability of the tables tied with the observed.
 * Every table is evaluated directly in log space with lgamma (slow, O(range)
 * lgamma calls, but with no recurrence to get wrong):
 *   P(k hets) = 2^k n! nA! nB! / (hA! k! hB! (2n)!),  hA = (nA-k)/2, hB = (nB-k)/2 */
static double hwe_lnprob(int64_t n, int64_t nA, int64_t nB, int64_t k) {
	int64_t hA = (nA - k) / 2, hB = (nB - k) / 2;
	return (double)k * M_LN2 + lgamma((double)n + 1) + lgamma((double)nA + 1) + lgamma((double)nB + 1) -
	       lgamma((double)hA + 1) - lgamma((double)k + 1) - lgamma((double)hB + 1) - lgamma(2.0 * (double)n + 1);
}

double pgo_hwe_lnp(int32_t obs_hets, int32_t obs_hom1, int32_t obs_hom2, uint32_t midp) {
	int64_t n = (int64_t)obs_hets + obs_hom1 + obs_hom2;
	if (n == 0) {
		return 0.0;
	}
	int64_t nA = 2 * (int64_t)obs_hom1 + obs_hets;
	int64_t nB = 2 * (int64_t)obs_hom2 + obs_hets;
	int64_t rare = nA < nB ? nA : nB;
	double ln_obs = hwe_lnprob(n, nA, nB, obs_hets);
	double ln_max = ln_obs;
	for (int64_t k = rare & 1; k <= rare; k += 2) {
		double lp = hwe_lnprob(n, nA, nB, k);
		if (lp > ln_max) {
			ln_max = lp;
		}
	}
	/* probabilities relative to the modal table, so nothing overflows */
	double total = 0.0, tail = 0.0, ties = 0.0;
	for (int64_t k = rare & 1; k <= rare; k += 2) {
		double lp = hwe_lnprob(n, nA, nB, k);
		double rel = exp(lp - ln_max);
		total += rel;
		if (lp <= ln_obs + 1e-9) {
			tail += rel;
			if (lp >= ln_obs - 1e-9) {
				ties += rel;
			}
		}
	}
	if (midp) {
		tail -= 0.5 * ties;
	}
	double pv = tail / total;
	if (pv > 1.0) {
		pv = 1.0;
	}
	return log(pv);
}

/* chrX exact test (Graffelman & Weir 2016): joint distribution of the number
 * of A-allele males and female heterozygotes given the allele and sex totals.
 *   P(mA, fAB) = nA! nB! nm! nf! 2^fAB / (mA! mB! fAA! fAB! fBB! nt!)
 * Evaluated with lgamma; the sum runs over every table (oracle sizes only). */
static double xchr_lnprob(int64_t nA, int64_t nB, int64_t nm, int64_t nf, int64_t mA, int64_t fAB) {
	int64_t mB = nm - mA;
	int64_t fA = nA - mA; /* A alleles among females */
	int64_t fAA = (fA - fAB) / 2;
	int64_t fBB = nf - fAA - fAB;
	int64_t nt = nA + nB;
	return lgamma((double)nA + 1) + lgamma((double)nB + 1) + lgamma((double)nm + 1) + lgamma((double)nf + 1) +
	       (double)fAB * M_LN2 - lgamma((double)mA + 1) - lgamma((double)mB + 1) - lgamma((double)fAA + 1) -
	       lgamma((double)fAB + 1) - lgamma((double)fBB + 1) - lgamma((double)nt + 1);
}

double pgo_hwe_xchr_lnp(int32_t female_hets, int32_t female_hom1, int32_t female_hom2, int32_t male1, int32_t male2,
                        uint32_t midp) {
	int64_t nf = (int64_t)female_hets + female_hom1 + female_hom2;
	int64_t nm = (int64_t)male1 + male2;
	if (nf + nm == 0) {
		return 0.0;
	}
	int64_t nA = 2 * (int64_t)female_hom1 + female_hets + male1;
	int64_t nB = 2 * (int64_t)female_hom2 + female_hets + male2;
	double ln_obs = xchr_lnprob(nA, nB, nm, nf, male1, female_hets);
	double total = 0.0, tail = 0.0, ties = 0.0;
	int64_t mA_lo = nA - 2 * nf > 0 ? nA - 2 * nf : 0;
	int64_t mA_hi = nA < nm ? nA : nm;
	for (int64_t mA = mA_lo; mA <= mA_hi; mA++) {
		int64_t fA = nA - mA;
		int64_t fB = 2 * nf - fA;
		if (fB < 0) {
			continue;
		}
		int64_t max_het = fA < fB ? fA : fB;
		for (int64_t fAB = fA & 1; fAB <= max_het; fAB += 2) {
			double lp = xchr_lnprob(nA, nB, nm, nf, mA, fAB);
			double rel = exp(lp - ln_obs);
			total += rel;
			if (rel <= 1.0 + 1e-9) {
				tail += rel;
				if (rel >= 1.0 - 1e-9) {
					ties += rel;
				}
			}
		}
	}
	if (midp) {
		tail -= 0.5 * ties;
	}
	double pv = tail / total;
	if (pv > 1.0) {
		pv = 1.0;
	}
	return log(pv);
}

/* ------------------------------------------------------------------------- */
/* Multi-threaded CPU scan with the reference's structure (cpu_baseline)     */
/* ------------------------------------------------------------------------- */

typedef struct {
	const pgo_file *h;
	uint32_t v_end;
	uint32_t *next; /* shared atomic cursor */
	uint32_t *out;  /* [v_end - v_begin][4] */
	uint32_t v_begin;
} scan_arg;

/* PgrGetCounts on a plain 2-bit record: 64-bit-word popcount tally. */
static void counts_words(const uint8_t *rec, uint32_t N, uint32_t out[4]) {
	const uint64_t m5 = 0x5555555555555555ull;
	uint64_t lo_ct = 0, hi_ct = 0, both_ct = 0;
	uint32_t full = N / 32;
	for (uint32_t w = 0; w < full; w++) {
		uint64_t x;
		memcpy(&x, rec + 8 * (size_t)w, 8);
		uint64_t lo = x & m5, hi = (x >> 1) & m5;
		lo_ct += (uint64_t)__builtin_popcountll(lo);
		hi_ct += (uint64_t)__builtin_popcountll(hi);
		both_ct += (uint64_t)__builtin_popcountll(lo & hi);
	}
	uint32_t rem = N - full * 32;
	if (rem) {
		uint64_t x = 0;
		memcpy(&x, rec + 8 * (size_t)full, (rem + 3) / 4);
		x &= rem == 32 ? ~0ull : ((1ull << (2 * rem)) - 1);
		uint64_t lo = x & m5, hi = (x >> 1) & m5;
		lo_ct += (uint64_t)__builtin_popcountll(lo);
		hi_ct += (uint64_t)__builtin_popcountll(hi);
		both_ct += (uint64_t)__builtin_popcountll(lo & hi);
	}
	out[1] = (uint32_t)(lo_ct - both_ct);
	out[2] = (uint32_t)(hi_ct - both_ct);
	out[3] = (uint32_t)both_ct;
	out[0] = N - out[1] - out[2] - out[3];
}

static void *scan_worker(void *argp) {
	scan_arg *a = (scan_arg *)argp;
	const pgo_file *h = a->h;
	for (;;) {
		/* batch claim of 128 variants: plink_freq.cpp:413,434-443 */
		uint32_t start = __atomic_fetch_add(a->next, 128u, __ATOMIC_RELAXED);
		if (start >= a->v_end) {
			break;
		}
		uint32_t stop = start + 128u < a->v_end ? start + 128u : a->v_end;
		for (uint32_t v = start; v < stop; v++) {
			uint32_t *o = a->out + 4 * (size_t)(v - a->v_begin);
			if ((h->vrtype[v] & 7) == 0) {
				counts_words(h->body + h->fpos[v], h->N, o);
			} else {
				pgo_get_counts(h, v, NULL, o);
			}
		}
	}
	return NULL;
}

int pgo_scan_counts_mt(const pgo_file *h, uint32_t v0, uint32_t v1, uint32_t n_threads, uint32_t *out) {
	if (n_threads < 1) {
		n_threads = 1;
	}
	if (n_threads > 256) {
		n_threads = 256;
	}
	uint32_t next = v0;
	scan_arg arg = {h, v1, &next, out, v0};
	pthread_t th[256];
	for (uint32_t t = 1; t < n_threads; t++) {
		pthread_create(&th[t], NULL, scan_worker, &arg);
	}
	scan_worker(&arg);
	for (uint32_t t = 1; t < n_threads; t++) {
		pthread_join(th[t], NULL);
	}
	return 0;
}
