"""Test-side .pgen writer producing every hardcall record type (0, 1, 2, 3, 4, 6, 7).

plink2 is not available to generate fixtures with long difflists (the reference's own
fixtures have N <= 256, so every difflist there is a single group with 1-byte sample
ids).  This writer builds records from a genotype matrix so the product's device
decoder, its host normaliser and the oracle's decoder can be compared on multi-group
difflists and 2/3-byte sample ids.  It is written from the same reading of the format
as the decoders, so it checks them against EACH OTHER, not against pgenlib: those
record shapes are "parity unpinned" (DESIGN.md section 4).

Genotype codes: 0 hom-ref, 1 het, 2 hom-alt, 3 missing."""

from __future__ import annotations

import numpy as np


def _varint(x: int) -> bytes:
    out = bytearray()
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


def _varints(values: np.ndarray) -> bytes:
    """Concatenated varints of a whole array (vectorised: difflists can hold millions of gaps)."""
    d = values.astype(np.uint64)
    if d.size == 0:
        return b""
    nb = np.ones(d.size, dtype=np.int64)
    for j in range(1, 5):
        nb += d >= (1 << (7 * j))
    start = np.cumsum(nb) - nb
    out = np.zeros(int(nb.sum()), dtype=np.uint8)
    for j in range(5):
        sel = nb > j
        if not sel.any():
            break
        byte = ((d[sel] >> np.uint64(7 * j)) & np.uint64(0x7F)).astype(np.uint8)
        byte |= ((nb[sel] > j + 1).astype(np.uint8) << 7)
        out[start[sel] + j] = byte
    return out.tobytes()


def _pack2(codes: np.ndarray) -> bytes:
    n = len(codes)
    pad = (-n) % 4
    c = np.concatenate([codes.astype(np.uint8), np.zeros(pad, dtype=np.uint8)]).reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8).tobytes()


def _id_bytes(n: int) -> int:
    return 1 if n < 0x100 else 2 if n < 0x10000 else 3 if n < 0x1000000 else 4


def _difflist(ids: np.ndarray, vals: np.ndarray | None, n_samples: int) -> bytes:
    """vals None: the id-only form the 0x20 dosage track uses."""
    ln = len(ids)
    out = bytearray(_varint(ln))
    if ln == 0:
        return bytes(out)
    w = _id_bytes(n_samples)
    groups = (ln + 63) // 64
    gaps = []
    for g in range(groups):
        chunk = ids[g * 64:(g + 1) * 64]
        out += int(chunk[0]).to_bytes(w, "little")
        gaps.append(_varints(np.diff(chunk)))
    for g in range(groups - 1):
        out.append((len(gaps[g]) - 63) & 0xFF)  # gap-section byte length, biased; decoders here skip it
    if vals is not None:
        out += _pack2(vals)
    for g in gaps:
        out += g
    return bytes(out)


_INV = np.array([2, 1, 0, 3], dtype=np.uint8)


def encode_record(kind: int, g: np.ndarray, base: np.ndarray | None) -> bytes:
    n = len(g)
    if kind == 0:
        return _pack2(g)
    if kind == 1:
        counts = np.bincount(g, minlength=4)
        low, high = sorted(np.argsort(-counts, kind="stable")[:2].tolist())
        code = low * 4 + (high - low)
        bits = np.packbits((g == high).astype(np.uint8), bitorder="little").tobytes()
        diff = np.flatnonzero((g != low) & (g != high))
        return bytes([code]) + bits + _difflist(diff, g[diff], n)
    if kind in (4, 6, 7):
        const = {4: 0, 6: 2, 7: 3}[kind]
        diff = np.flatnonzero(g != const)
        return _difflist(diff, g[diff], n)
    if kind in (2, 3):
        assert base is not None
        target = g if kind == 2 else _INV[g]  # type 3 is patched first, then inverted
        diff = np.flatnonzero(target != base)
        return _difflist(diff, target[diff], n)
    raise ValueError(kind)


def choose_kinds(geno: np.ndarray, rng: np.random.Generator) -> list[int]:
    """A legal, varied assignment: LD types only after a non-LD record."""
    kinds = []
    have_base = False
    for row in geno:
        options = [0, 1, 4, 6, 7] + ([2, 3] if have_base else [])
        k = int(rng.choice(options))
        kinds.append(k)
        have_base |= k not in (2, 3)
    return kinds


def phase_track(g: np.ndarray, rng: np.random.Generator) -> bytes:
    """vrtype bit 0x10: first bit 0 = every het is phased, then one phaseinfo bit per het; first bit 1 =
    a phasepresent bit per het, then one phaseinfo bit per phased het."""
    het_ct = int((g == 1).sum())
    explicit = bool(rng.integers(0, 2))
    if not explicit:
        bits = np.concatenate([[0], rng.integers(0, 2, het_ct)]).astype(np.uint8)
        return np.packbits(bits, bitorder="little").tobytes()
    present = rng.integers(0, 2, het_ct).astype(np.uint8)
    head = np.packbits(np.concatenate([[1], present]).astype(np.uint8), bitorder="little").tobytes()
    info = rng.integers(0, 2, int(present.sum())).astype(np.uint8)
    return head + np.packbits(info, bitorder="little").tobytes()


def dosage_track(kind: int, dos16: np.ndarray, n: int) -> bytes:
    """kind 0x20: sample-id list + values; 0x40: one value per sample (65535 = none); 0x60: presence bits + values."""
    have = np.flatnonzero(dos16 != 0xFFFF)
    vals = dos16[have].astype("<u2").tobytes()
    if kind == 0x20:
        return _difflist(have, None, n) + vals
    if kind == 0x40:
        return dos16.astype("<u2").tobytes()
    if kind == 0x60:
        return np.packbits((dos16 != 0xFFFF).astype(np.uint8), bitorder="little").tobytes() + vals
    raise ValueError(kind)


def _bits(values: np.ndarray, width: int) -> bytes:
    """values packed LSB-first at `width` bits each (width in 0, 1, 2, 4, 8, 16)."""
    if width == 0 or len(values) == 0:
        return b""
    v = values.astype(np.uint64)
    bits = ((v[:, None] >> np.arange(width, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8).reshape(-1)
    return np.packbits(bits, bitorder="little").tobytes()


def multiallelic_track(g: np.ndarray, alleles: int, rng: np.random.Generator) -> bytes:
    """vrtype bit 0x08, as this repo reads the PLINK 2 specification (parity unpinned -- no reference fixture):
    one byte of modes (low nibble: patches of the genotype-1 calls, high nibble: of the genotype-2 calls; 0 = a bit
    per such call, 1 = sample-id list, 15 = none), then per part its selector and the patched calls' allele codes."""
    n = len(g)
    out = bytearray()
    parts = []
    modes = 0
    for part, code in ((0, 1), (1, 2)):
        calls = np.flatnonzero(g == code)
        mode = int(rng.choice([0, 1, 15])) if len(calls) else 15
        patched = rng.random(len(calls)) < 0.4 if mode != 15 else np.zeros(len(calls), dtype=bool)
        if mode == 0:
            sel = np.packbits(patched.astype(np.uint8), bitorder="little").tobytes()
        elif mode == 1:
            sel = _difflist(calls[patched], None, n)
        else:
            sel = b""
        k = int(patched.sum())
        if part == 0:
            width = 0 if alleles == 3 else 1 if alleles == 4 else 2 if alleles <= 6 else 4 if alleles <= 18 else 8
            vals = _bits(rng.integers(0, max(1, alleles - 2), k), width)
        else:
            width = 1 if alleles == 3 else 4 if alleles <= 5 else 8 if alleles <= 17 else 16
            vals = _bits(rng.integers(0, 1 << min(width, 8), k), width)
        modes |= mode << (4 * part)
        parts.append(sel + vals)
    out.append(modes)
    for p in parts:
        out += p
    return bytes(out)


def write_pgen(path: str, geno: np.ndarray, kinds: list[int], dosage: np.ndarray | None = None,
               dosage_kinds: list[int] | None = None, phase_rng: np.random.Generator | None = None,
               allele_cts: list[int] | None = None, aux1_rng: np.random.Generator | None = None) -> None:
    """geno: [M][N] codes.  Writes mode 0x10 with 8-bit vrtypes and 4-byte record lengths.
    dosage: [M][N] uint16 (65535 = no explicit dosage), written per variant as dosage_kinds[v] (0 = no track).
    phase_rng: give every variant with a het a phase track (bit 0x10) of random content.
    allele_cts: alleles per variant (REF + ALTs); a variant with more than two gets a multiallelic track (0x08)
    behind its main track -- geno holds its calls with the ALT alleles collapsed, which is what PgrGet returns --
    and the header carries one byte of ALT allele count per variant."""
    m, n = geno.shape
    records = []
    kinds = list(kinds)
    base = None
    for v in range(m):
        g = geno[v].astype(np.uint8)
        k = kinds[v]
        rec = encode_record(k, g, base)
        if k not in (2, 3):
            base = g
        if allele_cts is not None and allele_cts[v] > 2:
            rec += multiallelic_track(g, allele_cts[v], aux1_rng or np.random.default_rng(v))
            kinds[v] |= 0x08
        if phase_rng is not None and (g == 1).any():
            rec += phase_track(g, phase_rng)
            kinds[v] |= 0x10
        if dosage is not None and dosage_kinds[v]:
            rec += dosage_track(dosage_kinds[v], dosage[v], n)
            kinds[v] |= dosage_kinds[v]
        records.append(rec)
    blocks = (m + 65535) // 65536
    head = bytearray([0x6C, 0x1B, 0x10]) + int(m).to_bytes(4, "little") + int(n).to_bytes(4, "little")
    ac_bytes = 1 if allele_cts is not None else 0
    head.append(0x40 | (ac_bytes << 4) | 4 | 3)  # no nonref flags; ALT allele counts; 8-bit vrtypes; 4-byte lengths
    tables = bytearray()
    table_len = blocks * 8 + sum(min(65536, m - b * 65536) * (5 + ac_bytes) for b in range(blocks))
    body_at = len(head) + table_len
    offsets = []
    fp = body_at
    for b in range(blocks):
        offsets.append(fp)
        lo, hi = b * 65536, min(m, (b + 1) * 65536)
        tables += bytes(kinds[lo:hi])
        for v in range(lo, hi):
            tables += len(records[v]).to_bytes(4, "little")
            fp += len(records[v])
        if ac_bytes:
            tables += bytes(int(a) - 1 for a in allele_cts[lo:hi])
    with open(path, "wb") as f:
        f.write(head)
        for o in offsets:
            f.write(int(o).to_bytes(8, "little"))
        f.write(tables)
        for r in records:
            f.write(r)


def rare_matrix(m: int, n: int, rng: np.random.Generator) -> np.ndarray:
    """Mostly-constant rows with a spread of minor-value rates so difflists span 0..many groups."""
    geno = np.empty((m, n), dtype=np.uint8)
    for v in range(m):
        major = int(rng.choice([0, 0, 0, 2, 3]))
        rate = float(rng.choice([0.0, 0.0005, 0.01, 0.05, 0.3]))
        row = np.full(n, major, dtype=np.uint8)
        hit = rng.random(n) < rate
        row[hit] = rng.integers(0, 4, hit.sum(), dtype=np.uint8)
        if v and rng.random() < 0.4:  # near-copies of the previous row make LD records worthwhile
            row = geno[v - 1].copy()
            flip = rng.random(n) < rate / 4
            row[flip] = rng.integers(0, 4, flip.sum(), dtype=np.uint8)
        geno[v] = row
    return geno
