#!/usr/bin/env python3
"""bench.py -- the north-star measurement: plink_freq over a synthetic
1,000,000-variant x 500,000-sample .pgen matrix resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the rank's resident variants:
the genotype-class tally kernel (PgrGetCounts for every variant), the ALT_FREQ /
OBS_CT epilogue kernel, and the device->pinned-host copy of the per-variant
results (what the table function hands to DuckDB).  Inputs are generated in HBM
before the timed region (seeded counter-based generator).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), variants
sharded across ranks, no data-path collective for plink_freq;
only the timing barrier / max-over-ranks uses the process group.  Under
`python -m torch.distributed.run` the ranks are the launcher's; started bare
(`python bench.py --gpus N`, no WORLD_SIZE) this process touches no GPU and
starts the N rank processes itself, relays rank 0's JSON line and exits non-zero
if the node has fewer than N GPUs or any rank fails.  The default is STRONG
scaling: --variants is the whole matrix (the north star's fixed 1 M x 500 k
file) split into N contiguous ranges; --scaling weak gives every rank its own
--variants rows.

After the timed region the last step's host-side results are checked (rows tally
to N, sampled rows equal a host recomputation from the seeded generator, the
fused pass's per-sample missing counts add up to the per-variant ones) and the
line carries "verified": true; a failed check exits non-zero without a line.

Prints ONE JSON line on rank 0 (see the driver contract), including
  "roofline":     algorithmic HBM bytes of the tally kernel / its mean launch
                  duration (HIP events on the launch stream), against 8 TB/s;
  "cpu_baseline": the CPU oracle run with the reference's scan structure
                  (T threads claiming 128-variant batches) on a bounded sample.

Other workloads (parity-tested configs, for DESIGN.md / profiles/):
  --workload fused   plink_freq + plink_hardy + plink_missing from one tally pass
  --workload unpack  read_pgen genotype column (2-bit -> int8 + validity)
  --workload score   plink_score, 16 weight columns
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20260807
MISSING_RATE = 0.02
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector/matrix peak
I8_PEAK_TOPS = 5000.0       # dense int8 MFMA: 2x the bf16 rate (MI355X_MICROARCH.md, Matrix cores)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variants", type=int, default=1_000_000, help="variants in total (strong scaling, the default) or per rank (weak)")
    ap.add_argument("--samples", type=int, default=500_000)
    ap.add_argument("--ld-variants", type=int, default=20000, help="anchors of the ld workload")
    ap.add_argument("--ld-window", type=int, default=64, help="partners per anchor of the ld workload")
    ap.add_argument("--workload", choices=["freq", "fused", "unpack", "score", "pca", "ld", "samplecounts", "missingsample", "dosagefreq", "dosagescore"], default="freq")
    ap.add_argument("--n-pcs", type=int, default=10)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): --variants is the whole matrix, split across the GPUs; "
                         "weak: every GPU holds --variants rows of its own")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--cpu-sample-variants", type=int, default=8192)
    ap.add_argument("--dosage-rate", type=float, default=0.1,
                    help="dosage workloads: fraction of samples with an explicit dosage per variant")
    ap.add_argument("--score-cols", type=int, default=16)
    ap.add_argument("--score-no-dosage-sum", action="store_true",
                    help="score workload without NAMED_ALLELE_DOSAGE_SUM (what SELECT IID, SCORE_SUM projects)")
    ap.add_argument("--configs", choices=["auto", "all", "none"], default="auto",
                    help="auto: the default single-GPU freq run at 1 M x 500 k also times BASELINE configs 2-5 "
                         "(fused, unpack, score x16 / x1, pca) on the same matrix and attaches them as \"configs\"")
    ap.add_argument("--config-steps", type=int, default=5, help="timed steps per extra config")
    ap.add_argument("--no-sql", dest="sql", action="store_false",
                    help="skip the table-function section (plink_freq / hardy / missing through the shells)")
    return ap.parse_args()


def cpu_baseline(ds, n_samples, budget_s, sample_variants):
    """The oracle (kind 'port': the reference binary cannot be built)
    scanning a bounded sample of the same matrix with the reference's thread structure."""
    import numpy as np

    from oracle import oracle

    v1 = min(ds.v_end, ds.v_begin + sample_variants)
    rows = ds.copy_rows_to_host(ds.v_begin, v1)
    m = rows.shape[0]
    head = bytes([0x6c, 0x1b, 0x02]) + int(m).to_bytes(4, "little") + int(n_samples).to_bytes(4, "little") + b"\x40"
    image = np.concatenate([np.frombuffer(head, dtype=np.uint8), rows.reshape(-1)])
    del rows
    pg = oracle.Pgen(mem=image)
    cores = os.cpu_count() or 1
    # the reference's default: min(range/500 + 1, 16) scan threads (src/plink_freq.cpp:84-87)
    threads = max(1, min(m // 500 + 1, 16, cores))
    pg.scan_counts_mt(0, min(m, 256), threads)  # touch
    passes, t0 = 0, time.perf_counter()
    while True:
        pg.scan_counts_mt(0, m, threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    geno_per_s = passes * m * n_samples / dt
    return {
        "value": geno_per_s,
        "unit": "genotypes/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{m} variants x {n_samples} samples of the same matrix, {passes} passes in {dt:.1f} s, "
                  f"{threads} threads claiming 128-variant batches ({cores} host cores visible)",
    }


def load_traffic(workload, variants, samples):
    """Per-launch HBM bytes from the committed rocprofv3 PMC pass (profiles/traffic.json),
    if it was collected for this exact launch shape."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(workload)
        if e and e.get("variants") == variants and e.get("samples") == samples:
            return e.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as fresh child processes.
    This process has not imported torch or touched HIP (a process that has initialised the GPU must
    not be re-executed), and it does not: the device count comes from a probe child."""
    import signal
    import socket
    import subprocess

    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True)
    try:
        have = int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit(f"bench.py: could not count the GPUs of this node: {probe.stderr.strip()[-300:]}")
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s); "
                         f"no line is reported for a rank count that was not run")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                         start_new_session=True))
    rc = 0
    pending = list(children)
    while pending and rc == 0:
        for c in list(pending):
            try:
                code = c.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(c)
            if code != 0:
                rc = code
    for c in pending:  # a rank failed: the others would wait in a collective forever
        try:
            os.killpg(c.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
    for c in pending:
        try:
            c.wait(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(c.pid, signal.SIGKILL)
    if rc != 0:
        raise SystemExit(f"bench.py: a rank exited with status {rc}; no line reported")


def i8_contract_ops(n_contract, n_out, n_cols, planes, extras):
    """int8 multiply-adds x 2 of one contraction on v_mfma_i32_16x16x64_i8 (score_i8.hip): 7 digit columns per
    real column (+6 with plink_score's dosage-sum / missing-count columns), 16 per tile, <= 22 real columns per
    pass, one instruction per plane, tile and 16 outputs x 64 contracted rows."""
    ops = 0.0
    for c0 in range(0, n_cols, 22):
        tiles16 = (7 * min(22, n_cols - c0) + (6 if extras else 0) + 15) // 16
        ops += 2.0 * (((n_contract + 127) // 128) * 128) * (((n_out + 15) // 16) * 16) * 16 * planes * tiles16
    return ops


def host_tallies(np, rec, n):
    """{hom_ref, het, hom_alt, missing} of one packed 2-bit record, recomputed on the host with numpy."""
    codes = (rec[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3
    return np.bincount(codes.reshape(-1)[:n], minlength=4).astype(np.int64), codes.reshape(-1)[:n]


def oracle_image(np, rows, n):
    """The rows (uint8[k][ceil(N/4)]) as a fixed-width .pgen image (mode 0x02) the oracle opens from memory."""
    from oracle import oracle
    k = rows.shape[0]
    head = bytes([0x6c, 0x1b, 0x02]) + int(k).to_bytes(4, "little") + int(n).to_bytes(4, "little") + b"\x40"
    return oracle.Pgen(mem=np.concatenate([np.frombuffer(head, dtype=np.uint8), rows.reshape(-1)]))


def recount_rows(np, ds, picks, v_begin, n):
    """The oracle's tallies and calls of the listed rows AS THEY SIT IN HBM (copied back, not regenerated):
    the check that can fail when a kernel drops or double-counts part of a row."""
    rows = np.concatenate([ds.copy_rows_to_host(v_begin + int(r), v_begin + int(r) + 1) for r in picks])
    pg = oracle_image(np, rows, n)
    return pg, pg.scan_counts_mt(0, len(picks), min(8, os.cpu_count() or 1)).astype(np.int64)


def verify(args, wl, C, env):
    """Size-independent checks on the LAST timed step's results (the timed region has ended and was
    synchronised).  True / False, or None for a workload without a check."""
    L, np, torch, ds = C.L, C.np, C.torch, C.ds
    n, m = env["n"], env["m"]
    v_begin = env["v_begin"]
    if m == 0:
        return True
    picks = sorted({0, m // 2, m - 1})
    rng = np.random.default_rng(SEED + 99)
    if wl in ("freq", "fused"):
        hc = env["h_counts"].numpy().astype(np.int64)
        # (hom_ref is N minus the other three in every tally kernel, so "rows add up to N" cannot fail; what
        # can is a recount:) 256 random rows plus the first and last, copied back from HBM and tallied by the
        # oracle's multi-threaded scan, bit-exact
        recount = sorted(set(rng.choice(m, size=min(m, 256), replace=False).tolist()) | set(picks))
        _, want = recount_rows(np, ds, recount, v_begin, n)
        if not np.array_equal(want, hc[recount]):
            bad = [r for i, r in enumerate(recount) if not np.array_equal(want[i], hc[r])]
            print(f"verify: {len(bad)} of {len(recount)} recounted rows differ from the oracle, first: variant "
                  f"{v_begin + bad[0]}: {hc[bad[0]]} != {want[recount.index(bad[0])]}", file=sys.stderr)
            return False
        obs = hc[:, :3].sum(axis=1)
        af = (hc[:, 1] + 2 * hc[:, 2]) / np.where(obs > 0, 2.0 * obs, np.nan)
        hf = env["h_freq"].numpy()
        if not (np.array_equal(env["h_obs"].numpy(), 2 * obs) and np.array_equal(hf[obs > 0], af[obs > 0])):
            print("verify: ALT_FREQ / OBS_CT differ from the host arithmetic on the tallies", file=sys.stderr)
            return False
        for r in picks:
            want, _ = host_tallies(np, L.synth_record_host(v_begin + r, n, SEED, MISSING_RATE), n)
            if not np.array_equal(want, hc[r]):
                print(f"verify: variant {v_begin + r}: {hc[r]} != host {want}", file=sys.stderr)
                return False
        if wl == "fused":
            hm = env["h_miss"].numpy().astype(np.int64)
            if int(hm.sum()) != int(hc[:, 3].sum()):
                print("verify: per-sample missing counts do not add up to the per-variant ones", file=sys.stderr)
                return False
            # ... and 64 random samples' columns against a second kernel (k_class_cols1) -- the column tally of
            # the fused pass and the per-sample pass share no code path below the loads
            alone = ds.missing_per_sample(v_begin, v_begin + m).astype(np.int64)
            cols = rng.choice(n, size=min(n, 64), replace=False)
            if not np.array_equal(alone[cols], hm[cols]):
                print("verify: per-sample missing counts differ from the per-sample kernel's", file=sys.stderr)
                return False
        return True
    if wl == "unpack":
        chunk = env["chunk"]
        last0 = (m - 1) // chunk * chunk  # the rows of the last launch
        out = env["d_out"]
        val = env["d_val"]
        # 64 random rows of the last launch plus its first and last, copied back from HBM and decoded by the oracle
        rows = sorted(set((last0 + rng.choice(m - last0, size=min(m - last0, 64), replace=False)).tolist()) | {last0, m - 1})
        pg, _ = recount_rows(np, ds, rows, v_begin, n)
        for i, r in enumerate(rows):
            codes = pg.geno(i)  # {0, 1, 2, -9}
            got = out[r - last0, :n].cpu().numpy()
            want = np.where(codes < 0, 0, codes).astype(np.int8)
            bits = np.unpackbits(val[r - last0].cpu().numpy().view(np.uint8), bitorder="little")[:n]
            if not (np.array_equal(got, want) and np.array_equal(bits, (codes >= 0).astype(np.uint8))):
                print(f"verify: unpacked variant {v_begin + r} differs from the oracle's decode", file=sys.stderr)
                return False
        return True
    if wl in ("missingsample", "samplecounts"):
        hc = ds.counts_range().astype(np.int64)
        if wl == "missingsample":
            ok = int(env["d_miss"][:n].cpu().numpy().astype(np.int64).sum()) == int(hc[:, 3].sum())
        else:
            cls = env["d_cls"][:, :n].cpu().numpy().astype(np.int64).sum(axis=1)
            ok = np.array_equal(cls, hc[:, 1:4].sum(axis=0))
        if not ok:
            print("verify: per-sample tallies do not add up to the per-variant ones", file=sys.stderr)
        return bool(ok)
    if wl == "score":
        # checksum of checksums: the column sums over samples of SCORE_SUM follow from the per-variant tallies
        # (mean imputation: a missing call contributes the variant's mean dosage, src/plink_score.cpp:598-631)
        hc = ds.counts_range().astype(np.float64)
        nonmiss = hc[:, :3].sum(axis=1)
        alt = hc[:, 1] + 2 * hc[:, 2]
        mean = np.where(nonmiss > 0, alt / np.maximum(nonmiss, 1), 0.0)
        per_variant = np.where(nonmiss > 0, alt + hc[:, 3] * mean, 0.0)
        want = per_variant @ env["w"]
        got = env["d_score"].cpu().numpy().sum(axis=0)
        scale = np.abs(per_variant[:, None] * env["w"]).sum(axis=0)
        if env["dist"] is not None:
            # rank 0 holds the reduced sums of every shard: what they must add up to is the sum of the shards' own
            # expectations (every rank takes part in the all-reduce; ranks other than 0 have nothing to compare)
            t = torch.tensor(np.concatenate([want, scale]), dtype=torch.float64, device=C.dev)
            env["dist"].all_reduce(t)
            want, scale = np.split(t.cpu().numpy(), 2)
            if C.rank != 0:
                return True
        ok = bool(np.all(np.abs(got - want) <= 1e-9 * scale))
        if not ok:
            print(f"verify: score column sums {got} != {want} from the tallies", file=sys.stderr)
        return ok
    if wl == "ld":
        # three pairs recomputed on the host from the synthetic records: {n, sum a, sum b, sum ab, sum a^2, sum b^2}
        # over the samples called at both variants (src/plink_ld.cpp's sample loop)
        hs = env["h_sums"].numpy().astype(np.int64)
        p_a, p_b = env["p_a"], env["p_b"]
        for i in sorted({0, len(p_a) // 2, len(p_a) - 1}):
            _, a = host_tallies(np, L.synth_record_host(int(p_a[i]), n, SEED, MISSING_RATE), n)
            _, b = host_tallies(np, L.synth_record_host(int(p_b[i]), n, SEED, MISSING_RATE), n)
            both = (a != 3) & (b != 3)
            a, b = a[both].astype(np.int64), b[both].astype(np.int64)
            want = [int(both.sum()), int(a.sum()), int(b.sum()), int((a * b).sum()), int((a * a).sum()), int((b * b).sum())]
            if want != hs[i].tolist():
                print(f"verify: pair ({p_a[i]}, {p_b[i]}): {hs[i].tolist()} != host {want}", file=sys.stderr)
                return False
        return True
    if wl == "pca":
        if not env["pca_vec"]:
            return None
        # identities that hold at any size: orthonormal eigenvectors, positive descending eigenvalues that repeat from
        # step to step, and their sum bounded by the total variance of the normalised matrix (from the tallies)
        vec, evs = env["pca_vec"][0], np.array(env["pca_ev"])
        k = vec.shape[1]
        ok = bool(np.allclose(vec.T @ vec, np.eye(k), atol=1e-8))
        ok = ok and bool(np.all(evs[-1] > 0) and np.all(np.diff(evs[-1]) <= 0))
        ok = ok and bool(np.allclose(evs, evs[-1], rtol=1e-9))
        counts = env["counts"][env["keep"]]
        c, inv = env["p_center"], env["p_inv"]
        total = sum((counts[:, g] * ((g - c) * inv) ** 2).sum() for g in range(3))
        if env["dist"] is not None:  # the variance of the whole matrix: every shard's share (all ranks hold the same PCs)
            t = torch.tensor([total], dtype=torch.float64, device=C.dev)
            env["dist"].all_reduce(t)
            total = float(t.item())
        ok = ok and bool(evs[-1].sum() <= total / env["m_total"] * (1 + 1e-9))
        if not ok:
            print(f"verify: plink_pca identities fail: eigenvalues {evs[-1]}, total variance / M {total / env['m_total']}",
                  file=sys.stderr)
        return ok
    if wl in ("dosagefreq", "dosagescore"):
        if env["dist"] is not None:
            return None  # as above: the single-GPU line carries the check
        # the per-variant {sum, sum of squares, count} on the 16384 scale (k_dosage_sums) against the moments of three
        # rows unpacked to doubles by a different kernel (k_dosage_unpack)
        sums = env["h_sums"].numpy().astype(np.uint64) if wl == "dosagefreq" else ds.dosage_sums().astype(np.uint64)
        rows = ds.dosage_unpack(vidx=[v_begin + r for r in picks])
        for i, r in enumerate(picks):
            u = np.rint(rows[i][rows[i] != -9.0] * 16384.0).astype(np.uint64)
            if (int(u.sum()), int((u * u).sum()), len(u)) != tuple(int(x) for x in sums[r]):
                print(f"verify: dosage moments of variant {v_begin + r} differ from its unpacked row", file=sys.stderr)
                return False
        if wl == "dosagefreq":
            return True
        # checksum of checksums: under mean imputation a sample with neither dosage nor call contributes the
        # variant's mean, so SCORE_SUM summed over samples is sum_v w_v * (dosage sum of v) * N / (samples observed)
        total, seen = sums[:, 0].astype(np.float64) / 16384.0, sums[:, 2].astype(np.float64)
        per_variant = np.where(seen > 0, total * n / np.maximum(seen, 1.0), 0.0)
        want = float(per_variant @ env["weights"][:, 0])
        got = float(env["d_score"].cpu().numpy().sum())
        scale = float(np.abs(per_variant * env["weights"][:, 0]).sum())
        ok = abs(got - want) <= 1e-9 * scale and bool((env["d_ac"].cpu().numpy() == 2 * m).all())
        if not ok:
            print(f"verify: dosage score sum {got} != {want} from the per-variant dosage sums (or ALLELE_CT != 2M)",
                  file=sys.stderr)
        return bool(ok)
    return None


def build_workload(args, wl, C, score_cols=None, no_dosage_sum=None):
    """Buffers, the step closure and the roofline inputs of one workload over the rank's resident matrix C.ds.
    Returns a namespace: step(timed), kernel_events, kernel_name, metric, dtype, algo_bytes, algo_flops, i8_ops,
    units_per_step, env (the locals verify() reads)."""
    from types import SimpleNamespace
    np, torch, L, sharding, ds, dist, dev = C.np, C.torch, C.L, C.sharding, C.ds, C.dist, C.dev
    stream, st, n, m, v_begin, v_end, record_bytes = C.stream, C.st, C.n, C.m, C.v_begin, C.v_end, C.record_bytes
    score_cols = args.score_cols if score_cols is None else score_cols
    no_dosage_sum = args.score_no_dosage_sum if no_dosage_sum is None else no_dosage_sum
    i8_ops = None
    kernel_events = []
    units_per_step = m * n  # genotypes
    algo_bytes = None
    algo_flops = None
    dtype = "u32"

    if wl in ("freq", "fused"):
        # Two sets of result buffers: while the tally of step i+1 streams the matrix, the results of step i
        # travel to pinned host memory on a side stream (the copy engine).
        side = torch.cuda.Stream(device=dev)
        d_counts = [torch.empty((m, 4), dtype=torch.int32, device=dev) for _ in range(2)]
        d_freq = [torch.empty(m, dtype=torch.float64, device=dev) for _ in range(2)]
        d_obs = [torch.empty(m, dtype=torch.int32, device=dev) for _ in range(2)]
        h_counts = torch.empty((m, 4), dtype=torch.int32, pin_memory=True)
        h_freq = torch.empty(m, dtype=torch.float64, pin_memory=True)
        h_obs = torch.empty(m, dtype=torch.int32, pin_memory=True)
        if wl == "fused":
            d_lnp = [torch.empty(m, dtype=torch.float64, device=dev) for _ in range(2)]
            h_lnp = torch.empty(m, dtype=torch.float64, pin_memory=True)
            d_miss = [torch.empty((n + 63) // 64 * 64, dtype=torch.int32, device=dev) for _ in range(2)]
            h_miss = torch.empty(n, dtype=torch.int32, pin_memory=True)
        algo_bytes = m * record_bytes  # ceil(N/4) bytes read per variant
        tallied = [torch.cuda.Event() for _ in range(2)]   # the tally of a buffer set is complete
        drained = [None, None]                             # its derived columns have left for the host
        step_no = [0]

        def step(timed):
            b = step_no[0] & 1
            step_no[0] += 1
            if drained[b] is not None:
                stream.wait_event(drained[b])  # the set is free again
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            if wl == "fused":
                # row sums (class tallies) and column sums (per-sample missing) in ONE pass
                ds.fused_tally_dev(v_begin, v_end, d_counts[b].data_ptr(), d_miss[b].data_ptr(), st)
            else:
                ds.counts_range_dev(v_begin, v_end, d_counts[b].data_ptr(), st)
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
            L.freq_from_counts_dev(d_counts[b].data_ptr(), m, d_freq[b].data_ptr(), d_obs[b].data_ptr(), st)
            tallied[b].record(stream)
            side.wait_event(tallied[b])
            if wl == "fused":
                # plink_hardy: exact test per variant from the same counts.  A latency-bound compute kernel
                # (one lane per variant): it runs on the side stream UNDER the next step's tally, which leaves
                # most of every CU's issue slots free; plink_missing variant mode: counts[:,3]; sample mode:
                # the column sums of the same pass
                L.hwe_lnp_batch_dev(d_counts[b].data_ptr(), m, d_lnp[b].data_ptr(), False, side.cuda_stream)
            with torch.cuda.stream(side):
                h_counts.copy_(d_counts[b], non_blocking=True)
                h_freq.copy_(d_freq[b], non_blocking=True)
                h_obs.copy_(d_obs[b], non_blocking=True)
                if wl == "fused":
                    h_lnp.copy_(d_lnp[b], non_blocking=True)
                    h_miss.copy_(d_miss[b][:n], non_blocking=True)
                drained[b] = torch.cuda.Event()
                drained[b].record(side)

        kernel_name = "k_counts_block" if wl == "freq" else "k_fused_tally"
        metric = "plink_freq genotypes/s" if wl == "freq" else "plink_freq+hardy+missing genotypes/s"
    elif wl == "unpack":
        # read_pgen genotype column: output is 4.5x the input, streamed in row chunks
        # the way DuckDB consumes it (2048-row vectors; here 8 vectors per launch)
        chunk = min(m, 16384)
        out_pitch = (n + 15) // 16 * 16
        val_words = (n + 63) // 64
        d_out = torch.empty((chunk, out_pitch), dtype=torch.int8, device=dev)
        d_val = torch.empty((chunk, val_words), dtype=torch.int64, device=dev)
        algo_bytes = chunk * (record_bytes + n + val_words * 8)

        def step(timed):
            for c0 in range(0, m, chunk):
                c1 = min(m, c0 + chunk)
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                ds.unpack_range_dev(v_begin + c0, v_begin + c1, d_out.data_ptr(), out_pitch, d_val.data_ptr(), 0, st)
                if timed and c1 - c0 == chunk:
                    e1.record(stream)
                    kernel_events.append((e0, e1))

        kernel_name = "k_unpack"
        metric = "read_pgen genotypes/s"
        dtype = "u8"
    elif wl in ("dosagefreq", "dosagescore"):
        # plink_freq(dosage := true) / plink_score over explicit dosages: synthetic 0x60-style tracks on every
        # variant, --dosage-rate of the samples carrying a value (the rest fall back to their hardcall)
        ds.synth_add_dosage(args.dosage_rate, SEED + 7)
        words = (n + 63) // 64
        # 2-bit record + presence bits + the explicit values; the score's explicit-entry sweep also needs the
        # per-word ranks (its second read of the record, hardcall sweep then dosage sweep, is not counted)
        algo_bytes = m * record_bytes + m * words * 8 + 2 * int(ds.info.dosage_value_ct)
        via_records = (wl == "dosagescore" and args.dosage_rate < 0.4
                       and os.environ.get("PGH_SCORE_DOSAGE_RECORDS", "1") != "0")
        if via_records:
            # sparse tracks: the hardcall contraction reads the 2-bit records, k_score_dosage_records the 4-byte entry
            # records (value, tile element, hardcall) and two ranks per 4096-sample tile -- no bits, no second
            # read of the rows
            algo_bytes = m * record_bytes + 4 * int(ds.info.dosage_value_ct) + m * ((words + 63) // 64) * 8
        elif wl == "dosagescore":
            algo_bytes += m * words * 4
        dtype = "u16"
        if wl == "dosagefreq":
            d_sums = torch.empty((m, 3), dtype=torch.int64, device=dev)
            h_sums = torch.empty((m, 3), dtype=torch.int64, pin_memory=True)

            def step(timed):
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                ds.dosage_sums_dev(v_begin, v_end, d_sums.data_ptr(), st)
                if timed:
                    e1.record(stream)
                    kernel_events.append((e0, e1))
                h_sums.copy_(d_sums, non_blocking=True)

            kernel_name = "k_dosage_sums"
            metric = "plink_freq(dosage) genotypes/s"
        else:
            rng = np.random.default_rng(SEED + 1)
            weights = rng.standard_normal((m, 1))
            plan = ds.score_plan(np.arange(v_begin, v_end, dtype=np.uint32), weights)
            d_score = torch.empty((n, 1), dtype=torch.float64, device=dev)
            d_dos = torch.empty(n, dtype=torch.float64, device=dev)
            d_ac = torch.empty(n, dtype=torch.int32, device=dev)
            dtype = "f64"

            def step(timed):
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                plan.run_dev(d_score.data_ptr(), d_dos.data_ptr(), d_ac.data_ptr(), st)
                if timed:
                    e1.record(stream)
                    kernel_events.append((e0, e1))
                if dist is not None:
                    dist.reduce(d_score, dst=0)

            # the plan picks the kernel by track density (api_analysis.cpp): all samples explicit, >= 40 %, below
            kernel_name = ("k_score_dosage_full" if args.dosage_rate >= 1.0 else
                           "k_score_dosage" if args.dosage_rate >= 0.42 else
                           "k_score_i8 + k_score_dosage_records" if via_records else "k_score_i8 + k_score_dosage_fix")
            metric = "plink_score(dosage) genotypes/s"
    elif wl == "missingsample":
        # plink_missing mode := 'sample': per-sample missing tallies over every variant (column sums)
        padded = (n + 63) // 64 * 64
        d_miss = torch.empty(padded, dtype=torch.int32, device=dev)
        h_miss = torch.empty(padded, dtype=torch.int32, pin_memory=True)
        algo_bytes = m * record_bytes

        def step(timed):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            ds.missing_per_sample_dev(v_begin, v_end, d_miss.data_ptr(), st)
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
            h_miss.copy_(d_miss, non_blocking=True)

        kernel_name = "k_class_cols1 + k_sum_cols1"
        metric = "plink_missing(sample) genotypes/s"
    elif wl == "samplecounts":
        # read_pfile orient := 'sample', genotypes := 'counts': per-sample {het, hom_alt, missing}
        # tallies over every variant (hom_ref by subtraction) -- one pass, three counter sets per lane
        padded = (n + 63) // 64 * 64
        d_cls = torch.empty((3, padded), dtype=torch.int32, device=dev)
        h_cls = torch.empty((3, padded), dtype=torch.int32, pin_memory=True)
        algo_bytes = m * record_bytes

        def step(timed):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            ds.sample_counts_dev(v_begin, v_end, d_cls.data_ptr(), st)
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
            h_cls.copy_(d_cls, non_blocking=True)

        kernel_name = "k_class_cols3 + k_sum_class_bits"
        metric = "read_pfile sample-orient counts genotypes/s"
    elif wl == "ld":
        # plink_ld windowed shape: every anchor of the first --ld-variants rows against its next
        # --ld-window variants; one launch for all pairs, six integer sums per pair
        mv = min(m, args.ld_variants)
        wdw = args.ld_window
        p_a = np.repeat(np.arange(mv - wdw, dtype=np.uint32), wdw) + np.uint32(v_begin)
        p_b = p_a + np.tile(np.arange(1, wdw + 1, dtype=np.uint32), mv - wdw)
        n_pairs = len(p_a)
        d_sums = torch.empty((n_pairs, 6), dtype=torch.int32, device=dev)
        h_sums = torch.empty((n_pairs, 6), dtype=torch.int32, pin_memory=True)
        units_per_step = n_pairs * n  # sample pairs
        # an anchor row is read once per four partners: 1.25 rows per pair (repeats of a row by later
        # anchors come out of L2/MALL; roofline.traffic would show the HBM share)
        algo_bytes = int(n_pairs * 1.25 * record_bytes)

        def step(timed):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            ds.ld_pairs_dev(p_a, p_b, d_sums.data_ptr(), st)
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
            h_sums.copy_(d_sums, non_blocking=True)

        kernel_name = "k_ld_pairs"
        metric = f"plink_ld sample pairs/s ({wdw} partners per anchor)"
    elif wl == "pca":
        # BASELINE config 5 shape: plink_pca, k = n_pcs (qq = (k+1)*2k), all passes + orthonormalisation
        k = args.n_pcs
        m_p = m if not getattr(C, "pca_variants", 0) else min(m, C.pca_variants)  # config 5 on a slice of the big matrix
        counts = ds.counts_range(v_begin, v_begin + m_p).astype(np.float64)
        obs = counts[:, :3].sum(axis=1)
        af = (counts[:, 1] + 2 * counts[:, 2]) / (2 * np.maximum(obs, 1))
        keep = (obs > 0) & (af > 0) & (af < 1)
        p_vidx = (np.flatnonzero(keep) + v_begin).astype(np.uint32)
        p_center = 2 * af[keep]
        p_inv = 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
        g1 = np.random.default_rng(SEED + 2).standard_normal((n, 2 * k))
        m_eff = len(p_vidx)
        qq = (k + 1) * 2 * k
        m_total = m_eff
        pca_allreduce = None
        if dist is not None:
            # one PCA over every rank's variants: X is split by rows, G2 / Gram blocks / BB are
            # all-reduced over RCCL through the library's callback (pgh_pca_sharded)
            t_total = torch.tensor([m_eff], dtype=torch.int64, device=dev)
            dist.all_reduce(t_total)
            m_total = int(t_total.item())
            pca_allreduce = sharding.device_allreduce(dist)
        algo_bytes = (k + 2) * m_eff * record_bytes
        # k power passes (A+B) + the last Step A + phase 3
        algo_flops = k * (2 * 2.0 * m_eff * n * 2 * k) + 2.0 * m_eff * n * 2 * k + 2.0 * m_eff * n * qq
        # on the int8 matrix cores: Step A = two single-plane contractions over the samples (transposed matrix),
        # Step B and phase 3 = two-plane contractions over the variants
        i8_ops = ((k + 1) * 2 * i8_contract_ops(n, m_eff, 2 * k, 1, False) + k * i8_contract_ops(m_eff, n, 2 * k, 2, False)
                  + i8_contract_ops(m_eff, n, qq, 2, False))
        pca_ev = []
        pca_vec = []

        def step(timed):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            if pca_allreduce is None:
                ev, vec = ds.pca(p_vidx, p_center, p_inv, k, g1)
            else:
                ev, vec = ds.pca_sharded(p_vidx, p_center, p_inv, m_total, k, g1, pca_allreduce)
            pca_ev.append(ev)
            pca_vec[:] = [vec]
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))

        units_per_step = m_p * n
        kernel_name = "pgh_pca (k_score_i8 over rows and over the transposed matrix + orthonormalisation)"
        metric = f"plink_pca genotypes/s (n_pcs={k}, {k + 2} passes)"
        dtype = "i8 x i8 -> i32 (exact base-256 digits of the f64 factors), f64 elsewhere"
    else:  # score
        ncol = score_cols
        rng = np.random.default_rng(SEED + 1)
        vidx = np.arange(v_begin, v_end, dtype=np.uint32)
        w = rng.standard_normal((m, ncol))
        d_score = torch.empty((n, ncol), dtype=torch.float64, device=dev)
        d_dos = torch.empty(n, dtype=torch.float64, device=dev)
        d_ac = torch.empty(n, dtype=torch.int32, device=dev)
        algo_bytes = m * (record_bytes + 8 * ncol)
        algo_flops = 2.0 * m * n * ncol
        # the contraction runs on v_mfma_i32_16x16x64_i8 over exact fixed-point digits of the weights
        # (score_i8.hip): 7 digit columns per weight column + 6 (dosage sum, missing count), 16 per tile,
        # two planes (calls, missing calls) per tile, <= 22 weight columns per pass
        i8_ops = i8_contract_ops(m, n, ncol, 2, True)
        plan = ds.score_plan(vidx, w, None, L.SCORE_MEAN_IMPUTE)  # weights + per-variant tables resident

        def step(timed):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            plan.run_dev(d_score.data_ptr(), 0 if no_dosage_sum else d_dos.data_ptr(), d_ac.data_ptr(), st)
            if timed:
                e1.record(stream)
                kernel_events.append((e0, e1))
            if dist is not None:
                # per-sample partials of the variant shards: RCCL reduce over xGMI
                sharding.reduce_partials(dist, [d_score, d_ac] if no_dosage_sum else [d_score, d_dos, d_ac])

        kernel_name = "k_score_i8"
        metric = f"plink_score genotypes/s ({ncol} weight columns{', no dosage sum' if no_dosage_sum else ''})"
        dtype = "i8 x i8 -> i32 (exact base-256 digits of the f64 coefficients), f64 out"

    env = dict(locals())
    return SimpleNamespace(step=step, kernel_events=kernel_events, kernel_name=kernel_name, metric=metric, dtype=dtype,
                           algo_bytes=algo_bytes, algo_flops=algo_flops, i8_ops=i8_ops, units_per_step=units_per_step,
                           env=env, workload=wl, score_cols=score_cols)

def roofline_of(args, W, kern_avg_ms, n_launches, m, n, store_ceiling=None):
    """The roofline object of one workload from its mean launch duration (HIP events on the launch stream)."""
    wl = W.workload
    if (wl == "score" and W.score_cols >= 2) or wl == "pca":
        # The kernel issues v_mfma_i32_16x16x64_i8 over base-256 digits of the real factors: `frac` is the ISSUED
        # int8 rate (7 digits per real column, tile padding included) over the dense int8 peak -- how busy the matrix
        # pipe is kept, not useful work.  The algorithmic rate (2 flop per genotype and column) is given
        # against the FP64 matrix peak the same contraction would be bound by without the digit form.
        achieved = W.i8_ops / (kern_avg_ms * 1e-3) / 1e12
        f64_eq = W.algo_flops / (kern_avg_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": achieved, "peak": I8_PEAK_TOPS, "unit": "TFLOP/s",
                "frac": achieved / I8_PEAK_TOPS, "frac_is": "issued int8 multiply-adds x 2 over the dense int8 peak "
                "(matrix-pipe utilisation; 7 digit columns per real column)", "traffic": None, "kernel": W.kernel_name,
                "kernel_ms_avg": kern_avg_ms, "launches_timed": n_launches, "ops": "int8 multiply-adds x 2",
                "f64_equivalent_tflops": f64_eq, "algorithmic_flops_frac_of_fp64_peak": f64_eq / FP64_PEAK_TFLOPS,
                "hbm_frac": W.algo_bytes / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    achieved = W.algo_bytes / (kern_avg_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
         "traffic": load_traffic("score1" if wl == "score" else wl, m, n), "kernel": W.kernel_name,
         "kernel_ms_avg": kern_avg_ms, "launches_timed": n_launches, "algorithmic_bytes_per_launch": W.algo_bytes}
    if wl == "unpack" and store_ceiling:
        # 4.5 of every 5.5 bytes of this kernel are stores: what a bare kernel of the same traffic shape (16 B read ->
        # 64 B + 8 B written per lane, no arithmetic) reaches on THIS box, measured in this run
        r["store_ceiling"] = store_ceiling
        r["frac_of_store_ceiling"] = achieved / store_ceiling["GB/s"]
    return r


def store_ceiling_probe(C):
    """~50 ms of a bare kernel with the unpack's traffic shape (pgh_probe_unpack_shape_dev)."""
    torch, L = C.torch, C.L
    n_vec = (2 << 30) // 16  # 2 GB in -> 8 GB + 1 GB out per launch
    src = torch.empty(n_vec * 16, dtype=torch.uint8, device=C.dev)
    dst = torch.empty(n_vec * 64, dtype=torch.uint8, device=C.dev)
    val = torch.empty(n_vec, dtype=torch.int64, device=C.dev)
    src.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.probe_unpack_shape_dev(src.data_ptr(), n_vec, dst.data_ptr(), val.data_ptr(), C.st)
    reps = 5
    e0.record(C.stream)
    for _ in range(reps):
        L.probe_unpack_shape_dev(src.data_ptr(), n_vec, dst.data_ptr(), val.data_ptr(), C.st)
    e1.record(C.stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del src, dst, val
    return {"GB/s": n_vec * 88 / (ms * 1e-3) / 1e9, "ms": ms,
            "shape": "16 B read -> 64 B + 8 B written per lane, non-temporal, no arithmetic"}


def run_timed(W, steps, warmup, barrier):
    for _ in range(warmup):
        W.step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        W.step(True)
    barrier()
    return time.perf_counter() - t0


def sql_section(args, n, expect, threads=16):
    """The same functions THROUGH THE TABLE-FUNCTION SHELLS (bind / init / scan threads, chunks drained as a
    consumer would) over the same resident shape: what a SQL query costs, next to what the kernel costs.
    expect: column sums of the bench's own (verified) tallies of the same seeded matrix; every function's integer
    columns, drained through the shells, must add up to them -- a checksum of checksums over all million rows."""
    from plinking_duck_amd import functions as F
    spec = f"synth:{args.variants}x{n}:{SEED}:{MISSING_RATE}"
    rec = (n + 3) // 4
    out = {}
    calls = [("plink_freq (first call on the file: one tally pass)", "plink_freq", dict(columns=["ID", "ALT_FREQ", "OBS_CT"])),
             ("plink_hardy (served by that pass)", "plink_hardy", dict(columns=["ID", "P_HWE"])),
             ("plink_missing (served by that pass)", "plink_missing", dict(columns=["ID", "F_MISS"])),
             ("plink_missing sample mode (served by that pass)", "plink_missing", dict(mode="sample", columns=["IID", "F_MISS"]))]
    for label, fn, kw in calls:
        r = F.query(fn, spec, threads=threads, drain=True, **kw)
        scan_s = max(r.timing_ms["scan"], 1e-3) * 1e-3
        out[label] = {"rows": len(r), "bind_ms": r.timing_ms["bind"], "init_ms": r.timing_ms["init"],
                      "scan_ms": r.timing_ms["scan"], "threads": r.threads,
                      "scan_genotypes_per_s": args.variants * n / scan_s,
                      "scan_frac_of_hbm_roofline": args.variants * rec / scan_s / 1e9 / HBM_PEAK_GBPS}
    # the rows themselves: integer columns only (their drained checksum is the plain sum of the cells)
    checks = [("plink_freq", {}, "OBS_CT", 2 * (expect["hom_ref"] + expect["het"] + expect["hom_alt"])),
              ("plink_freq", {"counts": True}, "HET_CT", expect["het"]),
              ("plink_hardy", {}, "HOM_ALT_CT", expect["hom_alt"]),
              ("plink_missing", {}, "MISSING_CT", expect["missing"]),
              ("plink_missing", {"mode": "sample"}, "MISSING_CT", expect["missing"])]
    for fn, kw, col, want in checks:
        r = F.query(fn, spec, threads=threads, drain=True, columns=[col], **kw)
        if int(r.checksum) != int(want):
            raise SystemExit(f"bench.py: {fn}({kw}) through the shells: sum of {col} = {r.checksum}, the tallies say {want}; "
                             f"no line reported")
    out["verified"] = True
    return out


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    from types import SimpleNamespace

    import numpy as np
    import torch

    import plinking_duck_amd.lib as L
    from plinking_duck_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) were launched (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # PGH_BENCH_ONE_GPU_REHEARSAL=1: every rank on GPU 0 and the process group over gloo (RCCL refuses two ranks on
    # one device) -- the rank logic of an N-GPU run (shard ranges, reductions, max-over-ranks timing, rank-0 line) on
    # a one-GPU box; the line says so in `config` and its value means nothing
    rehearsal = os.environ.get("PGH_BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    L.set_device(local_rank)
    dist = None
    # under torch.distributed.run the process group comes up even for one rank, so the
    # collective path of the N-rank run is the path that runs (and is rehearsed on one GPU)
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.samples
    if args.workload.startswith("dosage") and args.variants == 1_000_000:
        args.variants = 250_000  # rows + presence bits + ranks + values of 1M variants do not fit one GPU
    if args.workload == "pca" and args.variants == 1_000_000:
        args.variants = 100_000  # BASELINE.json's plink_pca configuration: 100k variants x 500k samples
    v_begin, v_end = sharding.shard_range(rank, world, args.variants, args.scaling)
    m = v_end - v_begin
    ds = L.Dataset.synth(v_begin, v_end, n, SEED, MISSING_RATE)
    record_bytes = ds.info.record_bytes
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    dev = torch.device("cuda", local_rank)

    C = SimpleNamespace(np=np, torch=torch, L=L, sharding=sharding, ds=ds, dist=dist, dev=dev, stream=stream, st=st, n=n,
                        m=m, v_begin=v_begin, v_end=v_end, record_bytes=record_bytes, rank=rank, world=world,
                        pca_variants=0)
    W = build_workload(args, args.workload, C)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed = run_timed(W, args.steps, args.warmup, barrier)
    if dist is not None:
        elapsed = sharding.max_over_ranks(dist, elapsed, dev)

    verified = verify(args, args.workload, C, W.env)
    if dist is not None:
        ok = torch.tensor([0 if verified is False else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            verified = False
    if verified is False:
        raise SystemExit("bench.py: the last step's results failed verification; no line reported")

    kernel_ms = [a.elapsed_time(b) for a, b in W.kernel_events]
    kern_avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    if os.environ.get("PGH_BENCH_STEP_TIMES"):  # every timed launch, for a look at drift across the steps
        print("step kernel ms:", " ".join(f"{t:.1f}" for t in kernel_ms), file=sys.stderr, flush=True)
    total_units = sharding.total_variants(world, args.variants, args.scaling) * n
    if args.workload == "ld":
        total_units = W.units_per_step * (world if args.scaling == "weak" else 1)  # sample pairs, every rank the same shape
    value = total_units * args.steps / elapsed
    store_ceiling = store_ceiling_probe(C) if (args.workload == "unpack" and world == 1) else None
    roofline = roofline_of(args, W, kern_avg_ms, len(kernel_ms), m, n, store_ceiling)
    W_metric, W_dtype = W.metric, W.dtype

    # The default single-GPU run also times the other BASELINE configurations on the same resident matrix
    # (configs 2-5: read_pgen's unpack, the fused freq + hardy + missing pass, plink_score with 16 columns and with
    # one, plink_pca over the first 100,000 variants), each verified like its own --workload run, a few steps each.
    configs = None
    sql_expect = None
    if args.workload in ("freq", "fused") and m:
        hc = W.env["h_counts"].numpy().astype(np.int64).sum(axis=0)
        sql_expect = {"hom_ref": int(hc[0]), "het": int(hc[1]), "hom_alt": int(hc[2]), "missing": int(hc[3])}
    want_all = args.configs == "all" or (args.configs == "auto" and args.workload == "freq" and world == 1
                                         and args.variants == 1_000_000 and n == 500_000)
    if want_all and rank == 0:
        configs = {}
        primary_env = W.env
        del W
        ceiling = store_ceiling_probe(C)
        plan = [("fused", "fused", {}), ("unpack", "unpack", {}), ("score16", "score", {"score_cols": 16}),
                ("score1", "score", {"score_cols": 1}), ("pca", "pca", {})]
        for name, wl, kw in plan:
            C.pca_variants = 100_000 if wl == "pca" else 0
            steps = 2 if wl == "pca" else args.config_steps
            Wc = build_workload(args, wl, C, **kw)
            el = run_timed(Wc, steps, 1, barrier)
            ok = verify(args, wl, C, Wc.env)
            if ok is False:
                raise SystemExit(f"bench.py: config {name}: the last step's results failed verification; no line reported")
            kms = [a.elapsed_time(b) for a, b in Wc.kernel_events]
            kavg = float(np.mean(kms)) if kms else float("nan")
            rf = roofline_of(args, Wc, kavg, len(kms), m, n, ceiling)
            configs[name] = {"metric": Wc.metric, "ms_per_step": el / steps * 1e3, "steps": steps,
                             "value": Wc.units_per_step * steps / el, "unit": "genotypes/s", "kernel": rf["kernel"],
                             "kernel_ms_avg": kavg, "bound": rf["bound"], "frac": rf["frac"], "verified": ok,
                             "roofline": rf}
            del Wc
            torch.cuda.empty_cache()
        del primary_env

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(ds, n, args.cpu_seconds, args.cpu_sample_variants)

    sql = None
    if want_all and rank == 0 and args.sql and sql_expect is not None:
        # through the SQL shells, over a resident source of the same shape (the bench's own matrix goes first: two
        # of them do not fit one GPU)
        ds.close()
        torch.cuda.empty_cache()
        sql = sql_section(args, n, sql_expect)

    if rank == 0:
        line = {
            "metric": W_metric,
            "value": value,
            "unit": "sample pairs/s" if args.workload == "ld" else "genotypes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": W_dtype,
            "data": "synthetic",
            "verified": verified,
            "verification": "recount of 258 random rows copied back from HBM by the oracle's multi-threaded scan, bit-exact"
                            if args.workload in ("freq", "fused") else "see bench.py:verify",
            "config": {
                "workload": f"{args.workload}: {args.variants} variants x {n} samples "
                            f"({'per GPU' if args.scaling == 'weak' else 'total'}), 2-bit hardcalls resident in HBM, "
                            f"seed {SEED}, {MISSING_RATE:.0%} missing"
                            + (f", dosage tracks on every variant with {args.dosage_rate:.0%} of samples explicit"
                               if args.workload.startswith("dosage") else "")
                            + (" [ONE-GPU REHEARSAL: all ranks on GPU 0 over gloo; not a measurement]" if rehearsal else ""),
                "variants_per_rank": m,
                "samples": n,
                "record_bytes": record_bytes,
                "sharding": "contiguous variant ranges per GPU, no data-path collective"
                            if args.workload != "score" else "variant shards + RCCL reduce of per-sample partials",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if configs is not None:
            line["configs"] = configs
        if sql is not None:
            line["sql"] = sql
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
