"""gpurun_out/profiles/ (written by tools/collect_profiles.sh on the GPU box) -> profiles/rNN_*.

    python3 tools/summarise_profiles.py --round 1
"""
import argparse
import collections
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles")
DST = os.path.join(ROOT, "profiles")

# the kernels whose HBM traffic a workload's roofline line is about (a tuple: their per-launch means are added)
DOMINANT = {"freq": ("k_counts_block",), "fused": ("k_fused_tally",), "unpack": ("k_unpack_wide",),
            "score1": ("k_score_i8",), "score": ("k_score_i8",),
            "dosagescore": ("k_score_i8", "k_score_dosage_records")}


def short(name):
    m = re.search(r"k_\w+(<[^>]*>)?", name)
    return m.group(0) if m else name.split("(")[0]


def pmc_means(path):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), r["Counter_Name"])
        s = acc.setdefault(key, [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=3)
    ap.add_argument("src", nargs="?", default=None, help="directory the collection left its files in (default: gpurun_out/profiles)")
    args = ap.parse_args()
    global SRC
    if args.src:
        SRC = args.src
    tag = f"r{args.round:02d}"
    for name in ("freq", "fused", "unpack", "score", "score1", "score2", "score4", "score8", "pca", "ld", "samplecounts", "missingsample", "dosagefreq", "dosagescore", "dosagefull", "dosagegaps"):
        src = os.path.join(SRC, f"bench_{name}.json")
        if os.path.exists(src):
            line = [ln for ln in open(src).read().splitlines() if ln.startswith("{")][-1]
            json.loads(line)
            with open(os.path.join(DST, f"{tag}_bench_{name}_n1.json"), "w") as f:
                f.write(line + "\n")
    for name in ("freq", "fused", "unpack", "score", "score1", "pca", "ld", "samplecounts", "missingsample", "dosagefreq", "dosagescore", "dosagefull"):
        src = os.path.join(SRC, f"{name}_kernel_stats.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, f"{tag}_{name}_kernel_stats.csv"))
    traffic = {
        "_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh) over "
                   "`python3 bench.py [--workload W] --steps 3 --warmup 1 --cpu-seconds 0`; FETCH_SIZE (KiB) doubled as "
                   "MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950, WRITE_SIZE (KiB) as is; mean "
                   "over the dominant kernel's dispatches. Per-counter means: profiles/rNN_*_pmc_*.csv "
                   "(tools/summarise_profiles.py).",
    }
    for name, kernels in DOMINANT.items():
        raw = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            src = os.path.join(SRC, f"{name}_pmc_{ctr}.csv")
            if not os.path.exists(src):
                continue
            means = pmc_means(src)
            with open(os.path.join(DST, f"{tag}_{name}_pmc_{ctr}.csv"), "w") as f:
                f.write("kernel,counter,dispatches,mean_value,unit\n")
                for (k, c), (n, total) in means.items():
                    f.write(f"\"{k}\",{c},{n},{total / n:.3f},KiB\n")
                    if any(k.startswith(kernel) for kernel in kernels):
                        seen = raw.get(ctr, (n, 0.0))
                        raw[ctr] = (seen[0], seen[1] + total / n)
        if len(raw) == 2:
            bench = json.loads(open(os.path.join(DST, f"{tag}_bench_{name}_n1.json")).read())
            traffic[name] = {
                "kernel": " + ".join(kernels),
                "variants": bench["config"]["variants_per_rank"],
                "samples": bench["config"]["samples"],
                "fetch_size_kib_raw": raw["FETCH_SIZE"][1],
                "write_size_kib_raw": raw["WRITE_SIZE"][1],
                "dispatches": raw["FETCH_SIZE"][0],
                "hbm_bytes_per_launch": (2 * raw["FETCH_SIZE"][1] + raw["WRITE_SIZE"][1]) * 1024,
                "algorithmic_bytes_per_launch": bench["roofline"].get(
                    "algorithmic_bytes_per_launch",
                    bench["config"]["variants_per_rank"] * bench["config"]["record_bytes"]),  # (a matrix-bound line: the rows)
            }
    # matrix-core counters of the int8 contraction
    for name in ("score", "score1", "pca"):
        src = os.path.join(SRC, f"{name}_pmc_mfma.csv")
        if not os.path.exists(src):
            continue
        means = pmc_means(src)
        with open(os.path.join(DST, f"{tag}_{name}_pmc_mfma.csv"), "w") as f:
            f.write("kernel,counter,dispatches,sum_value,mean_value\n")
            for (k, c), (n, total) in means.items():
                if k.startswith("k_score_i8"):
                    f.write(f"\"{k}\",{c},{n},{total:.6g},{total / n:.6g}\n")
    # text outputs: the full-size shell bench, the tally kernels alone, and the per-call kernel list of
    # tools/hardy_after_freq.py's trace (calls separated by torch's marker kernel)
    for name in ("shell_bench", "tally_kernels_alone"):
        src = os.path.join(SRC, f"{name}.txt")
        if os.path.exists(src):
            with open(os.path.join(DST, f"{tag}_{name}.txt"), "w") as f:
                f.writelines(ln for ln in open(src) if "amdgpu.ids" not in ln)
    trace, phases = os.path.join(SRC, "hardy_after_freq_kernel_trace.csv"), os.path.join(SRC, "hardy_after_freq.txt")
    if os.path.exists(trace) and os.path.exists(phases):
        rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
        calls, cur = [], None
        for r in rows:
            if "at::native" in r["Kernel_Name"]:
                if "add" in r["Kernel_Name"]:  # marker.add_(1.0): a call begins (the fill is the marker's creation)
                    cur = collections.OrderedDict()
                    calls.append(cur)
                continue
            if cur is not None:
                e = cur.setdefault(short(r["Kernel_Name"]).split("<")[0], [0, 0])
                e[0] += 1
                e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        titles = ["plink_freq (the first call on the source: generator + ONE tally pass)", "plink_hardy", "plink_missing",
                  "plink_missing mode := 'sample'"]
        with open(os.path.join(DST, f"{tag}_hardy_after_freq_kernels.txt"), "w") as f:
            f.write("# rocprofv3 --kernel-trace over tools/hardy_after_freq.py (synth:1000000x500000:20260807:0.02, 16 scan threads, "
                    "chunks drained):\n# the kernels each table-function call launched; the calls are separated in the trace by a "
                    "marker kernel of torch's.\n# plink_freq pays for the one walk of the matrix -- k_fused_tally per 131,072-variant "
                    "batch (16.4 GB) with its two small\n# epilogues -- plink_hardy launches the exact tests over the pass's resident "
                    "counts and NOTHING that reads the matrix,\n# plink_missing in either mode launches no kernel at all (tally passes "
                    "started by the process: 1 throughout).\n\n")
            f.writelines(ln for ln in open(phases) if "amdgpu.ids" not in ln)
            for title, call in zip(titles, calls):
                f.write(f"\n## {title}\n")
                for k, (n, ns) in sorted(call.items(), key=lambda kv: -kv[1][1]):
                    f.write(f"   {k:36s} {n:3d} launches {ns / 1e6:10.3f} ms in all\n")
                if not call:
                    f.write("   (no kernel launched)\n")
    with open(os.path.join(DST, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
        f.write("\n")
    for name, e in traffic.items():
        if isinstance(e, dict):
            print(name, e["kernel"], f"traffic/algorithmic = {e['hbm_bytes_per_launch'] / e['algorithmic_bytes_per_launch']:.4f}")


if __name__ == "__main__":
    main()
