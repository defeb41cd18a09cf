#!/usr/bin/env python3
"""profiles/<tag>_bench_fetch.csv + <tag>_bench_write.csv (tools/rocpd_summary.py output of separate rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes) -> profiles/<tag>_pmc_traffic.json: HBM bytes per launch per kernel, keyed by the
kernel names bench.py reports (tap-count template argument folded), corrected per tools/pmc_calibrate.py
(gfx950: FETCH_SIZE counts 0.5 x bytes -> doubled; WRITE_SIZE exact; both in KiB).

    python tools/make_pmc_traffic.py r01d "command that was profiled" """
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(name: str) -> str:
    name = name.replace(" ", "").replace("parrot::", "")
    m = re.match(r"(conv_split_kernel<Sch\w+,\d+,\d+,\d+,\d+,\d+),\d+(,\d+)?(,(true|false))?>", name)
    if m:
        return m.group(1) + ">"
    m = re.match(r"(conv_split16_kernel<Sch\w+,\d+,\d+,\d+,\d+),\d+(,\d+)?(,(true|false))?>", name)  # (.., taps, min waves, plane input)
    if m:
        return m.group(1) + ">"
    m = re.match(r"resblock_split_kernel<(Sch\w+),(\d+),(\d+),(true|false)>", name)  # (scheme, chunks, column groups, whole-MRF)
    if m:
        return f"resblock_split_kernel<{m.group(1)},{m.group(2)},{m.group(3)},MRF>" if m.group(4) == "true" else f"resblock_split_kernel<{m.group(1)},{m.group(2)}>"
    if name == "conv1_valu7_vec_kernel":
        return "conv1_valu_kernel"
    return re.sub(r"^(conv1_valu_kernel)<\d+>$", r"\1", name)  # instantiations that share one bench.py row


def load(path: str, col: str):
    tot, calls = {}, {}
    for r in csv.DictReader(open(path)):
        if not r.get(col + "_sum"):
            continue
        k = norm(r["kernel"])
        tot[k] = tot.get(k, 0.0) + float(r[col + "_sum"])
        calls[k] = calls.get(k, 0) + int(r["calls"])
    return {k: tot[k] / calls[k] for k in tot}


def load_busy(path: str):
    """MFMA-busy fraction per kernel from the SQ pass: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), launches summed."""
    if not os.path.exists(path):
        return {}
    mf, cu = {}, {}
    for r in csv.DictReader(open(path)):
        if not r.get("SQ_VALU_MFMA_BUSY_CYCLES_sum") or not r.get("SQ_BUSY_CU_CYCLES_sum"):
            continue
        k = norm(r["kernel"])
        mf[k] = mf.get(k, 0.0) + float(r["SQ_VALU_MFMA_BUSY_CYCLES_sum"])
        cu[k] = cu.get(k, 0.0) + float(r["SQ_BUSY_CU_CYCLES_sum"])
    return {k: mf[k] / (4.0 * cu[k]) for k in mf if cu[k] > 0}


def main():
    tag, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    fetch = load(os.path.join(ROOT, "profiles", f"{tag}_bench_fetch.csv"), "FETCH_SIZE")
    write = load(os.path.join(ROOT, "profiles", f"{tag}_bench_write.csv"), "WRITE_SIZE")
    busy = load_busy(os.path.join(ROOT, "profiles", f"{tag}_bench_sq.csv"))
    box_file = os.path.join(ROOT, "profiles", f"{tag}_box.txt")
    box = open(box_file).read().strip() if os.path.exists(box_file) else None
    out = {"_comment": f"HBM bytes per launch, rocprofv3 PMC passes of `{cmd}`. FETCH_SIZE / WRITE_SIZE in separate passes "
                       f"(profiles/{tag}_bench_fetch.csv, {tag}_bench_write.csv; KiB per launch averaged over all launches of the kernel, "
                       "tap-count instantiations of one tile folded); FETCH_SIZE doubled per the gfx950 calibration "
                       "(profiles/r01_calibration_copy_*.csv: 0.500 x bytes), WRITE_SIZE as is (1.000 x bytes).  mfma_busy: "
                       f"SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) of the SQ pass of the same command (profiles/{tag}_bench_sq.csv).",
           "_box": box}
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if "kernel" not in k:
            continue
        w = write.get(k, 0.0)
        out[k] = {"fetch_kib_raw": round(fetch[k], 1), "write_kib_raw": round(w, 1),
                  "hbm_bytes_per_launch": int(round((2.0 * fetch[k] + w) * 1024)),
                  "mfma_busy": (round(busy[k], 4) if k in busy else None)}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
