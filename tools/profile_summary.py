#!/usr/bin/env python3
"""Summarise rocprofv3 output (kernel stats + PMC passes) into profiles/<tag>_summary.json/.md.

  python tools/profile_summary.py --tag r01 --stats DIR --fetch DIR --write DIR [--sq DIR]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB
per dispatch, collected in separate passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
16-B-per-lane loads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
"""
import argparse
import collections
import csv
import glob
import json
import os


def read_counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not d:
        return out
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--dominant", default="", help="substring of the dominant kernel name for traffic.json; default: "
                                                   "read roofline.kernel from --bench-line (bench.py's JSON line)")
    ap.add_argument("--bench-line", default="", help="file holding bench.py's JSON line of the profiled command")
    a = ap.parse_args()
    summary = {"kernels": {}}
    if a.stats:
        for f in glob.glob(os.path.join(a.stats, "*kernel_stats.csv")):
            for r in csv.DictReader(open(f)):
                summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                 "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                                 "pct": float(r["Percentage"])}
    fetch, write, sq = read_counters(a.fetch), read_counters(a.write), read_counters(a.sq)
    for k in set(fetch) | set(write) | set(sq):
        e = summary["kernels"].setdefault(k, {})
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            v = fetch[k]["FETCH_SIZE"]
            e["fetch_kib_raw_avg"] = sum(v) / len(v)
            e["hbm_read_bytes_avg"] = 2.0 * 1024 * sum(v) / len(v)      # gfx950 correction (x2)
        if k in write and "WRITE_SIZE" in write[k]:
            v = write[k]["WRITE_SIZE"]
            e["hbm_write_bytes_avg"] = 1024 * sum(v) / len(v)
        if k in sq:
            c = {n: sum(v) / len(v) for n, v in sq[k].items()}
            e["pmc_avg"] = c
            if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    os.makedirs(a.out, exist_ok=True)
    if not a.dominant and a.bench_line and os.path.exists(a.bench_line):
        with open(a.bench_line) as f:
            for line in f:
                if line.startswith("{"):
                    a.dominant = json.loads(line)["roofline"]["kernel"].split(" (")[0]
    if a.dominant:
        for k, e in summary["kernels"].items():
            if a.dominant in k and "hbm_read_bytes_avg" in e and "hbm_write_bytes_avg" in e:
                with open(os.path.join(a.out, "traffic.json"), "w") as f:
                    json.dump({"kernel": k, "hbm_bytes_per_launch": e["hbm_read_bytes_avg"] + e["hbm_write_bytes_avg"],
                               "hbm_read_bytes_per_launch": e["hbm_read_bytes_avg"],
                               "hbm_write_bytes_per_launch": e["hbm_write_bytes_avg"],
                               "avg_launch_ms": e.get("avg_ms"), "calls": e.get("calls"),
                               "mfma_busy_frac": e.get("mfma_busy_frac"), "lds_conflict_frac": e.get("lds_conflict_frac"),
                               "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tag {a.tag}; "
                                         "read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"},
                              f, indent=1)
    # the side legs of bench.py's line (training step, bf16 tier): counters of the kernel with the most time in each
    legs = {}
    for leg, pats in (("train", ("wgrad3x3_x3_ws_kernel", "bn_bwd_apply_kernel")),
                      ("bf16", ("conv3x3_bf16_r512_kernel", "conv3x3_bf16_ws_kernel", "igemm_bf16_kernel"))):
        cands = [(e.get("total_ms", 0.0), k, e) for k, e in summary["kernels"].items()
                 if any(p in k for p in pats) and "hbm_read_bytes_avg" in e and "hbm_write_bytes_avg" in e]
        if cands:
            _, k, e = max(cands)
            legs[leg] = {"kernel": k, "hbm_bytes_per_launch": e["hbm_read_bytes_avg"] + e["hbm_write_bytes_avg"],
                         "avg_launch_ms": e.get("avg_ms"), "calls": e.get("calls"),
                         "hbm_tb_per_s": (e["hbm_read_bytes_avg"] + e["hbm_write_bytes_avg"]) / (e["avg_ms"] * 1e-3) / 1e12
                         if e.get("avg_ms") else None,
                         "mfma_busy_frac": e.get("mfma_busy_frac"), "lds_conflict_frac": e.get("lds_conflict_frac"),
                         "source": f"rocprofv3 --pmc passes, tag {a.tag} (read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB)"}
    if legs:
        with open(os.path.join(a.out, "traffic_legs.json"), "w") as f:
            json.dump(legs, f, indent=1)
    with open(os.path.join(a.out, f"{a.tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    lines = [f"# rocprofv3 summary {a.tag}", "",
             "| kernel | calls | avg ms | % | HBM read MB/launch (2xFETCH) | HBM write MB/launch | MFMA busy | LDS conflict |",
             "|---|---|---|---|---|---|---|---|"]
    for k, e in sorted(summary["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0)):
        lines.append("| `{}` | {} | {:.3f} | {:.2f} | {} | {} | {} | {} |".format(
            k[:90], e.get("calls", ""), e.get("avg_ms", 0), e.get("pct", 0),
            f"{e['hbm_read_bytes_avg'] / 1e6:.1f}" if "hbm_read_bytes_avg" in e else "",
            f"{e['hbm_write_bytes_avg'] / 1e6:.1f}" if "hbm_write_bytes_avg" in e else "",
            f"{e['mfma_busy_frac']:.3f}" if "mfma_busy_frac" in e else "",
            f"{e['lds_conflict_frac']:.3f}" if "lds_conflict_frac" in e else ""))
    with open(os.path.join(a.out, f"{a.tag}_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
