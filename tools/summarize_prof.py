#!/usr/bin/env python3
"""Summarises one rocprofv3 run directory written by tools/profile.sh into a
small JSON + markdown pair for profiles/ (per-kernel time from the kernel
trace, per-dispatch means of the counter passes, and the figures derived from
them as MI355X_MICROARCH.md prescribes: SQ_* cycle counters are quad-cycles,
FETCH_SIZE under-reports wide reads by 2x, GRBM_GUI_ACTIVE is summed over 8 XCDs).

    python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name> [kernel-substring] [workload]

With a workload name (c2, c3, c4, c5) it also writes profiles/traffic_<workload>.json, the measured HBM
bytes per launch that bench.py reports as roofline.traffic for that workload.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else "renderKernel"
    out = {"source": src, "kernel_filter": want}
    stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows[:6]]
        for r in rows:
            if want in r["Name"]:
                out["kernel"] = r["Name"]
                out["avg_ns"] = float(r["AverageNs"])
                out["calls"] = int(r["Calls"])
    trace = glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))
    if trace:
        for r in csv.DictReader(open(trace[0])):
            if want in r["Kernel_Name"]:
                out["launch"] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
                break
    counters = {}
    for path in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if want in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            counters[k] = sum(v) / len(v)
    out["counters_per_dispatch"] = counters
    d = {}
    c = counters
    if "avg_ns" in out:
        t = out["avg_ns"] * 1e-9
        if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
            d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
            d["hbm_read_bytes_corrected_x2"] = c["FETCH_SIZE"] * 1024 * 2
            d["hbm_bytes"] = d["hbm_write_bytes"] + d["hbm_read_bytes_corrected_x2"]
            d["hbm_GBps"] = d["hbm_bytes"] / t / 1e9
            d["hbm_frac_of_8TBps"] = d["hbm_GBps"] / 8000.0
        if "GRBM_GUI_ACTIVE" in c:
            d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
        if "SQ_WAVES" in c and "SQ_WAVE_CYCLES" in c:
            d["wave_lifetime_cycles"] = 4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
            clk = d.get("effective_clock_GHz", 2.4) * 1e9
            d["mean_resident_waves_per_simd"] = 4 * c["SQ_WAVE_CYCLES"] / (t * clk) / 1024
            d["occupancy_frac_of_8_waves_per_simd"] = d["mean_resident_waves_per_simd"] / 8
        if "SQ_ACTIVE_INST_VALU" in c:
            clk = d.get("effective_clock_GHz", 2.4) * 1e9
            d["valu_busy_frac_of_simd_time"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (t * clk * 1024)
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            d["wave_time_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)
    out["derived"] = d
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".md", "w") as f:
        f.write(f"# rocprofv3 summary: {os.path.basename(dst)}\n\nsource run: `{src}` (tools/profile.sh), kernel filter `{want}`\n\n")
        if "avg_ns" in out:
            f.write(f"* kernel `{out['kernel']}`: {out['calls']} calls, average {out['avg_ns'] / 1e3:.1f} us\n")
        if "launch" in out:
            f.write(f"* launch: {out['launch']}\n")
        f.write("\n| derived figure | value |\n|---|---|\n")
        for k, v in d.items():
            f.write(f"| {k} | {v:.6g} |\n")
        f.write("\n| counter (mean per dispatch) | value |\n|---|---|\n")
        for k, v in sorted(counters.items()):
            f.write(f"| {k} | {v:.6g} |\n")
    if len(sys.argv) > 4 and "hbm_bytes" in d:
        with open(os.path.join(os.path.dirname(dst) or ".", "traffic_%s.json" % sys.argv[4]), "w") as f:
            json.dump({"workload": sys.argv[4], "hbm_bytes_per_launch": d["hbm_bytes"], "hbm_write_bytes": d["hbm_write_bytes"],
                       "hbm_read_bytes": d["hbm_read_bytes_corrected_x2"], "kernel": out.get("kernel"), "kernel_avg_us": out["avg_ns"] / 1e3,
                       "source": os.path.basename(dst) + ".json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as "
                                 "prescribed for 16-byte-per-lane reads on gfx950)"}, f, indent=1)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
