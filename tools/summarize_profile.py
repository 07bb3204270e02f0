#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the tracked profiles/ summaries.

    python tools/summarize_profile.py r01 gpurun_out/prof_r01 gpurun_out/pmc_r01_fetch gpurun_out/pmc_r01_write \
        gpurun_out/pmc_r01_tcc gpurun_out/pmc_r01_sq

Traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are collected
in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream, so fetched bytes = 2 * FETCH_SIZE * 1024 (upper estimate for narrow accesses).
"""
import collections
import csv
import glob
import os
import sys


def rows(d, pat):
    f = glob.glob(os.path.join(d, "*", pat))
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    return name.replace("bsrnn::", "").replace("void ", "").split("(")[0]


def main():
    tag, prof, fetch, write, tcc, sq = sys.argv[1:7]
    extra = sys.argv[7:10]          # optional: the L1 -> L2 request pass, the LDS pass, the wave-cycle pass (tools/profile_round.sh)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = rows(prof, "*_kernel_stats.csv")
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
        for r in stats:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                        "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])

    def per_dispatch(d):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows(d, "*_counter_collection.csv"):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
        res = {}
        for k, cs in agg.items():
            res[k] = {}
            for c, vals in cs.items():
                per = collections.defaultdict(float)
                for did, v in vals:
                    per[did] += v
                res[k][c] = sum(per.values()) / max(1, len(per))
        return res

    f_, w_, t_, s_ = per_dispatch(fetch), per_dispatch(write), per_dispatch(tcc), per_dispatch(sq)
    dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in stats}
    lines = ["# rocprofv3 summary %s (MI355X, bench.py --steps 10 --warmup 2, R=64 x 8 s @ 16 kHz)" % tag, "",
             "Per launch (average over dispatches).  fetched MB = 2 x FETCH_SIZE KiB (gfx950 correction), written MB = WRITE_SIZE KiB.", "",
             "| kernel | avg us | fetched MB | written MB | L2 hit % | MFMA busy % (of GRBM_GUI_ACTIVE/8) |", "|---|---|---|---|---|---|"]
    for k in sorted(dur, key=lambda x: -dur[x]):
        if "rocclr" in k:
            continue
        fe = 2 * f_.get(k, {}).get("FETCH_SIZE", 0) * 1024 / 1e6
        wr = w_.get(k, {}).get("WRITE_SIZE", 0) * 1024 / 1e6
        hit, miss = t_.get(k, {}).get("TCC_HIT_sum", 0), t_.get(k, {}).get("TCC_MISS_sum", 0)
        gui, busy = s_.get(k, {}).get("GRBM_GUI_ACTIVE", 0), s_.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        lines.append("| %s | %.1f | %.1f | %.1f | %s | %s |" % (
            k, dur[k], fe, wr, "%.0f" % (100 * hit / (hit + miss)) if hit + miss else "-",
            "%.0f" % (100 * busy / 1024 / (gui / 8)) if gui else "-"))
    if len(extra) == 3:
        l_, d_, v_ = per_dispatch(extra[0]), per_dispatch(extra[1]), per_dispatch(extra[2])
        lines += ["", "What bounds the kernels (separate PMC passes; per launch).  L1 -> L2 read MB = TCP_TCC_READ_REQ x 128 B (one request per 128-B line:"
                  " calibrated on the fused chains, whose fragment streams are known byte for byte - 2.15 / 2.38 GB per launch), GB/s per CU = that / avg us / 256 CUs; L1 stalled on L2 = TCP_PENDING_STALL_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs);"
                  " LDS: bank-conflict cycles / LDS-active cycles, and the share of wave cycles spent waiting on LDS instructions; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", "",
                  "| kernel | L1->L2 read MB | L1->L2 write MB | L2->CU GB/s per CU | L1 stalled on L2 % | LDS conflict % of LDS cycles | wave cycles waiting on LDS % | wave cycles waiting (any) % |",
                  "|---|---|---|---|---|---|---|---|"]
        for k in sorted(dur, key=lambda x: -dur[x]):
            if "rocclr" in k or k not in l_:
                continue
            rd = l_[k].get("TCP_TCC_READ_REQ_sum", 0) * 128 / 1e6
            wrq = l_[k].get("TCP_TCC_WRITE_REQ_sum", 0) * 64 / 1e6
            gui = s_.get(k, {}).get("GRBM_GUI_ACTIVE", 0)
            stall = l_[k].get("TCP_PENDING_STALL_CYCLES_sum", 0)
            conf, act = d_.get(k, {}).get("SQ_LDS_BANK_CONFLICT", 0), d_.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0)
            wlds = d_.get(k, {}).get("SQ_WAIT_INST_LDS", 0)
            wc, wany = v_.get(k, {}).get("SQ_WAVE_CYCLES", 0), v_.get(k, {}).get("SQ_WAIT_ANY", 0)
            lines.append("| %s | %.0f | %.0f | %.1f | %s | %s | %s | %s |" % (
                k, rd, wrq, rd * 1e6 / (dur[k] * 1e-6) / 256 / 1e9,
                "%.0f" % (100 * stall / (gui * 256)) if gui else "-", "%.1f" % (100 * conf / act) if act else "-",
                "%.0f" % (100 * wlds / wc) if wc else "-", "%.0f" % (100 * wany / wc) if wc else "-"))
    open(os.path.join(out_dir, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    import json
    traffic = {k: {"fetched_bytes": 2 * f_.get(k, {}).get("FETCH_SIZE", 0) * 1024, "written_bytes": w_.get(k, {}).get("WRITE_SIZE", 0) * 1024,
                   "avg_us": dur[k]} for k in dur if "rocclr" not in k}
    json.dump(traffic, open(os.path.join(out_dir, "%s_traffic.json" % tag), "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
