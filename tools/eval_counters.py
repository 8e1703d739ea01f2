"""rocprofv3 --pmc counter_collection CSVs of tools/run_f_r06.sh -> what bounds nsx::density_fused_kernel in the evaluation image.

    python tools/eval_counters.py <dir with pass*/> <out.json>

Sums every counter over the dispatches of the kernel in ONE render (the last one of the run: 4 ray bundles), and derives the
rates that name the bound: L1 (TCP) tag lookups per CU and clock, L1 / L2 hit rates, the share of TA cycles stalled on the
cache, and the waves' issue-stall share."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, out = sys.argv[1:3]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if "density_fused_kernel" not in r["Kernel_Name"]:
                    continue
                per[r["Counter_Name"]][int(r.get("Dispatch_Id", 0) or 0)].append(float(r["Counter_Value"]))
    doc = {"kernel": "nsx::density_fused_kernel", "source": "tools/run_f_r06.sh: rocprofv3 --pmc (counters only, separate passes) of "
           "tools/eval_bench.py; the dispatches of the last render (25.9 M samples, 401 x 275 rays)", "counters": {}}
    for name, disp in per.items():
        ids = sorted(disp)
        n_last = max(1, len(ids) // 6)            # 6 renders per run (1 warm + 1 timed x 3 configurations that take this kernel)
        last = ids[-n_last:]
        doc["counters"][name] = {"sum_last_render": sum(sum(disp[i]) for i in last), "dispatches": len(last)}
    c = {k: v["sum_last_render"] for k, v in doc["counters"].items()}
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        d["gpu_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8.0
    cyc = d.get("gpu_cycles_per_xcd")
    for tag in ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_READ_sum"):
        if tag in c and cyc:
            d[tag + "_per_cu_per_clock"] = c[tag] / 256.0 / cyc
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
        d["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    if "TA_TA_BUSY_sum" in c and cyc:
        d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / 256.0 / cyc
    for tag in ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum",
                "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"):
        if tag in c and cyc:
            d[tag + "_frac_of_cu_cycles"] = c[tag] / 256.0 / cyc
    if "SQ_WAVE_CYCLES" in c:
        for tag in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if tag in c:
                d[tag + "_frac_of_wave_cycles"] = c[tag] / max(c["SQ_WAVE_CYCLES"], 1.0)
    doc["derived"] = d
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
