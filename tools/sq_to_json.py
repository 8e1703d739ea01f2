"""rocprofv3 --pmc counter_collection CSVs of tools/sq_counters.sh -> per-kernel counter means + derived figures.

    python tools/sq_to_json.py <dir with passA/ passB/ and mfma_bench.json> <out.json>

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over the SIMDs that report), SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_BUSY_CYCLES per SE.  Derived:
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
                     (the gfx94x MfmaUtil formula; GRBM_GUI_ACTIVE arrives summed over the 8 XCDs: deform_fwd 5.26 M "cycles"
                     in 0.34 ms would be 15.5 GHz, /8 = 1.93 GHz -- the clock under the profiler; the busy count is 32 cycles
                     x the number of 32x32x16 MFMAs = 1/peak issue rate, so this fraction x clock/2.4 GHz = fraction of peak)
  wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES        (waves parked on s_waitcnt / barriers)
  wait_inst_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls)
  active_frac      = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
Only the LAST dispatch of a kernel name is used (the warm-up launch before it pages code in)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def base_name(raw):
    m = re.match(r"_ZN3nsx(\d+)", raw)
    if m:
        n = int(m.group(1))
        name = "nsx::" + raw[m.end():m.end() + n]
        t = re.match(r"IL[ib](\d+)E", raw[m.end() + n:])          # first integer / bool template argument: mlp_fwd_kernel<1>
        return name + (f"<{t.group(1)}>" if t else "")
    return raw.split("(")[0].replace("void ", "").strip()


def load(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = base_name(r["Kernel_Name"])
                if not name.startswith("nsx::"):
                    continue
                acc[name][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    return acc


def main():
    root, out = sys.argv[1:3]
    merged = collections.defaultdict(dict)
    for sub in ("passA", "passB"):
        for name, counters in load(os.path.join(root, sub)).items():
            for c, vals in counters.items():
                vals.sort()
                # the counters of one dispatch may arrive as several rows (per dimension): sum rows of the last dispatch
                last = vals[-1][0]
                merged[name][c] = sum(v for d, v in vals if d == last)
                merged[name].setdefault("_dispatches", len({d for d, _ in vals}))
    doc = {"source": "tools/sq_counters.sh: rocprofv3 --pmc (two passes, counters only) of tools/mfma_bench.py at S = 2^20",
           "kernels": {}}
    try:
        doc["hip_event_times"] = json.load(open(os.path.join(root, "mfma_bench.json")))
    except Exception as e:            # noqa: BLE001
        doc["hip_event_times"] = f"unavailable: {e}"
    for name, c in sorted(merged.items()):
        d = {k: v for k, v in c.items()}
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for key, col in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_frac", "SQ_WAIT_INST_ANY"),
                             ("active_frac", "SQ_ACTIVE_INST_ANY"), ("wait_inst_lds_frac", "SQ_WAIT_INST_LDS")):
                if col in c:
                    d[key] = round(c[col] / wc, 4)
        if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4), 4)
            d["kernel_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8.0
        if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
            d["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c:
            d["valu_per_mfma"] = round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 3)
        doc["kernels"][name] = d
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for name, d in doc["kernels"].items():
        print(name, {k: v for k, v in d.items() if k.endswith("_frac") or k.endswith("_per_mfma") or k in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                                                                                "SQ_INSTS_VALU", "SQ_BUSY_CYCLES")})


if __name__ == "__main__":
    main()
