"""rocprofv3 --pmc counter_collection CSVs -> per-kernel means and per-launch HBM bytes of the C-ABI entry points.

    python tools/pmc_to_json.py <fetch_dir> <write_dir> <out.json> "<source note>"

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  Correction (MI355X_MICROARCH.md, HBM section): on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> x2; WRITE_SIZE as reported."""
import collections
import csv
import glob
import json
import os
import re
import sys

ENTRY_OF = {            # kernel-name prefix -> C-ABI entry point it belongs to
    "nsx::ens_fwd_kernel": "nsx_hash_ensemble_fwd",
    "nsx::ens_bwd_kernel": "nsx_hash_ensemble_bwd_factored",
    "nsx::ens_scatter_kernel": "nsx_hash_ensemble_bwd_scatter",
    "nsx::ens_fwd_sources_kernel": "nsx_lp_fwd_run (all source ranks in one launch)",
    "nsx::ens_bwd_sources_kernel": "nsx_lp_bwd_run (all source ranks in one launch)",
    "nsx::adam_hash_factored_kernel": "nsx_adam_hash_factored",
    "nsx::adam_hash_factored_mfma_kernel": "nsx_adam_hash_factored (> 64 planes: matrix-core expansion)",
    "nsx::deform_bwd_kernel": "nsx_deform_bwd",
    "nsx::deform_wgrad_kernel": "nsx_deform_bwd",
    "nsx::deform_fwd_kernel": "nsx_deform_fwd",
    "nsx::deform_fwd_terms_kernel": "nsx_deform_fwd_rows",
    "nsx::density_fused_kernel": "nsx_density_fused_fwd",
}


def kernel_base_name(raw):
    """'void nsx::k<32, 4>(args)' / 'nsx::k(args)' / the mangled '_ZN3nsx<len><name>I...' -> 'nsx::k'."""
    m = re.match(r"_ZN3nsx(\d+)", raw)
    if m:
        n = int(m.group(1))
        return "nsx::" + raw[m.end():m.end() + n]
    return raw.split("(")[0].replace("void ", "").split("<")[0].strip()


LAST = int(os.environ.get("PMC_LAST_DISPATCHES", "0"))      # > 0: only the last N dispatches of every kernel (a run that
#                                                              settles first: the pre-roll's dispatches are another state)


def means(directory, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                name = kernel_base_name(r["Kernel_Name"])
                acc[name].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    out = {}
    for k, rows in acc.items():
        if LAST > 0:
            keep = set(sorted({d for d, _ in rows})[-LAST:])
            rows = [r for r in rows if r[0] in keep]
        v = [x for _, x in rows]
        out[k] = {"mean_kb": sum(v) / len(v), "dispatches": len({d for d, _ in rows})}
    return out


def main():
    fetch_dir, write_dir, out, note = sys.argv[1:5]
    fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    per_launch = collections.defaultdict(float)
    for name, entry in ENTRY_OF.items():
        if name in fetch:
            per_launch[entry] += fetch[name]["mean_kb"] * 1024.0 * 2.0
        if name in write:
            per_launch[entry] += write[name]["mean_kb"] * 1024.0
    doc = {"source": note,
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts half the bytes of 16-B/lane coalesced reads on "
                         "gfx950 -> x2; WRITE_SIZE as reported; KB -> x1024; per launch = mean over the dispatches of the "
                         "run (a C-ABI call that launches two kernels sums them)",
           "per_launch_hbm_bytes": dict(per_launch),
           "kernels_fetch_kb": {k: v for k, v in sorted(fetch.items()) if k.startswith("nsx::")},
           "kernels_write_kb": {k: v for k, v in sorted(write.items()) if k.startswith("nsx::")}}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["per_launch_hbm_bytes"], indent=1))


if __name__ == "__main__":
    main()
