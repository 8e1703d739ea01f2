"""rocprofv3 --kernel-trace CSV -> the device timeline of one training step, averaged over the last steps of the run.

    python tools/timeline.py <dir with *kernel_trace.csv> [n_steps] [delimiter kernel prefix] > timeline.txt

A step is delimited by the table optimizer (nsx::adam_hash_factored_kernel; a data-parallel rank's shard optimizer:
pass "nsx::adam_f16grad"): step k = (end of Adam k-1, end of Adam k].
Per kernel position inside the step: mean start offset from the step start, mean duration, mean idle time of the device
in front of it (no kernel of any queue running), whether it overlapped another kernel.  The totals at the end split the
step period into "some kernel running" and "device idle" -- the idle part is dependent-launch latency, not work.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.match(r"_ZN3nsx(\d+)", name)
    if m:                                                    # some kernels arrive mangled
        return "nsx::" + name[m.end():m.end() + int(m.group(1))]
    name = name.replace("void ", "")
    base = name.split("(")[0]
    if len(base) > 70:
        base = base[:67] + "..."
    return base


def main():
    directory = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                             r.get("Queue_Id", "?")))
    # copies made by the DMA engines (host <-> device, sometimes device <-> device) are not kernels but the stream waits for
    # them all the same: with `--memory-copy-trace` in the same run they appear in the timeline as "memcpy <direction>"
    for path in glob.glob(os.path.join(directory, "**", "*memory_copy_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                             "memcpy " + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "dma"))
    rows.sort()
    delimiter = sys.argv[3] if len(sys.argv) > 3 else "nsx::adam_hash_factored_kernel"
    adam_ends = [e for s, e, n, q in rows if n.startswith(delimiter)]
    if len(adam_ends) < n_steps + 1:
        print(f"only {len(adam_ends)} table-optimizer launches in the trace")
        return
    bounds = adam_ends[-(n_steps + 1):]
    per_pos = collections.defaultdict(list)
    busy_total = idle_total = 0
    seqs = []
    step_rows = []
    for k in range(n_steps):
        t0, t1 = bounds[k], bounds[k + 1]
        ks = [r for r in rows if t0 < r[1] <= t1 and r[0] >= t0 - 5_000_000]
        ks.sort()
        seqs.append(tuple(n for _, _, n, _ in ks))
        horizon = t0                                        # the latest end of anything seen so far
        busy = 0
        step_rows.append([])
        for i, (s, e, n, q) in enumerate(ks):
            idle_before = max(0, s - horizon)
            overlapped = s < horizon
            per_pos[i].append((n, q, max(s, t0) - t0, e - s, idle_before, overlapped))
            step_rows[-1].append(per_pos[i][-1])
            if e > horizon:
                busy += e - max(s, horizon)
                horizon = e
        busy_total += busy
        idle_total += (t1 - t0) - busy
    same = len(set(seqs)) == 1
    print(f"steps analysed: {n_steps}; kernels per step: {sorted(set(len(s) for s in seqs))}; identical sequence: {same}")
    print(f"mean step period {1e-6 * (bounds[-1] - bounds[0]) / n_steps:.3f} ms = busy "
          f"{1e-6 * busy_total / n_steps:.3f} ms + device idle {1e-6 * idle_total / n_steps:.3f} ms")
    # the per-position table: steps whose kernel sequence is the most common one only (an occupancy-grid update every 16th
    # step has ~40 kernels more; folding it in position by position would average unrelated kernels).  The totals above and
    # the per-name table below cover all analysed steps.
    modal = collections.Counter(seqs).most_common(1)[0][0]
    members = [k for k, q in enumerate(seqs) if q == modal]
    print(f"table: mean over the {len(members)} step(s) with the most common sequence ({len(modal)} kernels)")
    print(f"{'#':>3} {'start us':>9} {'dur us':>8} {'idle us':>8} ovl q   kernel")
    for i in range(len(modal)):
        v = [step_rows[k][i] for k in members]
        names = collections.Counter(x[0] for x in v).most_common(1)[0][0]
        mean = lambda j: sum(x[j] for x in v) / len(v)
        print(f"{i:3d} {1e-3 * mean(2):9.1f} {1e-3 * mean(3):8.1f} {1e-3 * mean(4):8.1f} "
              f"{'*' if sum(x[5] for x in v) * 2 > len(v) else ' ':>3} {v[0][1]:<3} {names}")
    by_name = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for i, v in per_pos.items():
        for n, q, s, d, idle, o in v:
            by_name[n][0] += 1
            by_name[n][1] += d
            by_name[n][2] += idle
    print("\nper kernel name and step: launches, total us, idle us in front")
    for n, (c, d, idle) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
        print(f"{c / n_steps:6.1f} {1e-3 * d / n_steps:9.1f} {1e-3 * idle / n_steps:9.1f}  {n}")


if __name__ == "__main__":
    main()
