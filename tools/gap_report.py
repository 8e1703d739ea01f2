"""Aggregate GPU idle gaps from a rocprofv3 --kernel-trace CSV: which kernel boundary the device waits at.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py ...
    python tools/gap_report.py OUT/**/**_kernel_trace.csv [skip_first_n_kernels]
"""
import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.split("(")[0]
    if "<" in name:
        head, _, tail = name.partition("<")
        name = head.split("::")[-1] + "<" + tail[:40]
    else:
        name = name.split("::")[-1]
    return name[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = rows[int(len(rows) * frac):]                  # the tail of the run = the timed region
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = defaultdict(lambda: [0, 0])
    kern = defaultdict(lambda: [0, 0])
    last_end = rows[0][1]
    for i in range(1, len(rows)):
        s, e, n = rows[i]
        g = s - last_end
        if g > 0:
            k = (short(rows[i - 1][2]), short(n))
            gaps[k][0] += g
            gaps[k][1] += 1
        last_end = max(last_end, e)
    for s, e, n in rows:
        kern[short(n)][0] += e - s
        kern[short(n)][1] += 1
    print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
    print("--- top gaps (prev -> next): total ms, count, avg us")
    for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{t/1e6:8.3f} {c:6d} {t/c/1e3:8.1f}  {k[0]}  ->  {k[1]}")
    print("--- top kernels: total ms, count, avg us")
    for k, (t, c) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"{t/1e6:8.3f} {c:6d} {t/c/1e3:8.1f}  {k}")


if __name__ == "__main__":
    main()
