"""Development aid: print the device timeline of the last full training step in a rocprofv3 --kernel-trace CSV
(start offset, duration, queue, kernel), with the idle gaps of the main queue.

    python tools/timeline.py OUT/**/_kernel_trace.csv [step_from_end=1]
"""
import csv
import sys


def short(name: str) -> str:
    name = name.split("(")[0].replace("void ", "")
    if "<" in name:
        head, _, tail = name.partition("<")
        name = head.split("::")[-1] + "<" + tail[:48]
    else:
        name = name.split("::")[-1]
    return name[:80]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
marks = [i for i, r in enumerate(rows) if "march_count_kernel" in r[3]]
lo, hi = marks[-1 - back], marks[-back]
t0 = rows[lo][0]
main_q = rows[lo][2]
last_end = {}
print(f"step = kernels {lo}..{hi}, {(rows[hi][0] - t0) / 1e6:.3f} ms; main queue {main_q}")
for s, e, q, n in rows[lo:hi]:
    gap = s - last_end.get(q, s)
    flag = f"  (+{gap / 1e3:.0f} us idle on q{q})" if gap > 20000 else ""
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{q}  {short(n)}{flag}")
    last_end[q] = max(last_end.get(q, 0), e)
