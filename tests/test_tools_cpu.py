"""The evidence tools that turn rocprofv3 output into the tables under profiles/ (CPU, synthetic traces)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_trace(directory, steps):
    """steps: list of lists of (name, queue, start_us, dur_us) relative to the step's start; steps are 10 ms apart."""
    path = os.path.join(directory, "x_kernel_trace.csv")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name", "Queue_Id"])
        for k, rows in enumerate(steps):
            base = k * 10_000_000
            for name, queue, start, dur in rows:
                w.writerow([base + int(start * 1000), base + int((start + dur) * 1000), name, queue])


def test_timeline_folds_only_the_common_kernel_sequence(tmp_path):
    """A step with extra kernels (an occupancy-grid update) must not shift the per-position means of the others; device
    idle time is what no queue covers; a copy from the memory-copy trace shows up as a row."""
    adam = "nsx::adam_hash_factored_kernel(float*)"
    normal = [("void nsx::a_kernel(float*)", 1, 10, 100), ("void nsx::b_kernel(float*)", 1, 120, 50),
              ("void nsx::side_kernel(int)", 2, 130, 20), (adam, 3, 200, 1000)]
    update = [("void nsx::a_kernel(float*)", 1, 10, 100), ("void nsx::occ_update(int)", 1, 111, 500),
              ("void nsx::b_kernel(float*)", 1, 620, 50), (adam, 3, 700, 1000)]
    steps = [normal] * 4 + [update] + [normal] * 4
    _write_trace(tmp_path, steps)
    with open(tmp_path / "x_memory_copy_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Direction", "Start_Timestamp", "End_Timestamp"])
        w.writerow(["MEMORY_COPY", "MEMORY_COPY_HOST_TO_DEVICE", 1000, 2000])          # set-up, before the analysed steps
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), str(tmp_path), "8"],
                         capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    assert "kernels per step: [4]" in lines[0], lines[0]
    assert any(l.startswith("table: mean over the 7 step(s) with the most common sequence (4 kernels)") for l in lines), out
    table = [l.split() for l in lines if l.strip() and l.split()[0].isdigit() and "nsx::" in l]
    by_name = {row[-1]: row for row in table}
    # steps are delimited by the END of the table optimizer: a_kernel starts 8810 us after it (10 ms period - 1200 + 10)
    a = by_name["nsx::a_kernel"]
    assert abs(float(a[2]) - 100.0) < 0.5                                   # duration, not polluted by occ_update's 500
    b = by_name["nsx::b_kernel"]
    assert abs(float(b[2]) - 50.0) < 0.5 and abs(float(b[3]) - 10.0) < 0.5   # 10 us of idle between a's end and b's start
    side = by_name["nsx::side_kernel"]
    assert "*" in side                                                      # runs while b_kernel does
    assert any("nsx::occ_update" in l and l.split()[0] == "0.1" for l in lines), out   # per-name table: 1 of 8 steps


def test_pmc_to_json_doubles_fetch_on_gfx950_and_converts_units(tmp_path):
    """tools/pmc_to_json.py: FETCH_SIZE / WRITE_SIZE rows (KB) -> bytes per launch and kernel, FETCH x 2 (gfx950 counts
    64-byte requests in 32-byte units, MI355X_MICROARCH.md)."""
    src = os.path.join(ROOT, "tools", "pmc_to_json.py")
    for sub, counter, value in (("fetch", "FETCH_SIZE", 1000.0), ("write", "WRITE_SIZE", 500.0)):
        d = tmp_path / sub
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id"])
            for i in range(2):
                w.writerow(["void nsx::adam_hash_factored_kernel<32, false>(float*)", counter, value, i])
    out_json = tmp_path / "out.json"
    res = subprocess.run([sys.executable, src, str(tmp_path / "fetch"), str(tmp_path / "write"), str(out_json), "synthetic"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    d = json.load(open(out_json))
    # 1000 KB fetched x 1024 x 2 + 500 KB written x 1024, per launch (the mean over the two dispatches)
    assert d["per_launch_hbm_bytes"] == {"nsx_adam_hash_factored": 1000.0 * 1024 * 2 + 500.0 * 1024}
    assert d["kernels_fetch_kb"]["nsx::adam_hash_factored_kernel"] == {"mean_kb": 1000.0, "dispatches": 2}
