#!/bin/bash
# On the GPU box: the many-plane table optimizer pass (level-parallel runs) under rocprofv3 -- kernel statistics and the two
# HBM counter passes (separate runs) of tools/level_parallel_bench.py at the shape of the finest levels' owner of 8 ranks
# (192 planes, 2^20 entries).  Results: gpurun_out/r05_h/.
set -u
out=gpurun_out/r05_h
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=.
RUN="python tools/level_parallel_bench.py --adam-only --world 8 --last-rank-only"
$RUN > $out/plain.json 2> $out/plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o h -- $RUN > $out/trace.json 2> $out/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o h -- $RUN > /dev/null 2> $out/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o h -- $RUN > /dev/null 2> $out/write.err
python tools/pmc_to_json.py $out/fetch $out/write $out/pmc.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $RUN" > $out/pmc_summary.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $out/sq -o h -- $RUN > /dev/null 2> $out/sq.err
python - <<'PY' > $out/sq_summary.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r05_h/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "adam" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()})
    if m.get("SQ_BUSY_CYCLES"):
        print("   mfma busy / sq busy", round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_BUSY_CYCLES"], 4),
              " valu per mfma", round(m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1), 2),
              " wait_any / wave cycles", round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3))
PY
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out \( -name "*counter_collection.csv" -o -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
cat $out/plain.json; echo; head -6 $out/kernel_stats.csv | cut -c1-200; cat $out/pmc_summary.txt | head -12; cat $out/sq_summary.txt; tail -2 $out/*.err | head -30
