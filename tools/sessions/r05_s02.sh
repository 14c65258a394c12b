#!/bin/bash
# r05 session 2: Winograd stand-in timings, level-3 conv split-K sweep, OPT GEMM split-K sweep; FETCH_SIZE of one level-3 conv
R=$PWD; O=$R/gpurun_out/r05_s02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/tools/r05_probe.py > $O/probe.log 2>&1
cat $O/probe.log | tail -40
GILL_OP_REPEAT=10 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc -o x --output-format csv -- python $R/tools/one_op.py conv 8 8 8 1280 0 1280 > $O/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for fn in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = (r["Kernel_Name"][:50], r["Grid_Size"]); agg[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in agg: print(k, "FETCH_SIZE KiB per dispatch (raw, x2 for wide reads per the guide):", round(agg[k] / cnt[k]), "dispatches", cnt[k])
PY
