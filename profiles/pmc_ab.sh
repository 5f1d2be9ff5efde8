#!/bin/bash
# One counter pass (SQ instruction and wait counters) of the benchmark frame at 64 spp for the in-tree build and for variants.
# usage: profiles/pmc_ab.sh <outdir> <variant|main> ...
export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
  ITERATIONS=2 timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/$v -o p -- python3 profiles/stats_run.py 64 chess32 > $out/$v.log 2>&1 || { echo "pass $v failed"; exit 1; }
  python3 - $out/$v $v <<'PY'
import csv, glob, sys, collections
tot = collections.OrderedDict(); n = {}
for f in glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "trace_paths" not in row["Kernel_Name"]: continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]); n.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
paths = 1920 * 1080 * 64
print(sys.argv[2], {k: round(v / len(n[k]) / paths, 2) for k, v in tot.items()})
PY
done
