#!/bin/bash
# Run on the GPU box:  bash tools/profile_rowfetch.sh <name>      -> profiles/<name>_row_fetch.txt
# tools/micro/row_fetch (times) and the same binary under rocprofv3 --pmc FETCH_SIZE (what the counter tallies for 64-byte row reads against streaming reads).
set -o pipefail
NAME=${1:-r05_x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${NAME}_rowfetch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/micro/row_fetch > $OUT/times.txt || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- $R/tools/micro/row_fetch > $OUT/pmc_run.txt 2> $OUT/pmc_err.txt || { tail -5 $OUT/pmc_err.txt; exit 1; }
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/pmc/*/*_counter_collection.csv")[0]
per = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "read_kernel" in r["Kernel_Name"]:
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
B = 4 * 2 ** 30
want = {"0": ("full", B), "1": ("half", B // 2), "2": ("halfbuf", B // 2), "3": ("rows1600", B // 25)}
lines = open("$OUT/times.txt").read().rstrip().split("\n")
lines.append("")
lines.append("rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/row_fetch: FETCH_SIZE (KB, as reported) per launch, against the bytes the kernel requests")
for k, v in sorted(per.items()):
    mode = k.split("<")[1].split(">")[0]
    nm, req = want[mode]
    kb = sum(v) / len(v)
    lines.append("  %-9s FETCH_SIZE %12.0f KB = %6.3f x the %9.0f KB requested" % (nm, kb, kb * 1024 / req, req / 1024))
open("profiles/${NAME}_row_fetch.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_row_fetch.txt gpurun_out/profiles_$NAME/
