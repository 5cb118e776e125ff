#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python command, summary to gpurun_out/<tag>_kernel_stats.csv:  tools/exp/prof_cmd.sh <tag> <script> [args...]
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/"$@" > $O/stdout.txt 2> $O/stderr.txt )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/prof
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:25]:
    print("%-100s calls %5s avg %9.1f us %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
