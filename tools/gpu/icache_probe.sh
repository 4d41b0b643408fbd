#!/bin/bash
# instruction-cache counters of one bench configuration:  tools/gpu/icache_probe.sh <variant|base> <env-id> <precision>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/icache; mkdir -p $O
if [ "$1" != base ]; then export DART_STEPPER_LIB=$R/abtest/lib_$1.so; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $O/names.txt
B="python $R/bench.py --no-extras --env-id $2 --precision $3 --envs 65536 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/p1 -o p1 --output-format csv -- $B > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL -d $O/p2 -o p2 --output-format csv -- $B > $O/p2.log 2>&1
cd $R
python - <<PY
import csv, collections, glob
for d in ("p1", "p2"):
    for f in glob.glob("$O/%s/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print("$1 $2 f$3", k, "%.4e per launch" % (sum(v) / len(v)))
PY
cat $O/names.txt; echo; tail -3 $O/p1.log | cut -c1-300
find $O -name '*.db' -delete
