#!/bin/bash
# usage (through gpurun): profiles/train_prof.sh TAG [board batch]  -- gradient parity tests, then tests/time_train.py plain and
# under rocprofv3 --kernel-trace --stats; leaves gpurun_out/TAG_time_train.json and gpurun_out/TAG_train_kernel_stats.csv
tag="$1"; board="${2:-8}"; batch="${3:-512}"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py -x -q -k "native_gradients or train_step_matches" --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { echo "TESTS FAILED"; tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
timeout -k 10 200 python tests/time_train.py $board $batch 50 > gpurun_out/${tag}_time_train.json 2> gpurun_out/${tag}_time_train.err || { echo "TIMING FAILED"; tail -5 gpurun_out/${tag}_time_train.err; exit 1; }
cat gpurun_out/${tag}_time_train.json
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${tag}
ONLY_HIP=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -o run -- python3 tests/time_train.py $board $batch 50 > /dev/null 2> gpurun_out/${tag}_prof.err || { echo "PROFILE FAILED"; exit 1; }
f=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_train_kernel_stats.csv
python3 - gpurun_out/${tag}_train_kernel_stats.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    if r["Name"].startswith(("azg::", "void azg::")):
        per_step = float(r["TotalDurationNs"]) / 53.0 / 1e3
        tot += per_step
        print("%8.1f us/step  %4d calls  avg %7.1f us  %s" % (per_step, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
print("library kernels per train step + forward: %.1f us" % tot)
P
