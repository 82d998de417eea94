#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics and separate counter passes of
# bench.py for one workload.  usage: profiles/profile_gpu.sh TAG [bench.py arguments ...]
# Leaves gpurun_out/TAG_{bench_under_rocprof.json,kernel_stats.csv,pmc_search_kernel.json}.
# Every step has its own time limit; the first failure ends the script.
set -e -o pipefail
tag="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out"
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
rm -rf "$out/prof_${tag}"*
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_${tag}_stats" -o run -- python3 bench.py --no-cpu-baseline --model none --c4-steps 0 "$@" \
    > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.err"
stats="$(find "$out/prof_${tag}_stats" -name '*kernel_stats.csv' | head -1)"
cp "$stats" "$out/${tag}_kernel_stats.csv"
# (one block's counters per pass, within the per-block limits; L2 hits and misses, L1 address translation,
#  the translation cache's busy share)
for pass in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
            "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
    name="$(echo "$pass" | cut -d' ' -f1)"
    timeout -s KILL 400 rocprofv3 --pmc $pass --output-format csv -d "$out/prof_${tag}_pmc_$name" -o run -- python3 bench.py --no-cpu-baseline --model none --c4-steps 0 "$@" --preroll-max 40 --steps 8 --warmup 2 \
        > "$out/${tag}_pmc_$name.json" 2> "$out/${tag}_pmc_$name.err"
done
per_search="$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline'].get('kernel_launches_per_search', 1))" "$out/${tag}_bench_under_rocprof.json")"
python3 tests/pmc_summary.py "$out/${tag}_pmc_search_kernel.json" --launches-per-search "$per_search" "$out/prof_${tag}_pmc_"* > /dev/null
head -3 "$out/${tag}_kernel_stats.csv"
cat "$out/${tag}_pmc_search_kernel.json"
