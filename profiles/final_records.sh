#!/bin/bash
# usage (through gpurun): profiles/final_records.sh TAG  -- the records of a build: bench with the default and the driver's
# flags, configuration C4 (15x15 / 800 / 512), rocprofv3 kernel statistics and counter passes of both; under gpurun_out/TAG_*
tag=$1
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench_default_flags.json 2> gpurun_out/${tag}_bench_default_flags.err || { echo "bench failed"; exit 1; }
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench_driver_flags.err || { echo "bench failed"; exit 1; }
timeout -k 10 600 python bench.py --board 15 --playouts 800 --games 512 --steps 24 --warmup 8 > gpurun_out/${tag}_bench_c4_15x15.json 2> gpurun_out/${tag}_bench_c4.err || { echo "c4 bench failed"; exit 1; }
profiles/profile_gpu.sh ${tag} > gpurun_out/${tag}_profile.log 2>&1 || { echo "profile failed"; tail -5 gpurun_out/${tag}_profile.log; exit 1; }
profiles/profile_gpu.sh ${tag}_c4 --board 15 --playouts 800 --games 512 --steps 16 --warmup 4 > gpurun_out/${tag}_c4_profile.log 2>&1 || { echo "c4 profile failed"; tail -5 gpurun_out/${tag}_c4_profile.log; exit 1; }
python3 - $tag <<'P'
import json, sys
t = sys.argv[1]
for f in ("bench_default_flags", "bench_driver_flags", "bench_c4_15x15", "bench_under_rocprof", "c4_bench_under_rocprof"):
    d = json.load(open("gpurun_out/%s_%s.json" % (t, f)))
    tn = d.get("trained_net") or {}
    print("%-24s value %.0f  step %.3f ms  search %.3f ms  frac %.4f  cpu %s  trained: %s" % (f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"],
          (d.get("cpu_baseline") or {}).get("value"), (tn.get("avg_launch_ms"), tn.get("value"))))
P
