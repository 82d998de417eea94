#!/bin/bash
# Same-box A/B of library variants built by alphazero_gomoku_amd/csrc/build_variant.sh (abx/libazg_NAME.so):
# usage (through gpurun): profiles/run_ab.sh TAG "bench args" name1 name2 ...   two interleaved rounds, prints launch / step ms per variant
tag="$1"; shift; bargs="$1"; shift
mkdir -p gpurun_out/ab_$tag
for round in 1 2; do
  for v in "$@"; do
    AZG_LIBRARY=$PWD/abx/libazg_$v.so timeout -k 10 150 python bench.py $bargs --no-cpu-baseline --c4-steps 0 > gpurun_out/ab_$tag/${v}_$round.json 2> gpurun_out/ab_$tag/${v}_$round.err || { echo "FAILED $v round $round"; tail -5 gpurun_out/ab_$tag/${v}_$round.err; exit 1; }
    python - "$v" "$round" gpurun_out/ab_$tag/${v}_$round.json <<'P'
import json,sys
d=json.load(open(sys.argv[3]))
t = d.get("trained_net") or {}
print("%-14s r%s launch %.3f ms  step %.3f ms  (step-launch %.3f)  frac %.4f  playouts/s %.3e games/s %.0f%s" % (sys.argv[1], sys.argv[2], d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["ms_per_step"]-d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["playouts_per_s"], d["value"],
      "   trained net: launch %.3f ms  step %.3f ms  %.1f plies/game" % (t["avg_launch_ms"], t["ms_per_step"], t["plies_per_game"]) if t else ""))
P
  done
done
