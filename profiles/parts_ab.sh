#!/bin/bash
# usage (through gpurun): profiles/parts_ab.sh "bench args" p1 p2 ...  -- bench.py with a move's playouts spread over
# p launches (AZG_SEARCH_PARTS), two interleaved rounds
bargs="$1"; shift
mkdir -p gpurun_out/ab_parts
for round in 1 2; do
  for v in "$@"; do
    AZG_SEARCH_PARTS=$v timeout -k 10 200 python bench.py $bargs --no-cpu-baseline --c4-steps 0 > gpurun_out/ab_parts/${v}_$round.json 2> gpurun_out/ab_parts/${v}_$round.err || { echo "FAILED $v round $round"; tail -5 gpurun_out/ab_parts/${v}_$round.err; exit 1; }
    python - "$v" "$round" gpurun_out/ab_parts/${v}_$round.json <<'P'
import json,sys
d=json.load(open(sys.argv[3]))
t = d.get("trained_net") or {}
print("parts %-3s r%s search %.3f ms per move  step %.3f ms  frac %.4f  games/s %.0f%s" % (sys.argv[1], sys.argv[2], d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["roofline"]["frac"], d["value"],
      "   trained net: search %.3f ms  step %.3f ms" % (t["avg_launch_ms"], t["ms_per_step"]) if t else ""))
P
  done
done
