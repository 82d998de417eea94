#!/bin/bash
# usage (through gpurun): profiles/train_ab.sh name1 name2 ...   -- tests/time_train.py with each library variant
# abx/libazg_NAME.so ("default": the shipped library), two rounds
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    lib=""; [ "$v" = default ] || lib=$PWD/abx/libazg_$v.so
    AZG_LIBRARY=$lib ONLY_HIP=1 timeout -k 10 120 python tests/time_train.py 8 512 100 > gpurun_out/train_ab_${v}_$round.json 2> gpurun_out/train_ab_${v}_$round.err || { echo "FAILED $v"; tail -3 gpurun_out/train_ab_${v}_$round.err; exit 1; }
    python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))['hip']; print('%-10s r%s  train step %.3f ms  forward %.3f ms  %.1f TFLOP/s' % (sys.argv[2], sys.argv[3], d['train_step_ms'], d['forward_ms'], d['train_step_tflops']))" gpurun_out/train_ab_${v}_$round.json $v $round
  done
done
