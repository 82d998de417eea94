#!/bin/bash
for pad in 0 65536 40000; do
  AZG_TRAIN_LDSPAD=$pad ONLY_HIP=1 timeout -k 10 120 python tests/time_train.py 8 512 100 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)['hip']; print('pad $pad  train step %.3f ms  forward %.3f ms  %.1f TFLOP/s' % (d['train_step_ms'], d['forward_ms'], d['train_step_tflops']))"
done
