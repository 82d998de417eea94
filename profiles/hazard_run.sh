#!/bin/bash
# usage (through gpurun): [BOARD=8] profiles/hazard_run.sh name1 name2 ...  -- tests/hazard_check.py once per library variant
# abx/libazg_NAME.so; a variant that fails or times out ends the script (nothing else touches the GPU after that).
mkdir -p gpurun_out
for v in "$@"; do
  AZG_LIBRARY=$PWD/abx/libazg_$v.so timeout -k 10 120 python tests/hazard_check.py ${SLOTS:-256} 8 > gpurun_out/hazard_$v.json 2> gpurun_out/hazard_$v.err || { echo "FAILED $v"; tail -5 gpurun_out/hazard_$v.err; exit 1; }
  cat gpurun_out/hazard_$v.json
done
