#!/bin/bash
# usage (through gpurun): profiles/stamps_run.sh TAG  -- tests/stamps.py with the -DAZG_STAMPS build abx/libazg_stamps.so:
# random-init net and the trained 8x8 net of the fixture; leaves gpurun_out/TAG_stamps_{random,trained}.txt
tag=$1
mkdir -p gpurun_out
AZG_LIBRARY=$PWD/abx/libazg_stamps.so timeout -k 10 200 python tests/stamps.py 1024 8 > gpurun_out/${tag}_stamps_random.txt 2> gpurun_out/${tag}_stamps_random.err || { echo FAILED; tail -5 gpurun_out/${tag}_stamps_random.err; exit 1; }
AZG_LIBRARY=$PWD/abx/libazg_stamps.so MODEL=tests/golden/g10_trained.npz WARM=100 timeout -k 10 200 python tests/stamps.py 1024 8 > gpurun_out/${tag}_stamps_trained.txt 2> gpurun_out/${tag}_stamps_trained.err || { echo FAILED; tail -5 gpurun_out/${tag}_stamps_trained.err; exit 1; }
head -32 gpurun_out/${tag}_stamps_random.txt; echo; head -32 gpurun_out/${tag}_stamps_trained.txt
