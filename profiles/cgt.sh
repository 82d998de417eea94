#!/bin/bash
# usage (through gpurun): profiles/cgt.sh  -- every tests/micro/cgt_* timing variant, one after the other
for f in $(ls tests/micro/cgt_* | sort -t_ -k2 -n); do timeout -k 10 30 $f || exit 1; done
