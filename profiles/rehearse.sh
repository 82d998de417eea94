#!/bin/bash
# usage (through gpurun): profiles/rehearse.sh TAG  -- the 2-rank rehearsals on one GPU (gloo, collectives staged
# through the host): sharded bench, training loop in cohort and in continuous mode; records under gpurun_out/TAG_*
tag=$1
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 3 > gpurun_out/${tag}_rehearsal_bench_2ranks.json 2> gpurun_out/${tag}_rehearsal_bench.err || { echo "bench rehearsal failed"; tail -5 gpurun_out/${tag}_rehearsal_bench.err; exit 1; }
tail -c 600 gpurun_out/${tag}_rehearsal_bench_2ranks.json; echo
for mode in cohort continuous; do
  flag=""; [ $mode = continuous ] && flag="--continuous"
  AZG_HANG_WATCHDOG=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2971$((RANDOM % 10)) \
    -m alphazero_gomoku_amd.train --rehearse-on-one-gpu $flag --games 2048 --play-batch 512 --slots 256 --updates-per-batch 8 \
    > gpurun_out/${tag}_rehearsal_train_2ranks_$mode.json 2> gpurun_out/${tag}_rehearsal_train_$mode.err || { echo "$mode rehearsal failed"; tail -5 gpurun_out/${tag}_rehearsal_train_$mode.err; exit 1; }
  echo "$mode: $(grep -a '^{' gpurun_out/${tag}_rehearsal_train_2ranks_$mode.json | head -c 400)"
done
