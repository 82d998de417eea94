#!/bin/bash
# usage (through gpurun): profiles/rehearse_hang.sh  -- the 2-rank continuous-mode rehearsal on one GPU with gloo
# handed DEVICE tensors (the configuration that hung), first with all updates behind a finished search
# (today's schedule), then with one update per move step beside the search in flight; a watchdog in every
# rank reports where a stuck run is stuck (host or device).  Stops at the first run that does not finish.
mkdir -p gpurun_out
port=29611
run() {  # name, extra env...
  name=$1; shift
  for i in 1 2 3 4; do
    port=$((port + 1))
    env "$@" AZG_GLOO_DEVICE_TENSORS=1 AZG_HANG_WATCHDOG=60 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port $port -m alphazero_gomoku_amd.train --rehearse-on-one-gpu --continuous \
      --games 2048 --play-batch 512 --slots 256 --updates-per-batch 8 > gpurun_out/hang_${name}_$i.json 2> gpurun_out/hang_${name}_$i.err
    rc=$?
    echo "$name run $i: rc=$rc $(head -c 120 gpurun_out/hang_${name}_$i.json)"
    if [ $rc -ne 0 ]; then grep -a "watchdog" gpurun_out/hang_${name}_$i.err | head -8; return 1; fi
  done
  return 0
}
run behind_search AZG_X=0 && run beside_search AZG_UPDATES_BESIDE_SEARCH=1
echo finished
