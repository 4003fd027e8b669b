#!/bin/bash
# Rehearsal of bench.py's multi-rank path on ONE GPU: N ranks under torch.distributed.run, all on GPU 0
# (--share-device), the accumulator exchange over gloo (RCCL refuses two ranks on one device: "Duplicate
# GPU detected").  It exercises everything of the N > 1 path except RCCL itself: rendezvous, the split of
# a frame's samples over the ranks, the asynchronous all-reduce per frame, the fence, the max over ranks.
# Usage (repository root, GPU box): tools/rehearse_multi_gpu.sh [ranks=2] [spp=64] [config=c1] [tiles=]
#   e.g. tools/rehearse_multi_gpu.sh 4 32 c4 4x2   (BASELINE.json configs[4]: hindu 4K, tile x sample-chunk shards)
set -eu
n=${1:-2}; spp=${2:-64}; config=${3:-c1}; tiles=${4:-}
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus "$n" --steps 2 --warmup 1 --spp "$spp" --config "$config" ${tiles:+--tiles "$tiles"} --backend gloo --share-device --no-cpu-baseline --no-ray-batches --no-configs
