#!/bin/bash
# The N > 1 code paths of bench.py on a box with ONE GPU: two ranks share GPU 0, the process group
# is gloo on the CPU.  (1) no data-path collective: tiles, slabs, per-rank bookkeeping of the bench;
# (2) the RCCL exchange: works only if RCCL accepts two ranks on one device.
#   gpurun --timeout 600 -- 'bash profiles/tools/two_ranks_one_gpu.sh r02x'
set -u
O=gpurun_out/${1:-two_ranks}; mkdir -p $O
export VKR_BENCH_DEVICE=0 VKR_BENCH_BACKEND=gloo
RUN="timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 40 --warmup 5"
$RUN --exchange none --no-secondary > $O/none.json 2> $O/none.err; echo "exchange none rc $?"
grep '^{' $O/none.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['value'], d['ms_per_step'], d['config']['parallelism'])"
$RUN --no-secondary > $O/rgba32f.json 2> $O/rgba32f.err; echo "exchange rgba32f rc $?"
grep '^{' $O/rgba32f.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['value'], d['ms_per_step'], d.get('stages'), d.get('scaling_parity'))"
tail -3 $O/rgba32f.err
