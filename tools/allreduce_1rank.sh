#!/bin/bash
# what the CAPTURED gradient all-reduce costs per step with ONE rank (no 8-GPU node is available to the
# builder: the collective then moves no data, what remains is its launch / graph-node cost) - bench.py as a plain process
# (no exchange), under torch.distributed.run with one all-reduce, with the two-bucket exchange, and with --ar-buckets auto.
#   tools/allreduce_1rank.sh > profiles/r04_allreduce_1rank.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
pick='import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], d["ms_per_step"], "ms/step", d["value"], "samples/s", json.dumps(d["config"].get("allreduce")))'
for rep in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-parity 2>/dev/null | python -c "$pick" "single-process(no-exchange)"
  for b in 1 2 auto; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + rep * 10 + ${#b})) \
      bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-parity --ar-buckets $b 2>/dev/null | python -c "$pick" "torchrun-1-rank-buckets-$b"
  done
done
