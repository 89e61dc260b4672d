#!/bin/bash
# tools/bench/torchrun_n1.sh -- the SCALE curve's N = 1 point is `python -m torch.distributed.run ... bench.py --gpus 1`; the BENCH line
# is `python bench.py`.  Both forms, alternating, three times each, on one box -> profiles/r05_torchrun_n1.txt
cd "$GRAFT_REPO_ROOT"
F="--gpus 1 --no-cpu-baseline --no-other-workloads --no-facade-hostfed --no-placement-off"
for i in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain    value %.0f  ms_per_step %.4f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('launcher value %.0f  ms_per_step %.4f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
