#!/bin/bash
# tools/bench/torchrun_n1.sh -- the SCALE curve's N = 1 point is `python -m torch.distributed.run ... bench.py --gpus 1`; the BENCH line
# is `python bench.py`.  Both forms, alternating, three times each, on one box, printing the argument-determined keys of `config`
# next to the rates (-> profiles/r06_torchrun_n1.txt): the keys must be EQUAL, the rates within the run-to-run spread.  The last two
# lines: the launcher form with LUMAHIP_BENCH_FORCE_DIST=1 (process group on RCCL with one rank: init, table broadcast, barrier,
# the all_reduce of ones behind `rccl_ranks_seen`) and `--stream-frames 2000 --plan-only --gpus 8` (no GPU work).
cd "$GRAFT_REPO_ROOT"
F="--gpus 1 --no-cpu-baseline --no-other-workloads --no-facade-hostfed --no-placement-off"
SHOW='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); c=d["config"]
print("%-9s value %.0f  ms_per_step %.4f  frac %.4f | resident_frames %d stream_frames %d per_rank %s frames_per_step %d %dx%d scaling %s degraded %s rccl_ranks_seen %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], c["resident_frames"], c["stream_frames"], c["resident_frames_per_rank"], c["frames_per_step"], c["width"], c["height"], c["scaling"], d["config_degraded"], d.get("rccl_ranks_seen")))'
for i in 1 2 3; do
  python bench.py $F 2>/dev/null | python -c "$SHOW" plain
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py $F 2>/dev/null | python -c "$SHOW" launcher
done
LUMAHIP_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py $F 2>/dev/null | python -c "$SHOW" rccl_n1
python bench.py --gpus 8 --stream-frames 2000 --plan-only | python -c 'import json,sys; d=json.load(sys.stdin); print("plan --gpus 8 --stream-frames 2000: per rank", d["config"]["resident_frames_per_rank"], "scaling", d["scaling"], "expected digest", d["expected_stream_digest"], "fits", d["fits"])'
