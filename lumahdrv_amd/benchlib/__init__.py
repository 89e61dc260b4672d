"""What bench.py is made of (bench.py itself is the command line and the one JSON line):

    plan.py     the workload as a PURE FUNCTION of the arguments: which configuration, how many frames every rank keeps
                resident, what it asks of the chunk pool; `--plan-only`.  No GPU, no process group, no free-memory query.
    timing.py   the W warm-up / exactly-K-steps timed region (barrier + synchronize, MAX over ranks).
    resident.py the resident stream of a plan in HBM: batches placed in pool chunks where the pool has them, plain
                allocations for the rest -- never fewer frames than the plan says.
    legs.py     one workload's legs (encode, decode, round trip, packed decode, float inputs) and its roofline blocks.
    stream.py   BASELINE configs[4]: ONE stream block-sharded over the ranks (torch.distributed or the C ABI's many-GPU layer).

Nothing here imports `oracle/`: the CPU baseline leg lives in bench.py.
"""
