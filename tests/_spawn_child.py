"""rank program for tests/test_bench_spawn_cpu.py: what bench.py's ranks do around the measurement, without a GPU —
rendezvous from the torchrun environment (gloo), one collective, rank 0 prints the JSON line, optional failure of one rank"""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print("chatter from rank %d" % rank)
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(3)
if rank == 0:
    print(json.dumps({"metric": "spawn self-test", "value": float(t.item()), "n_gpus": world, "argv": sys.argv[1:]}))
dist.barrier()
dist.destroy_process_group()
