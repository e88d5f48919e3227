"""Process-group glue of the training CLIs: `python -m torch.distributed.run --nproc-per-node N -m nasrec_amd.main_train ...` runs one
process per GPU (backend "nccl" == RCCL over xGMI); without torchrun's environment the CLIs run single-process as before.

The reference has no distributed mode.  Under data parallelism every rank reads its own shards, the global batch
is `train_batch_size x world`, the path of a supernet step is drawn from an identically seeded np.random on every rank
(supernet.py:525-529 draws one path per step), and the step equals a single process at the global batch (nasrec_amd/parallel.py)."""
import os

import torch
import torch.distributed as dist


def init_from_env(args=None):
    """-> (rank, world).  Must run before anything touches the GPU (a process that has initialised the GPU is never re-executed)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if args is not None:
        args.gpu = local
    import numpy as np
    np.random.seed(int(os.environ.get("NASREC_PATH_SEED", "0")))  # one path per step, the same on every rank
    return rank, world


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1
