"""Data-parallel set-up: one process per GPU (launched by torch.distributed.run), gradients
all-reduced inside libkernel over RCCL.  This module only does the rendezvous: rank 0 creates the
RCCL unique id, every rank receives it through torch.distributed's CPU (gloo) group, then each
rank's libkernel joins the communicator.  There is no counterpart in the reference (single GPU)."""
import os


def env():
    """(rank, world_size, local_rank) from the launcher's environment; (0, 1, 0) when not launched."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))


def init_control_group():
    """CPU-side process group used for the id broadcast, barriers and timing reductions."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo")
    return dist


def connect(libmodel, rank, world, force_single=False):
    """Collective: every rank calls this after initModel."""
    if world == 1 and not force_single:
        return
    if world > 1:
        dist = init_control_group()
        box = [libmodel.commUniqueId() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    else:
        uid = libmodel.commUniqueId()
    libmodel.initComm(rank, world, uid, 1 if force_single else 0)


def shard_columns(cases, minibatch, rank, world):
    """Column indices of rank `rank`: global minibatch i covers cases [i*mb*world, (i+1)*mb*world);
    each rank takes an equal contiguous slice of it (a remainder that does not divide is dropped)."""
    import numpy as n
    gmb = minibatch * world
    cols = []
    for s in range(0, cases, gmb):
        per = (min(cases, s + gmb) - s) // world
        cols.append(n.arange(s + rank * per, s + (rank + 1) * per))
    return n.concatenate(cols) if cols else n.zeros(0, dtype=n.int64)
