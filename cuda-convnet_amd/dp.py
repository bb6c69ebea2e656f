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
    """Column indices of rank `rank`: global minibatch i covers cases [i*mb*world, (i+1)*mb*world) and every rank takes a
    contiguous slice of it.  Full global minibatches split evenly; a ragged last one (the reference's tail minibatch,
    Kernel/src/data/data.cu:67-104, e.g. 10 000 cases = 78 x 128 + 16) is dealt out as evenly as it goes -- the first
    `tail % world` ranks take one case more -- so no case is dropped.  Gradients of that minibatch must then be scaled by
    its true global size, not by local cases x world: pass global_cases(...) to libkernel.startBatch."""
    import numpy as n
    gmb = minibatch * world
    cols = []
    for s in range(0, cases, gmb):
        t = min(cases, s + gmb) - s
        if t < world:
            # fewer cases than ranks: some rank would sit out the minibatch and its all-reduce would never complete.  The
            # only case left out anywhere (at most world - 1 cases of a batch); said aloud, once per call, by rank 0.
            if rank == 0:
                print("dp.shard_columns: the last %d case(s) of this %d-case batch are fewer than the %d ranks and are skipped" % (t, cases, world))
            break
        per, extra = divmod(t, world)
        lo = s + rank * per + min(rank, extra)
        cols.append(n.arange(lo, lo + per + (1 if rank < extra else 0)))
    return n.concatenate(cols) if cols else n.zeros(0, dtype=n.int64)


def global_cases(cases, minibatch, world):
    """Cases of a `cases`-case batch that shard_columns hands out over all ranks (everything, unless the tail is shorter than
    the world size): the third argument of libkernel.startBatch, from which it takes each minibatch's true global size."""
    t = cases % (minibatch * world)
    return cases - t if 0 < t < world else cases


def sum_costs(result, world):
    """finishBatch's (costs {name: [values]}, cases) of THIS rank's shard -> the same pair for the whole batch, on every rank:
    cost values and case counts are summed over the ranks through the CPU control group (float64, one small all-reduce per
    batch -- the reference has one device and simply reads the sums, Python/convnet.py:120-134).  What rank 0 prints, what a
    test pass reports and what conditional_save compares are then about the global batch, not about 1 / world of it."""
    if world <= 1:
        return result
    import torch
    costs, cases = result
    names = sorted(costs)                      # every rank holds the same model: same names, same value counts
    flat = [float(cases)] + [float(v) for name in names for v in costs[name]]
    t = torch.tensor(flat, dtype=torch.float64)
    init_control_group().all_reduce(t)         # sum
    flat = t.tolist()
    out, at = {}, 1
    for name in names:
        k = len(costs[name])
        out[name] = flat[at:at + k]
        at += k
    return dict((name, out[name]) for name in costs), int(round(flat[0]))
