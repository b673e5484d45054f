"""Data parallelism for the training step: one process per GPU, decoder replicated, scenes (latent rows and their
Adam state) sharded by owner rank, ONE sum all-reduce of the flat decoder-gradient arena per step (RCCL over xGMI
through torch.distributed's "nccl" backend; "gloo" on CPU for the tests).

Replaces the reference's single-process nn.DataParallel (train_deep_sdf.py:353: per-step parameter broadcast +
input scatter + output gather + grad reduce).  Equivalence to a single-process run with G x the scenes per batch
holds because every rank normalises by the GLOBAL point count (train_deep_sdf.py:519,527) and the reduce is a SUM;
latent rows are owned by exactly one rank, so they need no communication at all.
"""
import os

# ROCr / RCCL read their HSA_* / NCCL_* variables when the HIP runtime initialises, i.e. at the first torch.cuda call of
# the process: defaults must therefore be in the environment BEFORE anything touches the GPU -- at import time of this
# module (deepsdf_amd/__init__.py imports it first), not inside init().  This pool's host driver only supports dmabuf
# IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0, RCCL's hipIpcGetMemHandle fails with "invalid argument".
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch                          # noqa: E402  (importing torch does not initialise HIP)
import torch.distributed as dist      # noqa: E402


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def owned_scenes(num_scenes, rank, world):
    """Contiguous block partition of scene ids: rank r owns [lo, hi)."""
    base, rem = divmod(num_scenes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat fp32 arena (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_async(flat):
    """Start the in-place SUM all-reduce of a flat fp32 arena and return its Work handle (None for a single process).
    With the nccl (= RCCL) backend the collective runs on RCCL's own stream, ordered after everything already enqueued on
    the current stream; kernels enqueued on the current stream before ``work.wait()`` overlap with it, and ``wait()`` makes
    the current stream (not the host) wait.  With gloo (CPU tests, single-card rehearsal) ``wait()`` blocks the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def is_multi():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
