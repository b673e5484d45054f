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
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world).

    Rehearsal knobs for a ONE-GPU box (tests/test_gpu_dist.py; the driver's multi-GPU runs use neither):
    DSDF_DIST_BACKEND=gloo selects the transport when the caller names none, DSDF_SINGLE_DEVICE=1 puts every rank on
    device 0 (returned as local_rank), so that bench.py, the trainer and the tests' worker share one definition.
    DSDF_DIST_FORCE_GROUP=1 creates the process group even for ONE rank and makes is_multi() true for it: the whole
    data-parallel call sequence (asynchronous all-reduce on RCCL's stream, stream-ordered wait, object collectives) then
    executes through real RCCL on a one-GPU box -- what gloo cannot show, because its wait() blocks the host."""
    rank, local, world = env_world()
    if os.environ.get("DSDF_SINGLE_DEVICE") == "1":
        local = 0
    if (world > 1 or force_group()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("DSDF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def force_group():
    return os.environ.get("DSDF_DIST_FORCE_GROUP") == "1"


def _active():
    """A process group exists and the data-parallel call sequence applies to it (more than one rank, or the one-rank
    rehearsal group of DSDF_DIST_FORCE_GROUP=1)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_group())


def replicas_identical(flat):
    """True iff every rank holds bit-identical contents of `flat` (MAX and MIN over ranks of the raw bit patterns agree).
    The data-parallel step keeps decoder replicas identical by construction (same reduced gradient, same deterministic
    Adam); the trainer checks it before every checkpoint, so a diverged replica is an error, not a silently saved rank 0."""
    if not is_multi():
        return True
    bits = flat.detach().contiguous().view(torch.int32)
    hi, lo = bits.clone(), bits.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))


def owned_scenes(num_scenes, rank, world):
    """Contiguous block partition of scene ids: rank r owns [lo, hi)."""
    base, rem = divmod(num_scenes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat fp32 arena (no-op for a single process)."""
    if _active():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_sum_async(flat):
    """Start the in-place SUM all-reduce of a flat fp32 arena and return its Work handle (None for a single process).
    With the nccl (= RCCL) backend the collective runs on RCCL's own stream, ordered after everything already enqueued on
    the current stream; kernels enqueued on the current stream before ``work.wait()`` overlap with it, and ``wait()`` makes
    the current stream (not the host) wait.  With gloo (CPU tests, single-card rehearsal) ``wait()`` blocks the host."""
    if _active():
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def is_multi():
    return _active()


def barrier():
    if _active():
        dist.barrier()


def shutdown():
    """Leave the process group in step (barrier) and tear it down; a no-op for a single process."""
    if dist.is_available() and dist.is_initialized():
        if _active():
            dist.barrier()
        dist.destroy_process_group()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
