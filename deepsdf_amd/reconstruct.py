"""Latent-only reconstruction (BASELINE config 4): optimise per-shape codes against a FROZEN decoder.

The fork deleted upstream's ``reconstruct.py``; only its helper vestiges remain
(deep_sdf/data.py:66-71,113-139, deep_sdf/workspace.py:137-149).  The semantics here are therefore DEFINED from the
pieces of the training step (oracle.latent_step, golden g7): eval-mode decoder, loss = mean|clamp(f(z,x)) - clamp(t)|
+ l2reg * mean(z^2), torch-Adam on z.  PARITY UNPINNED by reference files; pinned against the reference Decoder +
torch.optim.Adam in tests/golden/g7_latent_only.npz.

MI355X shape: many shapes are reconstructed AT ONCE -- the codes form a small latent table [B, L], every step is
one fused forward + dX-chain (no weight gradients: DsdfLossCfg.frozen_decoder) over B x S points, followed by a
fused Adam on the table.
"""
import torch

from . import _lib
from .engine import Engine, _ptr, _stream


def reconstruct(engine: Engine, samples_xyz, samples_sdf, *, num_iterations=800, clamp_dist=0.1, lr=5e-3,
                l2reg=1e-4, init_std=0.01, lr_drop_every=None, generator=None, z0=None, callback=None, graph=None, resample=None):
    """samples_xyz [B, S, G], samples_sdf [B, S] (device tensors; the SAME S points are used every iteration unless
    `callback(it)` returns new (xyz, sdf)).  Returns (codes [B, L], loss): `loss` is the engine's DEVICE scalar tensor (shape
    [1], the last iteration's sum over the B per-shape mean losses) -- reading it with float() synchronises, so it is left to
    the caller.

    resample = (DeviceSampleCache, scene_ids [B][, generator]): every iteration draws a FRESH balanced subsample of S points per shape from the
    cache (what upstream's reconstruct.py did) on the device; the draw key follows the device-side iteration counter
    (DeviceSampleCache.sample_sequence), so it works inside the captured graph; samples_xyz / samples_sdf then only give the shape
    [B, S, G] and are overwritten.

    graph=True (opt-in; DSDF_GRAPH=1 makes it the default for calls without a `callback`): the iteration -- [draw], hoist, forward +
    dX chain, the per-shape latent gradient, Adam -- is captured ONCE into a HIP graph and replayed, so 800 iterations are 800 graph
    launches instead of 800 x 6 kernel launches from Python.  Nothing in the iteration depends on the host: the step's Adam scalars
    (bias corrections, the learning-rate drop) come from a device-resident schedule indexed by a device-side counter
    (dsdf_adam_latent_sched).  The captured kernels are the eager ones: both ways give the same bits (tested).  NOT the default
    because it is slower on this stack: measured on MI355X / ROCm 7.2 (profiles/r04_config4.log), one shape x 8000 points:
    0.474 ms per iteration eagerly -- the host's six asynchronous launches per iteration stay ahead of the GPU -- against 0.592 ms
    as a replayed graph (hipGraphLaunch runs the six nodes with more idle time between them than stream launches leave)."""
    import math
    import os
    B, S, G = samples_xyz.shape
    L = engine.spec.latent_size
    dev = engine.device
    z = (torch.randn(B, L, generator=generator) * init_std).to(dev) if z0 is None else z0.to(dev, torch.float32).clone()
    dz, m, v = torch.zeros_like(z), torch.zeros_like(z), torch.zeros_like(z)
    seg_scene = torch.arange(B, dtype=torch.int64, device=dev)
    seg_off = torch.arange(0, B * S + 1, S, dtype=torch.int64, device=dev)
    xyz = samples_xyz.reshape(B * S, G).contiguous()
    sdf = samples_sdf.reshape(B * S).contiguous()
    lr_drop_every = int(num_iterations / 2) if lr_drop_every is None else lr_drop_every
    if num_iterations <= 0:
        return z, engine.loss
    # per-iteration Adam scalars, in double like dsdf_adam_latent_only computes them on the host: {lr_t / (1 - b1^t), sqrt(1 - b2^t)}
    b1, b2, eps = 0.9, 0.999, 1e-8
    sched = torch.tensor([[(lr * (0.1 ** (it // lr_drop_every)) if lr_drop_every > 0 else lr) / (1.0 - b1 ** (it + 1)),
                           math.sqrt(1.0 - b2 ** (it + 1))] for it in range(num_iterations)], dtype=torch.float64).to(torch.float32).to(dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    l2c = 2.0 * l2reg / L if l2reg else 0.0            # d/dz of l2reg * mean(z^2), per shape

    draw = None
    if resample is not None:
        if callback is not None:
            raise ValueError("reconstruct: `resample` and `callback` are two ways of feeding an iteration; give one")
        cache, ids = resample[0], resample[1]
        draw = cache.sample_sequence(ids, S, num_iterations, counter, xyz, sdf, generator=resample[2] if len(resample) > 2 else None)

    def iteration(x, s):
        if draw is not None:
            draw()                                     # this iteration's subsample (key: draw number *counter of the sequence)
        # every shape has its OWN mean over its S points: normalise by S, then the B losses are independent
        engine.train_forward_backward(z, dz, seg_scene, seg_off, x, s, n_norm=S, clamp_dist=clamp_dist, reg_coef=0.0,
                                      code_bound=None, training=False, accumulate=False, seg_len=S, frozen_decoder=True)
        _lib.check(engine.lib.dsdf_adam_latent_sched(_ptr(z), _ptr(dz), _ptr(m), _ptr(v), z.numel(), _ptr(sched), num_iterations,
                                                     _ptr(counter), b1, b2, eps, l2c, _stream()))

    if graph is None:
        graph = callback is None and os.environ.get("DSDF_GRAPH") == "1"
    if graph and callback is not None:
        raise ValueError("reconstruct(graph=True) replays ONE captured iteration: it cannot take a per-iteration callback")
    if not graph or num_iterations < 3:
        for it in range(num_iterations):
            if callback is not None:
                nb = callback(it)
                if nb is not None:
                    xyz, sdf = nb[0].reshape(B * S, G).contiguous(), nb[1].reshape(B * S).contiguous()
            iteration(xyz, sdf)
        return z, engine.loss
    # iteration 0 eagerly on a side stream (workspace allocation, code objects), iteration 1 captured (not executed), then replayed
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        iteration(xyz, sdf)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            iteration(xyz, sdf)
        for _ in range(1, num_iterations):
            g.replay()
    torch.cuda.current_stream(dev).wait_stream(side)
    for t in (z, dz, m, v, xyz, sdf, sched, counter):      # used on `side`: keep the caching allocator from recycling them early
        t.record_stream(side)
    return z, engine.loss
