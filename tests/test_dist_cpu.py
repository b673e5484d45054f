"""CPU, world_size 2, gloo: the data-parallel path (deepsdf_amd/dist.py).  Each rank owns half of the scenes, computes
its local gradients with the GLOBAL normaliser, the product all-reduce sums the decoder arena; the result must equal
the single-process gradients of the concatenated batch.  The oracle stands in for the HIP step here
(no GPU in this tier); the GPU box runs the same reduction over RCCL in bench.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

KW = dict(dims=[32] * 3, dropout=[0, 1, 2], dropout_prob=0.0, norm_layers=[0, 1, 2], latent_in=[2], xyz_in_all=False,
          use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(scenes, S, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.tensor(scenes).repeat_interleave(S)
    xyz = torch.rand(idx.numel(), 3, generator=g) * 2 - 1
    gt = xyz.norm(dim=1, keepdim=True) - 0.5
    return idx, xyz, gt


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepsdf_amd import dist
    from deepsdf_amd.net import NetSpec
    from oracle import deepsdf_oracle as orc
    torch.set_num_threads(1)
    r, _, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    L, S, n_scenes = 6, 16, 8
    net = orc.make_net(L, **KW)
    spec = NetSpec(L, **KW)
    params = orc.init_params(net, 3)
    lat = torch.randn(n_scenes, L, generator=torch.Generator().manual_seed(4)) / 2
    lo, hi = dist.owned_scenes(n_scenes, rank, world)
    idx, xyz, gt = _batch(list(range(lo, hi)), S, 10 + rank)
    n_global = S * n_scenes
    res = orc.step_gradients(net, params, lat.clone(), idx, xyz, gt, delta=0.1, code_bound=1.0, code_reg=True,
                             code_reg_lambda=1e-4, epoch=50, n_norm=n_global, training=False)
    flat = torch.zeros(spec.n_params)
    for p in spec.params:                                   # arena in named_parameters order, as the engine holds it
        flat[p.offset:p.offset + p.numel] = res["grads"][p.name].reshape(-1)
    dist.allreduce_sum_(flat)
    loss = torch.tensor([float(res["loss"])], dtype=torch.float64)
    torch.distributed.all_reduce(loss)
    mx = dist.max_over_ranks(float(rank), "cpu")
    dist.barrier()
    torch.save(dict(flat=flat, dlat=res["dlat"], loss=loss, lo=lo, hi=hi, mx=mx), os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path):
    from deepsdf_amd import dist
    from deepsdf_amd.net import NetSpec
    from oracle import deepsdf_oracle as orc
    assert dist.owned_scenes(10, 0, 3) == (0, 4) and dist.owned_scenes(10, 2, 3) == (7, 10)
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=True) for r in range(world)]
    assert torch.equal(outs[0]["flat"], outs[1]["flat"])          # replicas see the identical reduced gradient
    assert outs[0]["mx"] == 1.0
    L, S, n_scenes = 6, 16, 8
    net = orc.make_net(L, **KW)
    spec = NetSpec(L, **KW)
    params = orc.init_params(net, 3)
    lat = torch.randn(n_scenes, L, generator=torch.Generator().manual_seed(4)) / 2
    parts = [_batch(list(range(o["lo"], o["hi"])), S, 10 + r) for r, o in enumerate(outs)]
    idx, xyz, gt = (torch.cat([p[i] for p in parts]) for i in range(3))
    ref = orc.step_gradients(net, params, lat.clone(), idx, xyz, gt, delta=0.1, code_bound=1.0, code_reg=True,
                             code_reg_lambda=1e-4, epoch=50, training=False)
    for p in spec.params:
        got = outs[0]["flat"][p.offset:p.offset + p.numel].reshape(p.shape)
        assert torch.allclose(got, ref["grads"][p.name].reshape(p.shape), rtol=1e-4, atol=1e-7), p.name
    # latent rows are owned by exactly one rank: no communication, rows outside the shard stay zero
    dl = sum(o["dlat"] for o in outs)
    assert torch.allclose(dl, ref["dlat"], rtol=1e-4, atol=1e-8)
    for o in outs:
        mask = torch.ones(n_scenes, dtype=torch.bool); mask[o["lo"]:o["hi"]] = False
        assert float(o["dlat"][mask].abs().max()) == 0.0
    assert abs(float(outs[0]["loss"]) - float(ref["loss"])) < 1e-6
