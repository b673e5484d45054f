"""-m gpu: building-block kernels through the C ABI against plain torch fp32 / the oracle's integer hash."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.golden_io import rel_err

pytestmark = pytest.mark.gpu


def _lib():
    from deepsdf_amd import _lib as L
    return L, L.lib()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 512, 512), (300, 253, 259), (64, 1, 40), (1000, 260, 253),
                                   (16384, 512, 512)])
def test_gemm_nt(M, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    lda = ldb = (K + 3) // 4 * 4
    A = torch.zeros(M, lda); A[:, :K] = torch.randn(M, K, generator=g)
    B = torch.zeros(N, ldb); B[:, :K] = torch.randn(N, K, generator=g)
    if lda > K:  # poison the padding: the kernel must zero-fill beyond K itself
        A[:, K:] = float("nan"); B[:, K:] = float("nan")
    bias = torch.randn(N, generator=g)
    Ad, Bd, bd = A.cuda(), B.cuda(), bias.cuda()
    Cd = torch.full((M, N + 3), 7.0, device="cuda")
    L.check(lib.dsdf_gemm_nt(Ad.data_ptr(), lda, Bd.data_ptr(), ldb, Cd.data_ptr(), N + 3, M, N, K, bd.data_ptr(), _s()))
    ref = (A[:, :K].double() @ B[:, :K].double().t() + bias.double())
    assert rel_err(Cd[:, :N].cpu(), ref) < 2e-6
    assert bool((Cd[:, N:] == 7.0).all())  # nothing written outside [M, N]


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (512, 512, 16384), (253, 259, 1000), (1, 48, 96), (512, 259, 300)])
def test_gemm_tn(M, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    lda, ldb = (M + 3) // 4 * 4 + 4, (N + 3) // 4 * 4
    A = torch.full((K, lda), float("nan")); A[:, :M] = torch.randn(K, M, generator=g)
    B = torch.full((K, ldb), float("nan")); B[:, :N] = torch.randn(K, N, generator=g)
    Ad, Bd = A.cuda(), B.cuda()
    Cd = torch.zeros(M, N, device="cuda")
    ws = torch.empty(64 * ((M * N + 63) // 64 * 64) * 4 + 4 * M + 1024, dtype=torch.uint8, device="cuda")
    L.check(lib.dsdf_gemm_tn(Ad.data_ptr(), lda, Bd.data_ptr(), ldb, Cd.data_ptr(), N, M, N, K, ws.data_ptr(), ws.numel(), _s()))
    ref = A[:, :M].double().t() @ B[:, :N].double()
    assert rel_err(Cd.cpu(), ref) < 2e-6


def test_gemm_nt_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    L, lib = _lib()
    n = 128
    A = torch.eye(n).cuda()
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97).cuda()
    Cd = torch.zeros(n, n, device="cuda")
    L.check(lib.dsdf_gemm_nt(A.data_ptr(), n, B.data_ptr(), n, Cd.data_ptr(), n, n, n, n, None, _s()))
    assert torch.equal(Cd, B.t())


@pytest.mark.parametrize("row_offset", [0, 37, 4096])
def test_dropout_hash_bit_exact(row_offset):
    from oracle import deepsdf_oracle as orc
    L, lib = _lib()
    key = orc.dropout_layer_key(99, 3, 5)
    rows, cols = 301, 512
    out = torch.zeros(rows, cols, dtype=torch.uint8, device="cuda")
    L.check(lib.dsdf_dropout_mask(key, 0.2, rows, cols, row_offset, out.data_ptr(), _s()))
    ref = orc.dropout_keep(key, rows, cols, 0.2, row_offset)
    assert np.array_equal(out.cpu().numpy().astype(bool), ref)


def test_errors_are_reported():
    L, lib = _lib()
    from deepsdf_amd.net import NetSpec
    spec = NetSpec(4, [32, 32], 3)
    net = spec.c_struct()
    net.out_dim[spec.n_layers - 1] = 2
    b = C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(net), 10, 1, C.byref(b)) == -1
    assert b"out_dim 1" in lib.dsdf_last_error()


def test_sample_batch_matches_the_oracle_permutation():
    """dsdf_sample_batch (deepsdf_amd/csrc/sample.hpp) against oracle.sample_rows: the gathered rows are EXACTLY the rows
    the specification names (integer work: bit-exact), in the reference's order (positives then negatives per scene)."""
    from deepsdf_amd.data import DeviceSampleCache
    from oracle import deepsdf_oracle as orc
    sizes = [(5000, 4000), (37, 9000), (8000, 20), (129, 131), (1 << 14, 1 << 14)]
    items, base = [], 0
    for npos, nneg in sizes:          # a row's content encodes (scene, sign, index): xyz = (scene, index, sign), sdf = +/-(index + 1)
        k = len(items)
        pos = torch.stack([torch.full((npos,), float(k)), torch.arange(npos).float(), torch.zeros(npos), torch.arange(npos).float() + 1], 1)
        neg = torch.stack([torch.full((nneg,), float(k)), torch.arange(nneg).float(), torch.ones(nneg), -(torch.arange(nneg).float() + 1)], 1)
        items.append((pos, neg))
    cache = DeviceSampleCache(items, 3, "cuda")
    ids = torch.tensor([3, 0, 1, 4, 2, 0])
    for subsample, key in ((256, 0x0123456789ABCDEF), (101, 77)):
        xyz, sdf = cache.sample(ids, subsample, key=key)
        S = 2 * (subsample // 2)
        assert xyz.shape == (len(ids) * S, 3) and sdf.shape == (len(ids) * S,)
        xyz, sdf = xyz.cpu().numpy(), sdf.cpu().numpy()
        for b, k in enumerate(ids.tolist()):
            p, q = orc.sample_rows(sizes[k][0], sizes[k][1], subsample, key, k)
            blk, lab = xyz[b * S:(b + 1) * S], sdf[b * S:(b + 1) * S]
            assert (blk[:, 0] == k).all()
            assert np.array_equal(blk[:len(p), 1].astype(np.int64), p.astype(np.int64)) and (blk[:len(p), 2] == 0).all()
            assert np.array_equal(blk[len(p):, 1].astype(np.int64), q.astype(np.int64)) and (blk[len(p):, 2] == 1).all()
            assert np.array_equal(lab[:len(p)], p.astype(np.float32) + 1) and np.array_equal(lab[len(p):], -(q.astype(np.float32) + 1))
    a = cache.sample(ids, 64, generator=torch.Generator().manual_seed(3))[0]
    assert not torch.equal(a, cache.sample(ids, 64, generator=torch.Generator().manual_seed(3))[0])   # the draw counter advances
    with pytest.raises(ValueError, match="fewer than"):
        cache.sample(torch.tensor([3]), 512)


def test_device_sampler_reproduces_the_reference_count_rule(tmp_path):
    """Golden G12 (counts the REFERENCE's unpack_sdf_samples returned, deep_sdf/data.py:74-110) through the product's loader +
    sampling kernel: DeviceSampleCache.from_files (NaN rows filtered, float64 files) -> dsdf_sample_batch.  Per scene: the
    reference's number of positives, then its number of negatives, no row twice, no NaN row."""
    import json, os
    from deepsdf_amd.data import DeviceSampleCache
    from tests.golden_io import GOLDEN, g12_scene_file
    g12 = json.load(open(os.path.join(GOLDEN, "g12_sample_counts.json")))["cases"]
    d = os.path.join(str(tmp_path), "SdfSamples")
    os.makedirs(d)
    for c in g12:
        g12_scene_file(os.path.join(d, c["id"] + ".npz"), c)
    cache = DeviceSampleCache.from_files(str(tmp_path), [c["id"] + ".npz" for c in g12], 3, "cuda")
    for k, c in enumerate(g12):
        assert (cache.n_pos[k], cache.n_neg[k]) == (c["n_pos"] - c["nan_pos"], c["n_neg"] - c["nan_neg"]), c["id"]
        if cache.n_pos[k] + cache.n_neg[k] < 2 * (c["subsample"] // 2):
            with pytest.raises(ValueError, match="fewer than"):     # the reference returns a short, un-collatable sample here
                cache.sample(torch.tensor([k]), c["subsample"])
            continue
        xyz, sdf = cache.sample(torch.tensor([k]), c["subsample"], key=1234 + k)
        xyz, sdf = xyz.cpu(), sdf.cpu()
        assert sdf.shape[0] == c["rows_returned"] and not bool(torch.isnan(sdf).any()), c["id"]
        n_p = int((sdf > 0).sum())
        assert (n_p, sdf.shape[0] - n_p) == (c["pos_rows"], c["neg_rows"]), c["id"]
        assert bool((sdf[:n_p] > 0).all()) and bool((xyz[:n_p, 2] == 1).all()) and bool((xyz[n_p:, 2] == -1).all()), c["id"]
        ip, iq = xyz[:n_p, 0].long(), xyz[n_p:, 0].long()
        assert ip.unique().numel() == n_p and iq.unique().numel() == sdf.shape[0] - n_p, c["id"]      # without replacement
        assert int(ip.min()) >= c["nan_pos"] and int(iq.min()) >= c["nan_neg"], c["id"]              # NaN rows are gone
        assert torch.equal(sdf[:n_p], ip.float() + 1) and torch.equal(sdf[n_p:], -(iq.float() + 1)), c["id"]   # rows stay intact


def test_staged_sample_cache_draws_the_resident_cache_batches():
    """StagedSampleCache (scenes in pinned host memory, only the batches in flight on the device; datasets beyond the HBM budget)
    must hand out EXACTLY the batches DeviceSampleCache does for the same keys -- over more draws than it has windows (reuse),
    with a scene repeated inside a batch, and through make_sample_cache's budget switch."""
    import numpy as np
    from deepsdf_amd import data
    rng = np.random.default_rng(3)
    scenes = []
    for k in range(7):
        n_pos, n_neg = int(rng.integers(150, 400)), int(rng.integers(150, 400))
        pos = np.concatenate([rng.uniform(-1, 1, (n_pos, 3)), rng.uniform(0.01, 0.5, (n_pos, 1))], 1).astype(np.float32)
        neg = np.concatenate([rng.uniform(-1, 1, (n_neg, 3)), -rng.uniform(0.01, 0.5, (n_neg, 1))], 1).astype(np.float32)
        scenes.append((torch.from_numpy(pos), torch.from_numpy(neg)))
    res = data.DeviceSampleCache(scenes, 3, "cuda")
    stg = data.StagedSampleCache(scenes, 3, "cuda", max_batch_scenes=3, depth=2)
    g1, g2 = torch.Generator(device="cuda"), torch.Generator(device="cuda")
    g1.manual_seed(11); g2.manual_seed(11)
    batches = [[0, 1, 2], [6, 3, 3], [5, 0, 4], [2, 2, 6], [1, 5, 3], [4, 6, 0]]
    outs = []
    for b in batches:                                   # several draws in flight before anything is read back
        ids = torch.tensor(b)
        outs.append((res.sample(ids, 100, generator=g1), stg.sample(ids, 100, generator=g2)))
    torch.cuda.synchronize()
    for b, ((xr, sr), (xs, ss)) in zip(batches, outs):
        assert torch.equal(xr, xs) and torch.equal(sr, ss), b
        assert xs.shape == (300, 3) and int((ss[:50] > 0).sum()) == 50 and int((ss[50:100] < 0).sum()) == 50
    assert stg.uploaded_bytes > 0 and stg.windows[0].shape[0] < res.data.shape[0]      # a window is smaller than the dataset
    with pytest.raises(ValueError, match="sized for"):
        stg.sample(torch.tensor([0, 1, 2, 3]), 100)
    with pytest.raises(IndexError):
        stg.sample(torch.tensor([7]), 100)
