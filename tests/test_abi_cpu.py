"""CPU (no GPU needed): the C-ABI library builds, loads, and exports every symbol include/dsdf.h declares;
host-side layout logic agrees with the library.  No compute call is made."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from deepsdf_amd.build import build_library
    build_library()
    from deepsdf_amd import _lib
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dsdf.h")).read()
    names = set(re.findall(r"\b(dsdf_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dsdf.h but not exported"
    from deepsdf_amd._lib import PROTOTYPES
    assert names == set(PROTOTYPES) | {"dsdf_last_error"}


def test_layout_and_sizes(lib):
    from deepsdf_amd import _lib
    from deepsdf_amd.net import NetSpec
    spec = NetSpec(256, [512] * 8, 3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)),
                   latent_in=[4], weight_norm=True)
    assert spec.n_params == 1843195 and spec.w_mac == 1835520
    assert spec.out_dim[3] == 253 and spec.in_dim[4] == 512
    net = spec.c_struct()
    lay = _lib.DsdfParamLayout()
    assert lib.dsdf_param_layout(C.byref(net), C.byref(lay)) == 0
    assert lay.total == spec.n_params
    for p in spec.params:
        off = {"bias": lay.bias_off, "g": lay.g_off, "v": lay.v_off, "weight": lay.v_off}[p.kind][p.layer]
        assert off == p.offset, p.name
    b = C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(net), 16384, 64, C.byref(b)) == 0
    assert 300e6 < b.value < 900e6
    assert lib.dsdf_decode_workspace_bytes(C.byref(net), 16384, C.byref(b)) == 0
    assert b.value < 200e6


def test_gradient_buckets(lib):
    """dsdf_grad_buckets: where the K-bucket data-parallel backward (DsdfLossCfg.dw_phase / dw_buckets) cuts the layers and the
    arena; dsdf_dw_phase_supported; dsdf_workspace_bytes_buckets."""
    from deepsdf_amd.net import NetSpec
    spec = NetSpec(256, [512] * 8, 3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4], weight_norm=True)
    net = spec.c_struct()
    first, off = (C.c_int32 * 2)(), (C.c_int64 * 3)()
    assert lib.dsdf_grad_buckets(C.byref(net), 2, first, off) == 0
    assert list(first) == [4, 0] and off[0] == spec.n_params and off[2] == 0
    assert off[1] == min(p.offset for p in spec.params if p.layer == 4) == sum(p.numel for p in spec.params if p.layer < 4)
    assert 0.35 < off[1] / spec.n_params < 0.65                                     # two buckets of comparable size
    first, off = (C.c_int32 * 4)(), (C.c_int64 * 5)()
    assert lib.dsdf_grad_buckets(C.byref(net), 4, first, off) == 0
    assert list(first) == [6, 4, 2, 0]
    assert list(off) == [spec.n_params] + [sum(p.numel for p in spec.params if p.layer < k) for k in (6, 4, 2, 0)]
    plain = NetSpec(5, [48] * 3, 3)                                                 # no weight norm: weight first, then bias
    first, off = (C.c_int32 * 2)(), (C.c_int64 * 3)()
    assert lib.dsdf_grad_buckets(C.byref(plain.c_struct()), 2, first, off) == 0
    assert first[0] == 2 and off[1] == min(p.offset for p in plain.params if p.layer == 2)
    first, off = (C.c_int32 * 8)(), (C.c_int64 * 9)()                               # 4 layers, 8 buckets: empty buckets repeat an offset
    assert lib.dsdf_grad_buckets(C.byref(plain.c_struct()), 8, first, off) == 0
    assert list(first) == [3, 3, 2, 2, 1, 1, 0, 0] and off[0] == plain.n_params and off[8] == 0
    assert all(off[b + 1] <= off[b] for b in range(8)) and off[1] == off[2]
    assert lib.dsdf_grad_buckets(C.byref(plain.c_struct()), 2, None, off) == -1
    assert lib.dsdf_grad_buckets(C.byref(plain.c_struct()), 1, first, off) == -1
    assert lib.dsdf_grad_buckets(C.byref(plain.c_struct()), 9, first, off) == -1
    assert lib.dsdf_dw_phase_supported(C.byref(net)) == 1
    wide = NetSpec(8, [640] * 3, 3)                                                 # wider than the fused kernels take
    assert lib.dsdf_dw_phase_supported(C.byref(wide.c_struct())) == 0
    b2, b8 = C.c_size_t(), C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(net), 16384, 64, C.byref(b2)) == 0
    assert lib.dsdf_workspace_bytes_buckets(C.byref(net), 16384, 64, 2, C.byref(b8)) == 0 and b8.value == b2.value
    assert lib.dsdf_workspace_bytes_buckets(C.byref(net), 16384, 64, 8, C.byref(b8)) == 0 and b8.value > b2.value
    assert lib.dsdf_workspace_bytes_buckets(C.byref(net), 16384, 64, 9, C.byref(b8)) == -1


def test_invalid_nets_rejected(lib):
    from deepsdf_amd.net import NetSpec
    with pytest.raises(NotImplementedError):
        NetSpec(4, [32] * 2, 3, xyz_in_all=True, forward_bf16=True)
    with pytest.raises(NotImplementedError, match="output layer"):      # config 5's kernels: the output layer sees activations only
        NetSpec(4, [32] * 2, 3, latent_in=[2], forward_bf16=True)
    last_skip = NetSpec(4, [32] * 2, 3, latent_in=[2]).c_struct()          # fine in fp32 ...
    bq = C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(last_skip), 8, 1, C.byref(bq)) == 0
    last_skip.fwd_bf16 = 1                                                  # ... refused by the library too
    assert lib.dsdf_workspace_bytes(C.byref(last_skip), 8, 1, C.byref(bq)) == -1 and b"output layer" in lib.dsdf_last_error()
    ln = NetSpec(4, [32] * 2, 3, norm_layers=[0, 2], weight_norm=False)    # LayerNorm variant: bn modules, also the unused last one
    assert [p.name for p in ln.params] == ["lin0.weight", "lin0.bias", "bn0.weight", "bn0.bias", "lin1.weight", "lin1.bias",
                                           "lin2.weight", "lin2.bias", "bn2.weight", "bn2.bias"]
    both = ln.c_struct()
    both.weight_norm_mask = 1
    b1 = C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(both), 8, 1, C.byref(b1)) == -1 and b"exclude" in lib.dsdf_last_error()
    x = NetSpec(4, [32] * 3, 3, xyz_in_all=True, latent_in=[2])           # deep_sdf_decoder.py:42-48 layer arithmetic
    assert x.out_dim == [29, 25, 29, 1] and x.in_dim == [7, 32, 32, 32]
    bad = x.c_struct()
    bad.xyz_in_all = 0                                                    # widths no longer add up without the xyz columns
    b0 = C.c_size_t()
    assert lib.dsdf_workspace_bytes(C.byref(bad), 8, 1, C.byref(b0)) == -1 and b"in_dim" in lib.dsdf_last_error()
    spec = NetSpec(4, [32, 32, 32], 3, latent_in=[1, 2])
    b = C.c_size_t()
    net = spec.c_struct()
    assert lib.dsdf_workspace_bytes(C.byref(net), 8, 1, C.byref(b)) == -1
    assert b"latent_in" in lib.dsdf_last_error()


def test_param_names_match_oracle():
    from deepsdf_amd.net import NetSpec
    from oracle import deepsdf_oracle as orc
    kw = dict(dims=[64] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=[0, 1, 2, 3], latent_in=[2],
              weight_norm=True, geom_dimension=3)
    assert [p.name for p in NetSpec(4, **kw).params] == orc.param_names(orc.make_net(4, **kw))
    from deepsdf_amd.net import dropout_layer_key
    assert dropout_layer_key(1234, 5, 3) == orc.dropout_layer_key(1234, 5, 3)
    assert dropout_layer_key((1 << 40) + 7, (1 << 33) + 1, 0) == orc.dropout_layer_key((1 << 40) + 7, (1 << 33) + 1, 0)


def test_size_queries_survive_degenerate_batches(lib):
    from deepsdf_amd.net import NetSpec
    net = NetSpec(4, [32, 32], 3, norm_layers=[0, 1], weight_norm=True).c_struct()
    b = C.c_size_t()
    for n in (0, 1, 63, 64, 65):
        assert lib.dsdf_workspace_bytes(C.byref(net), n, 1 if n else 0, C.byref(b)) == 0 and b.value > 0
        assert lib.dsdf_decode_workspace_bytes(C.byref(net), n, C.byref(b)) == 0
    assert lib.dsdf_workspace_bytes(C.byref(net), -1, 0, C.byref(b)) == -1


def test_argument_errors_are_reported_before_any_launch(lib):
    """Entry points validate their arguments on the host and fail with a message (no GPU needed to see that)."""
    from deepsdf_amd import _lib
    from deepsdf_amd.net import NetSpec
    lib.dsdf_last_error.restype = C.c_char_p
    assert lib.dsdf_sample_batch(None, 3, None, None, None, None, None, 4, 64, 1, None, None, None) == -1
    assert b"NULL" in lib.dsdf_last_error()
    dummy = C.c_void_p(256)     # never dereferenced: the shape checks come first
    assert lib.dsdf_sample_batch(dummy, 0, dummy, dummy, dummy, dummy, dummy, 4, 64, 1, dummy, dummy, None) == -1
    assert b"geom_dim" in lib.dsdf_last_error()
    assert lib.dsdf_sample_batch(dummy, 3, dummy, dummy, dummy, dummy, dummy, 4, 1, 1, dummy, dummy, None) == -1   # 2*(1//2) == 0 rows
    # the bf16 forward exists for widths <= 512 only, and the single-code decode needs the fp32 fused forward
    wide = NetSpec(8, [640, 640], 3, forward_bf16=True).c_struct()
    n = C.c_int64()
    assert lib.dsdf_packed_floats(C.byref(wide), C.byref(n)) == -1 and b"fwd_bf16" in lib.dsdf_last_error()
    net = NetSpec(8, [64, 64], 3).c_struct()
    assert lib.dsdf_decode_latent(C.byref(net), None, None, None, None, 10, None, None, 0, None) == -1
    # dsdf_decode_latent_supported is the ONE definition of "the single-code decode takes this net" (decode_sdf asks it)
    big = NetSpec(256, [512] * 8, 3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4], weight_norm=True)
    assert lib.dsdf_decode_latent_supported(C.byref(big.c_struct())) == 1
    assert lib.dsdf_decode_latent_supported(C.byref(net)) == 1
    for kw in (dict(norm_layers=[0, 1], weight_norm=False),             # LayerNorm
               dict(xyz_in_all=True), dict(latent_dropout=True)):       # the other two layer-by-layer variants
        assert lib.dsdf_decode_latent_supported(C.byref(NetSpec(8, [64, 64], 3, **kw).c_struct())) == 0, kw
    assert lib.dsdf_decode_latent_supported(C.byref(NetSpec(8, [640, 640], 3).c_struct())) == 0       # wider than the fused kernels
    assert lib.dsdf_decode_latent_supported(C.byref(NetSpec(8, [64], 3).c_struct())) == 0             # one hidden layer
    assert lib.dsdf_decode_latent_supported(C.byref(NetSpec(8, [64, 64], 5).c_struct())) == 0         # geom_dim > 4
    broken = NetSpec(8, [64, 64], 3).c_struct()
    broken.in_dim[1] = 63
    assert lib.dsdf_decode_latent_supported(C.byref(broken)) == -1 and b"in_dim" in lib.dsdf_last_error()
    # an xyz_in_all net's d/d(xyz) scratch rows hold 4 floats: more geometry columns are refused, on both sides of the ABI
    x5 = NetSpec(8, [64, 64], 4, xyz_in_all=True).c_struct()
    x5.geom_dim, x5.in_dim[0] = 5, 13
    x5.in_dim[1], x5.out_dim[0] = 64, 59
    assert lib.dsdf_packed_floats(C.byref(x5), C.byref(n)) == -1 and b"xyz_in_all needs geom_dim" in lib.dsdf_last_error()
    with pytest.raises(NotImplementedError, match="geom_dimension <= 4"):
        NetSpec(8, [64, 64], 5, xyz_in_all=True)
    # gemm_split: widths <= 512, not together with the bf16 forward or the layer-by-layer variants; its planes enlarge `packed`
    sp = NetSpec(8, [64, 64], 3, gemm_split=True).c_struct()
    plain = NetSpec(8, [64, 64], 3, gemm_split=False).c_struct()
    n2 = C.c_int64()
    assert lib.dsdf_packed_floats(C.byref(sp), C.byref(n)) == 0 and lib.dsdf_packed_floats(C.byref(plain), C.byref(n2)) == 0
    assert n.value > n2.value
    sp.fwd_bf16 = 1                      # together with the bf16 forward: allowed (the backward chain is the split one)
    assert lib.dsdf_packed_floats(C.byref(sp), C.byref(n)) == 0
    sp.fwd_bf16, sp.latent_dropout = 0, 1
    assert lib.dsdf_packed_floats(C.byref(sp), C.byref(n)) == -1 and b"gemm_split" in lib.dsdf_last_error()
    with pytest.raises(NotImplementedError):
        NetSpec(8, [640, 640], 3, gemm_split=True)
    with pytest.raises(NotImplementedError):
        NetSpec(8, [64, 64], 3, gemm_split=True, latent_dropout=True)


def test_warm_own_code_bound_is_inside_the_text_section(tmp_path):
    """common.hpp warm_own_code reads the kernel's own instructions as data, clamped to the address of
    dsdf_text_end_marker.  Pinned here on the gfx950 code object inside libdsdf_hip.so: the marker lies INSIDE .text (no read
    can leave the section) and BEHIND every kernel that warms itself (so each of them has room to cover its own code)."""
    import re
    import shutil
    import subprocess
    from deepsdf_amd.build import LIB
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = {t: os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")}
    if not all(os.path.exists(p) for p in tools.values()) or not os.path.exists(LIB):
        pytest.skip("ROCm LLVM tools or the built library are not available")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "co.elf")
    subprocess.run([tools["llvm-objcopy"], "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    subprocess.run([tools["clang-offload-bundler"], "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    sec = subprocess.run([tools["llvm-readelf"], "-SW", co], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+[0-9a-f]+\s+([0-9a-f]+)", sec)
    text_lo, text_hi = int(m.group(1), 16), int(m.group(1), 16) + int(m.group(2), 16)
    syms = subprocess.run([tools["llvm-readelf"], "-sW", co], capture_output=True, text=True, check=True).stdout
    funcs = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            funcs[f[7]] = (int(f[1], 16), int(f[2]))
    marker = [v for k, v in funcs.items() if "dsdf_text_end_marker" in k]
    assert len(marker) == 1
    mk = marker[0][0]
    assert text_lo <= mk < text_hi
    warmers = [k for k in funcs if re.search(r"fused_(forward|backward|fwd_bwd)_kernel", k) and "bf16" not in k]
    assert len(warmers) == 3
    for k in warmers:
        addr, size = funcs[k]
        assert addr + size <= mk, (k, hex(addr + size), hex(mk))
    shutil.rmtree(str(tmp_path), ignore_errors=True)


def test_split_wait_window_of_the_dw_split_kernel_is_clean():
    """dwstream.hpp dw_block_split keeps eight ds_read_b128 in flight across compiler-scheduled code (asm statement 1 ends in
    lgkmcnt(8), statement 2 is the lgkmcnt(0)).  Correct only if nothing in between touches their destination registers -- a
    register-allocation outcome, so it is checked on the gfx950 code object of the library that was just built (the build does the
    same and refuses a violating library), and the checker itself is checked on hand-made instruction streams."""
    from deepsdf_amd import asmcheck
    from deepsdf_amd.build import LIB
    reads = [("ds_read_b128", f"v[{16 + 4 * k}:{19 + 4 * k}], v1 offset:{1024 * k}") for k in range(16)]
    body = [("v_and_b32_e32", "v0, 0xffff0000, v16"), ("v_sub_f32_e32", "v1, v16, v0"), ("v_perm_b32", "v2, v20, v24, s5")]
    close = [("s_waitcnt", "lgkmcnt(0)")]
    ok = reads + [("s_waitcnt", "lgkmcnt(8)")] + body + close + [("v_mov_b32_e32", "v3, v48")]
    assert asmcheck.check_split_wait_windows(ok) == 1                     # reads of A registers (v16..v47) are fine; B = v48..v79
    for bad in (("v_mov_b32_e32", "v3, v50"), ("v_accvgpr_write_b32", "a7, v79"), ("v_pk_mul_f32", "v[4:5], v[78:79], v[8:9]"),
                ("scratch_store_dwordx4", "off, v[4:7], off offset:16"), ("v_mov_b32_e32", "v48, v3")):
        with pytest.raises(asmcheck.AsmHazard):
            asmcheck.check_split_wait_windows(reads + [("s_waitcnt", "lgkmcnt(8)")] + body + [bad] + close)
    with pytest.raises(asmcheck.AsmHazard):                                # a window that is never closed
        asmcheck.check_split_wait_windows(reads + [("s_waitcnt", "lgkmcnt(8)")] + body)
    # the woven block leaves all sixteen reads in flight: A registers (v16..v47) are protected too; MFMAs on other registers may run
    mf = ("v_mfma_f32_32x32x16_bf16", "a[0:15], v[100:103], v[104:107], a[0:15]")
    assert asmcheck.check_split_wait_windows(reads + [mf, ("v_sub_f32_e32", "v2, v3, v4")] + close) == 1
    with pytest.raises(asmcheck.AsmHazard):
        asmcheck.check_split_wait_windows(reads + [mf, ("v_and_b32_e32", "v0, 0xffff0000, v16")] + close)
    assert asmcheck.check_split_wait_windows(reads + close) == 0             # reads + wait in one statement: no window at all
    assert asmcheck.vgprs("a[0:15], v[4:7], v9, s[0:3], 0xff, v12 offset:16") == {4, 5, 6, 7, 9, 12}
    if not asmcheck.tools_available() or not os.path.exists(LIB):
        pytest.skip("ROCm LLVM tools or the built library are not available")
    assert asmcheck.check_library(LIB) >= 1


def test_bf16x8_kloop_never_reloads_the_sources_of_the_mfma_in_front():
    """fused_bf16x8.hpp runs two waves per SIMD; its first k-loop let the register allocator hand a ds_read / buffer_load the registers
    the MFMA straight in front of it reads as srcA / srcB, and 3 % of the rows differed from run to run (profiles/r02_lab_bf16x8_race.log;
    on the code object of that version, rebuilt from commit 5f67ff3: 24 + 22 such loads, profiles/r03_bf16x8_reuse_audit.log).  The
    rewritten loop keeps a full step of MFMAs between an MFMA and any load into its source registers -- a property of the EMITTED code,
    so it is measured there (the build does the same and refuses a library that violates it); the measure itself is checked on
    hand-made streams."""
    from deepsdf_amd import asmcheck
    from deepsdf_amd.build import LIB
    mk = lambda op, args, addr, target=None: dict(op=op, args=args, addr=addr, target=target)   # noqa: E731
    mf = lambda a, acc, sa, sb: mk("v_mfma_f32_32x32x16_bf16", f"v[{acc}:{acc + 15}], v[{sa}:{sa + 3}], v[{sb}:{sb + 3}], v[{acc}:{acc + 15}]", a)   # noqa: E731
    bad = [mf(0, 0, 100, 104), mk("ds_read_b128", "v[104:107], v9", 8)]                                     # reload right behind the reader
    assert asmcheck.mfma_src_reuse_distances(bad) == {1: 0}
    good = [mf(0, 0, 100, 104), mf(8, 16, 108, 112), mf(16, 32, 116, 120), mk("buffer_load_dwordx4", "v[100:103], v9, s[8:11], s2 offen", 24)]
    assert asmcheck.mfma_src_reuse_distances(good) == {3: 2}
    drained = [mf(0, 0, 100, 104), mk("v_readfirstlane_b32", "s3, v15", 8), mk("ds_read_b128", "v[104:107], v9", 16)]   # a VALU read of the result
    assert asmcheck.mfma_src_reuse_distances(drained) == {}
    loop = [mk("ds_read_b128", "v[104:107], v9", 0), mf(8, 0, 100, 104), mk("s_cbranch_scc1", "65533", 16, target=0)]  # across the back-edge
    assert asmcheck.mfma_src_reuse_distances(loop) == {0: 0} and asmcheck.mfma_src_reuse_distances(loop, fall_through_only=True) == {}
    assert asmcheck.mfma_src_reuse_distances(loop, edges="loops") == {0: 0}       # the build gate follows loop back-edges: the steady state
    # ... but not a FORWARD branch over a guarded step's MFMAs (the path only exists together with the guard that skips the load too)
    guarded = [mf(0, 0, 100, 104), mk("s_cbranch_vccnz", "2", 8, target=32), mf(16, 16, 108, 112), mf(24, 32, 116, 120),
               mk("ds_read_b128", "v[104:107], v9", 32)]
    assert asmcheck.mfma_src_reuse_distances(guarded, edges="loops") == {4: 2} and asmcheck.mfma_src_reuse_distances(guarded) == {4: 0}
    if not asmcheck.tools_available() or not os.path.exists(LIB):
        pytest.skip("ROCm LLVM tools or the built library are not available")
    worst, pairs = asmcheck.check_mfma_src_reuse(LIB, min_distance=2)
    assert worst is not None and worst >= 2 and pairs > 20
