"""ML_MATH_F32X3 -- fp32 tensors, every product of the dense convs as three f16 MFMAs on operands split into two halves
(x = hi + 2^-11 lo, 22 bits; fp32 accumulation) -- held to the bars of the fp32 path, unchanged: the end-to-end tests of
tests/test_gpu_model.py run again under this mode (indices bit-exact, in order; floats within 1e-3 of the oracle), the op
tests of tests/test_gpu_ops.py are parametrised over it (same absolute tolerances), and here: its error against an fp64
convolution of the same fp32 operands is not larger than that of the exact-product fp32 MFMA path, at three input scales;
the documented operand range; which kernels a forward in this mode runs on.  -m gpu."""
import importlib.util
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import tfops as T

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name + "_under_f32x3", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module", autouse=True)
def _mode():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from masklab_hip import _lib, ops
    _lib.check(_lib.load().ml_device_check(), "ml_device_check")
    ops.set_conv_math("f32x3")
    yield
    ops.set_conv_math("f32")


@pytest.fixture(scope="module")
def M():
    return _load("test_gpu_model")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


# ------------------------------------------------------------------ the model-level bars of the fp32 path, unchanged
@pytest.mark.parametrize("bt", ["mobilenet", "resnext50", "resnext101"])
def test_full_forward_matches_oracle(M, bt):
    M.test_full_forward_matches_oracle(bt)


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50"])
def test_full_forward_with_detections(M, bt):
    M.test_full_forward_with_detections(bt)


@pytest.mark.parametrize("case", ["forward_mobilenet_128", "forward_resnext50_128"])
def test_forward_matches_committed_golden(M, case, golden_dir):
    M.test_forward_matches_committed_golden(case, golden_dir)


@pytest.mark.parametrize("bt,size", [("resnext50", 1024), ("resnext101", 1280)])
def test_headline_size_indices_bit_exact(M, bt, size):
    M.test_headline_size_indices_bit_exact(bt, size)


def test_batch_sharding_equals_full_batch(M):
    M.test_batch_sharding_equals_full_batch()
    M.test_batch_sharding_where_the_split_k_decision_differs()


@pytest.mark.parametrize("bt", ["mobilenet", "resnext50"])
def test_hipgraph_replay_equals_eager(M, bt):
    M.test_hipgraph_replay_equals_eager(bt)


@pytest.mark.parametrize("bt,shape,thr", [("mobilenet", (3, 128, 256, 3), 0.5), ("resnext50", (2, 192, 160, 3), 0.5)])
def test_fixed_capacity_stage2_matches_oracle(M, bt, shape, thr):
    M.test_fixed_capacity_stage2_matches_oracle(bt, shape, thr)


def test_which_kernels_a_forward_runs_on(M):
    """Every dense conv of the forward runs the split-operand kernel (no silent fp32 launches, no pipelined-fp32 1x1);
    grouped 3x3, depthwise and the fused mask-head tail keep their fp32 kernels."""
    from masklab_hip import ops
    cfg, model, w = M._build("resnext50", seed=5, hot_cls=True)
    images = np.random.default_rng(99).integers(0, 256, (2, 128, 256, 3), dtype=np.uint8)
    model.predict(images)
    ops.PROFILE = []
    model.predict(images)
    names, ops.PROFILE = sorted({r["kernel"] for r in ops.PROFILE}), None
    conv = [n for n in names if n.startswith(("conv_mfma", "conv1x1"))]
    assert conv and all(n.endswith("_x3") for n in conv), names
    assert any(n.startswith("gconv3x3") for n in names)


# ------------------------------------------------------------------ not a reduced precision: error against fp64
@pytest.mark.parametrize("k,cin,cout,hw,stride,res,scale", [
    (3, 64, 96, (19, 23), 1, False, 1.0), (1, 256, 128, (24, 24), 1, True, 1.0), (3, 256, 256, (32, 32), 1, False, 1.0),
    (3, 128, 128, (33, 33), 2, False, 1.0), (1, 2048, 256, (16, 16), 1, False, 1.0),
    (3, 256, 256, (32, 32), 1, False, 1e-3), (3, 256, 256, (32, 32), 1, False, 1e3), (3, 256, 256, (16, 16), 1, False, 1e-6)])
def test_error_against_fp64_not_above_the_exact_fp32_product_path(k, cin, cout, hw, stride, res, scale):
    """Same fp32 operands through both modes, compared with the oracle's fp64 convolution.  The split drops at most
    2^-22 |a b| per product; the fp32 accumulation (both modes) rounds at 2^-24 of the running sum per step, which
    dominates -- measured, the split-operand path is the CLOSER one (a 16-deep MFMA step adds 16 exact products before it
    rounds; the fp32 instruction rounds after every 2).  Asserted: rms and max error <= 1.05 x the fp32 path's, for
    operands in the normal range of a half (2^-14 <= |x| < 65520: inputs x 1e-3, x 1, x 1e3); the last case puts every
    activation BELOW it and checks the documented absolute floor instead."""
    from masklab_hip import _lib, ops, packing
    rng = np.random.default_rng(7)
    x = (rng.normal(size=(2, hw[0], hw[1], cin)) * scale).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = (rng.normal(size=(cout,)) * scale).astype(np.float32)
    ref = T.conv2d(x.astype(np.float64), w, b, stride, "same")
    r = None
    if res:
        r = (rng.normal(size=ref.shape) * scale).astype(np.float32)
        ref = ref + r
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    err = {}
    for mode in ("f32", "f32x3"):
        ops.set_conv_math(mode)
        got = host(ops.conv2d(dev(x), dc, stride=stride, padding="same", residual=None if r is None else dev(r)))
        e = got.astype(np.float64) - ref
        err[mode] = (float(np.sqrt((e * e).mean())), float(np.abs(e).max()))
    ops.set_conv_math("f32x3")
    if scale < 2.0 ** -14:
        # activations below the smallest normal half: their low halves are subnormals, 2^-36 ABSOLUTE per activation
        # (negligible beside any product of normal-range operands, but no longer relative) -- the documented floor
        floor = 2.0 ** -36 * float(np.abs(w.astype(np.float64)).sum((0, 1, 2)).max())
        assert err["f32x3"][1] <= 1.05 * err["f32"][1] + floor, (err, floor)
        return
    assert err["f32x3"][0] <= 1.05 * err["f32"][0], err
    assert err["f32x3"][1] <= 1.05 * err["f32"][1], err
    assert err["f32x3"][1] <= 1e-6 * float(np.abs(ref).max()), err


def test_operand_range_and_special_values():
    """The split keeps 22 bits wherever |x| < 65520 (include/masklab_hip.h): powers of two, values with all 24 bits set,
    tiny values (below 2^-14 the low half is a subnormal: 2^-36 absolute), zeros and mixed signs against fp64, one
    input channel at a time so that a product's error is not averaged away."""
    from masklab_hip import ops, packing
    vals = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.1415927, -2.7182817, 32767.998, -32000.123,
                     1e-3, -1.2345678e-4, 6.1e-5, 5.9e-8, 2.0 ** -20 * 1.9999999, 1024.0009765625, 255.99998, 1e-7, -3e-8, 12345.678,
                     0.1, 0.2, 0.3, 0.7, 1.9999999, 0.99999994, 7.0, 65.0, 4097.0, 65000.5, 2.0 ** -14, 2.0 ** -14 * 1.0000001],
                    np.float32)
    cin = 32
    assert vals.size == cin
    x = np.zeros((1, 8, 16, cin), np.float32)                     # pixel p carries vals[p % 32] in channel p % 32 only
    for p in range(128):
        x[0, p // 16, p % 16, p % cin] = vals[p % cin]
    wv = np.array([1.0, -1.0, 0.33333334, 1.0 + 2.0 ** -23, 3.0e-3, -7.7777777, 1.9999999, 2.0 ** -12], np.float32)
    w = np.zeros((1, 1, cin, 32), np.float32)
    for o in range(32):
        w[0, 0, :, o] = wv[o % 8] * (1.0 + o // 8 * 2.0 ** -20)
    ref = T.conv2d(x.astype(np.float64), w, None, 1, "valid")
    got = host(ops.conv2d(dev(x), ops.DeviceConv(packing.pack_dense(w, None), "cuda"), padding="valid")).astype(np.float64)
    assert np.isfinite(got).all()
    # one product a b per output: |error| <= (3 x 2^-22 + 2^-24) |a b| -- two split operands, the dropped lo x lo term, the
    # fp32 store -- plus, for operands below 2^-14 (whose low half is a subnormal), 2^-36 absolute per operand
    a = np.abs(x.reshape(128, cin).astype(np.float64))                                   # [pixel, channel]
    bw = np.abs(w[0, 0].astype(np.float64))                                             # [channel, out]
    bound = (np.abs(ref.reshape(128, 32)) * (3 * 2.0 ** -22 + 2.0 ** -24) +
             2.0 ** -36 * ((a > 0) @ bw + a @ (bw > 0)))
    err = np.abs(got.reshape(128, 32) - ref.reshape(128, 32))
    assert np.all(err <= bound), float(np.max(err / np.maximum(bound, 1e-300)))


@pytest.mark.parametrize("k,cin,cout,shape,res,dil,n_small", [
    (3, 64, 128, (4, 128, 128), False, 1, 3),  # 256 tiles of 256 rows: the smallest launch that takes them; 18 chunks
    (1, 32, 128, (4, 128, 128), False, 1, 3),  # ONE chunk: the ring's second request goes through an empty resource
    (1, 320, 256, (4, 128, 128), True, 1, 1),  # ten chunks, two N tiles, residual (read in the epilogue)
    (1, 96, 128, (5, 120, 111), False, 1, 3),  # three chunks, ragged last tile (66 600 rows)
    (3, 32, 128, (3, 160, 150), False, 2, 2),  # dilated taps, out-of-image rows through the out-of-range rule
])
def test_256_row_pipelined_tile(k, cin, cout, shape, res, dil, n_small):
    """Launches that fill the chip with 256 x 128 tiles run the 8-wave form of the kernel (3-deep ring, counted waits, the
    next step's fragments read and split under the current step's MFMAs).  Its results equal the 128-row kernel's BIT FOR
    BIT (the first n_small images in a launch too small for the big tile, yet not so small that the library would cut
    its K sum) and the oracle's within the op tolerance."""
    from masklab_hip import _lib, ops, packing
    rng = np.random.default_rng(5)
    B, H, W = shape
    x = rng.normal(size=(B, H, W, cin)).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    r = rng.normal(size=(B, H, W, cout)).astype(np.float32) if res else None
    dc = ops.DeviceConv(packing.pack_dense(w, b, tile=1), "cuda")
    lib = _lib.load()
    import ctypes as C

    def run(n):
        xd, rd = dev(x[:n]), (None if r is None else dev(r[:n]))
        d, _, _ = ops._conv_desc(xd, dc, 1, "same", dil, _lib.ACT_RELU, rd)
        mt = lib.ml_conv2d_launch_mtile(C.byref(d), 1, 1)
        return mt, host(ops.conv2d(xd, dc, padding="same", dilation=dil, act=_lib.ACT_RELU, residual=rd))

    mt_big, big = run(B)
    mt_small, small = run(n_small)
    assert (mt_big, mt_small) == (256, 128)
    np.testing.assert_array_equal(big[:n_small], small)
    ref = T.conv2d(x[B - 1:].astype(np.float64), w, b, 1, "same", dil)
    if res:
        ref = ref + r[B - 1:]
    np.testing.assert_allclose(big[B - 1:], T.relu(ref), rtol=0, atol=2e-5)


def test_short_k_residual_1x1_runs_on_128x64_tiles():
    """The conv3 shapes of ResNeXt stages 1 / 2 (K <= 256, residual) on maps too small for the persistent kernel's rule (fewer
    than 4 096 pixels per image): HBM-bound, half a tile's time is its epilogue -- three 128 x 64 blocks per CU overlap
    epilogues and K loops.  Same values as any other tile shape; checked against the oracle."""
    from masklab_hip import _lib, ops, packing
    import ctypes as C
    rng = np.random.default_rng(12)
    x = rng.normal(size=(8, 48, 48, 128)).astype(np.float32)
    w = (rng.normal(size=(1, 1, 128, 256)) / np.sqrt(128)).astype(np.float32)
    b = rng.normal(size=(256,)).astype(np.float32)
    r = rng.normal(size=(8, 48, 48, 256)).astype(np.float32)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    xd, rd = dev(x), dev(r)
    d, _, _ = ops._conv_desc(xd, dc, 1, "valid", 1, _lib.ACT_RELU, rd)
    lib = _lib.load()
    assert not lib.ml_conv2d_uses_pipe(C.byref(d))
    assert lib.ml_conv2d_launch_ntile(C.byref(d), 1, 1) == 64 and lib.ml_conv2d_launch_mtile(C.byref(d), 1, 1) == 128
    got = host(ops.conv2d(xd, dc, padding="valid", act=_lib.ACT_RELU, residual=rd))
    ref = T.relu(T.conv2d(x.astype(np.float64), w, b, 1, "valid") + r)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    d2, _, _ = ops._conv_desc(xd, dc, 1, "valid", 1, _lib.ACT_RELU, None)            # no residual: the rule does not apply
    assert lib.ml_conv2d_launch_ntile(C.byref(d2), 1, 1) == 128


@pytest.mark.parametrize("cin,cout,shape,res,act", [
    (64, 256, (2, 96, 96), False, "relu"),      # two chunks, two N tiles of one group
    (128, 256, (2, 96, 96), True, "relu"),      # the stage-1 conv3: residual, four chunks
    (32, 128, (1, 64, 64), True, "none"),       # ONE chunk per tile: every chunk stores the previous tile
    (256, 128, (3, 70, 67), False, "relu6"),    # ragged last panel (14 070 rows), clamp at 6
    (512, 1024, (1, 64, 64), True, "relu"),     # eight N tiles: groups of tiles per block, 16 chunks
])
def test_short_k_1x1_on_the_persistent_kernel(cin, cout, shape, res, act):
    """K <= 512 1x1 convs of maps with >= 4 096 pixels (ResNeXt stages 1-2 at the bench sizes) run the persistent pipelined
    kernel with split-operand products (csrc/conv1x1_pipe.hip, f32x3_t): same three products per step and the same k order
    as the generic kernel, bias and residual added after the sum instead of before (one rounding placed differently).
    Against the fp64 oracle within the op tolerance, against the generic kernel (tile = 1) to 1e-5, and -- scheduling only --
    one image alone equals the same image inside the batch bit for bit."""
    from masklab_hip import _lib, ops, packing
    import ctypes as C
    rng = np.random.default_rng(21)
    B, H, W = shape
    x = rng.normal(size=(B, H, W, cin)).astype(np.float32)
    w = (rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    r = rng.normal(size=(B, H, W, cout)).astype(np.float32) if res else None
    a = {"relu": _lib.ACT_RELU, "relu6": _lib.ACT_RELU6, "none": _lib.ACT_NONE}[act]
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    dc1 = ops.DeviceConv(packing.pack_dense(w, b, tile=1), "cuda")
    xd, rd = dev(x), (None if r is None else dev(r))
    d, _, _ = ops._conv_desc(xd, dc, 1, "valid", 1, a, rd)
    assert _lib.load().ml_conv2d_uses_pipe(C.byref(d)) == 1
    ops.PROFILE = []
    got_d = ops.conv2d(xd, dc, padding="valid", act=a, residual=rd)
    names, ops.PROFILE = [q["kernel"] for q in ops.PROFILE], None
    assert names == ["conv1x1_pipe_x3"], names
    got = host(got_d)
    ref = T.conv2d(x.astype(np.float64), w, b, 1, "valid")
    if res:
        ref = ref + r
    ref = {"relu": T.relu, "relu6": lambda v: np.clip(v, 0.0, 6.0), "none": lambda v: v}[act](ref)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    generic = host(ops.conv2d(xd, dc1, padding="valid", act=a, residual=rd))
    np.testing.assert_allclose(got, generic, rtol=0, atol=1e-5)
    one = host(ops.conv2d(dev(x[B - 1:]), dc, padding="valid", act=a, residual=None if r is None else dev(r[B - 1:])))
    np.testing.assert_array_equal(one, got[B - 1:])
    for _ in range(5):              # counted waits / cross-tile requests: repeated launches give the same bits
        np.testing.assert_array_equal(host(ops.conv2d(xd, dc, padding="valid", act=a, residual=rd)), got)


def test_groupnorm_partials_from_the_256_row_tile():
    """ml_conv2d_desc.gn_partials from the 8-wave kernel: 4 pairs per 128-row tile like the 4-wave kernel writes (waves w and
    w + 4 added through LDS), so the GroupNorm apply pass is unchanged."""
    from masklab_hip import _lib, ops, packing
    import ctypes as C
    rng = np.random.default_rng(9)
    B, H, W = 4, 128, 128                                        # 256 tiles of 256 rows
    x = rng.normal(size=(B, H, W, 128)).astype(np.float32)
    w, b = (rng.normal(size=(3, 3, 128, 128)) * 0.03).astype(np.float32), rng.normal(size=(128,)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, 128).astype(np.float32), rng.normal(size=(128,)).astype(np.float32)
    dc = ops.DeviceConv(packing.pack_dense(w, b), "cuda")
    plain = ops.conv2d(dev(x), dc, act=_lib.ACT_RELU)
    want = host(ops.groupnorm_chunk(plain, dev(gamma), dev(beta), 16))
    part = torch.full((B * H * W // 128, 4, 2), float("nan"), dtype=torch.float64, device="cuda")
    xd = dev(x)
    d, _, _ = ops._conv_desc(xd, dc, act=_lib.ACT_RELU, gn_partials=part)
    assert _lib.load().ml_conv2d_launch_mtile(C.byref(d), 1, 1) == 256
    y = ops.conv2d(xd, dc, act=_lib.ACT_RELU, gn_partials=part)
    np.testing.assert_array_equal(host(y), host(plain))
    yh = host(y).astype(np.float64).reshape(-1, 128 * 128)
    np.testing.assert_allclose(host(part)[..., 0].sum(1), yh.sum(1), rtol=1e-7)
    np.testing.assert_allclose(host(part)[..., 1].sum(1), (yh * yh).sum(1), rtol=1e-6)
    (got,) = ops.groupnorm_chunk_multi([dict(x=y, gamma=dev(gamma), beta=dev(beta), groups=16, out=y, partials=(part, 32))])
    np.testing.assert_allclose(host(got), want, rtol=0, atol=2e-6)
    # three images of the same batch: 192 tiles of 256 rows -> the 4-wave kernel; the normalised outputs agree bit for bit
    part3 = torch.full((3 * H * W // 128, 4, 2), float("nan"), dtype=torch.float64, device="cuda")
    y3 = ops.conv2d(dev(x[:3]), dc, act=_lib.ACT_RELU, gn_partials=part3)
    (got3,) = ops.groupnorm_chunk_multi([dict(x=y3, gamma=dev(gamma), beta=dev(beta), groups=16, out=y3, partials=(part3, 32))])
    np.testing.assert_array_equal(host(got3), host(got)[:3])


def test_weights_are_split_once_on_the_host():
    """DeviceConv.wgt_x3: same bytes and strides as the fp32 packing; hi + 2^-11 lo reproduces every weight to 2^-22."""
    from masklab_hip import ops, packing
    rng = np.random.default_rng(3)
    w = (rng.normal(size=(3, 3, 40, 72)) * 0.05).astype(np.float32)
    dc = ops.DeviceConv(packing.pack_dense(w, None), "cuda")
    a, s = host(dc.wgt), host(dc.wgt_x3)
    assert a.shape == s.shape and a.dtype == s.dtype == np.float32
    h = s.view(np.float16).reshape(a.shape[0], -1, 64).astype(np.float64)
    back = (h[..., :32] + h[..., 32:] * 2.0 ** -11).reshape(a.shape)
    assert np.all(np.abs(back - a) <= np.maximum(np.abs(a) * 2.0 ** -22, 2.0 ** -36))      # (below 2^-14: the absolute floor)
