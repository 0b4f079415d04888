"""GPU: the "split3" mode of the frozen IMU_Net forward (mmego_amd/csrc/split3.hip; reference Net/IMU_Net.py:58-62,76-83) -- fp32
products on the bf16 matrix pipe: every fp32 operand split EXACTLY into three bf16 pieces, six piece products per product, fp32
accumulation.  Unlike the lossy bf16 mode this one is held to the fp32 path's own bars:

* the split itself, bit for bit, against a CPU restatement in torch (bf16 casts = round to nearest even), incl. zeros, subnormals,
  inf, NaN and values that overflow bf16; split -> join returns the input bit for bit in the exact range;
* the product kernel against float64 (its error must be of the native fp32 product's order) and against the 6-piece-product sum
  written out in float64;
* a whole BiLSTM(512) stack against the fp32 step-kernel path of this repository at fp32 rounding;
* IMU_Net's forward against the reference golden G7 and the CPU oracle at test_imu_forward's tolerances (R, t within 2e-5);
* one U+L training step at the benchmarked shape against the oracle at test_bench_shape's bars (joints <= 1e-3 cm).
"""
import numpy as np
import pytest
import torch

from conftest import golden, load_weights
from oracle import nets as on

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from mmego_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def _split_cpu(x):
    """The split of split3.hip in torch on the CPU: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); a1 inf / NaN -> no residual."""
    a1 = x.to(torch.bfloat16)
    f1 = a1.float()
    fin = torch.isfinite(f1)
    r1 = torch.where(fin, x - f1, torch.zeros_like(x))
    a2 = r1.to(torch.bfloat16)
    r2 = r1 - a2.float()
    a3 = r2.to(torch.bfloat16)
    return a1, a2, a3


def _pieces_from_sfrag(y, rows, K):
    """sfrag [Rp/32][K/16][3][64][8] -> three [rows, K] bf16 matrices (lane l = row l % 32, k half l / 32)."""
    nrb, SK = y.shape[0], y.shape[1]
    v = y.view(nrb, SK, 3, 2, 32, 8).permute(2, 0, 4, 1, 3, 5).reshape(3, nrb * 32, SK * 16)
    return v[0, :rows, :K], v[1, :rows, :K], v[2, :rows, :K]


def _bits(t):
    return t.contiguous().view(torch.int16) if t.dtype == torch.bfloat16 else t.contiguous().view(torch.int32)


def test_split_and_join_are_exact(dev):
    from mmego_amd import blocks
    g = torch.Generator().manual_seed(1)
    rows, K = 70, 64                                            # 70 rows: a ragged third row block (zero padded)
    x = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, K, generator=g) * 8.0)       # ~ 1e-10 .. 1e10
    special = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 3.3895e38, 3.4e38, -3.4e38, float("inf"), float("-inf"),
                            float("nan"), 1.1754944e-38, -1.1754944e-38, 1e-39, -1e-39, 1.4e-45, 2.0 ** -100, 2.0 ** -109,
                            2.0 ** -120, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.99998, 256.0, 0.1, 1.0 / 3.0, 65504.0,
                            1.00390625, 1.0039062, 1.0078125 - 2.0 ** -23, 3.1415927, -2.7182817], dtype=torch.float32)
    x.view(-1)[:special.numel()] = special
    y = blocks.split3_cvt(x.to(dev))
    got = [p.cpu() for p in _pieces_from_sfrag(y, rows, K)]
    want = _split_cpu(x)
    for gp, wp, name in zip(got, want, ("a1", "a2", "a3")):
        gb, wb = _bits(gp), _bits(wp)
        nan = torch.isnan(wp.float())
        assert torch.equal(gb[~nan], wb[~nan]), name
        assert torch.isnan(gp.float()[nan]).all(), name + ": NaN stays NaN"
    # padding rows of the last row block are zero pieces
    full = y.view(y.shape[0], y.shape[1], 3, 2, 32, 8)
    assert (full[2, :, :, :, rows - 64:, :].float() == 0).all()
    # split -> join: bit-exact wherever the three pieces can hold the value (|a| >= 2^-110, no overflow of bf16), and for +-0
    back = blocks.split3_join(y, rows, K).cpu()
    ax = x.abs()
    exact = torch.isfinite(x) & (ax >= 2.0 ** -110) & (ax <= 3.38e38)
    assert torch.equal(_bits(back)[exact], _bits(x)[exact])
    assert (back[x == 0] == 0).all()                            # (-0 comes back as +0: (-0) + (+0) = +0; the pieces of -0 are -0, +0, +0)
    assert torch.equal(back[torch.isinf(x)], x[torch.isinf(x)]) and torch.isnan(back[torch.isnan(x)]).all()
    tiny = torch.isfinite(x) & (ax < 2.0 ** -110) & (x != 0)
    assert tiny.any() and float((back[tiny].double() - x[tiny].double()).abs().max()) <= 2.0 ** -133
    # the CPU statement of the same claim on a large random sample: a1 + a2 + a3 == a
    z = torch.randn(1 << 20, generator=g) * torch.exp(torch.randn(1 << 20, generator=g) * 4.0)
    z1, z2, z3 = _split_cpu(z)
    assert torch.equal((z1.float() + z2.float()) + z3.float(), z)
    # time-major form: rows b*T + t in, rows t*Bp + b out
    Bn, T, Bp = 5, 3, 32
    xt = torch.randn(Bn * T, 32, generator=g)
    yt = blocks.split3_cvt(xt.to(dev), tm=(Bn, T, Bp))
    bt = blocks.split3_join(yt, T * Bp, 32).cpu().view(T, Bp, 32)
    assert torch.equal(bt[:, :Bn], xt.view(Bn, T, 32).permute(1, 0, 2)) and (bt[:, Bn:] == 0).all()


@pytest.mark.parametrize("M,N,K,wm", [(256, 256, 512, 2), (512, 384, 1024, 4), (160, 128, 64, 2), (288, 160, 96, 4), (200, 256, 1024, 1),
                                      (512, 4096, 1024, 0)])
def test_split3_gemm_against_float64(dev, M, N, K, wm):
    """C = A . W^T + bias: 6 piece products are fp32-accurate (error against float64 of the order of the native fp32 MFMA product's),
    9 are at least as good; both agree with the piece-product sums written out in float64; tile-major and row-major outputs agree."""
    from mmego_amd import blocks, hip, ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    scale = (A.double().abs() @ W.double().abs().t()).max().item()
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    Ap, Wp = blocks.split3_cvt(Ad), blocks.split3_cvt(Wd)
    Mrb, Nrb = Ap.shape[0], Wp.shape[0]
    native = torch.empty(M, N, device=dev)
    ops.linear(Ad, Wd, bd, native)
    e_native = (native.cpu().double() - ref).abs().max().item()
    a1, a2, a3 = [p.double() for p in _split_cpu(A)]
    w1, w2, w3 = [p.double() for p in _split_cpu(W)]
    six = a1 @ w1.t() + a1 @ w2.t() + a2 @ w1.t() + a2 @ w2.t() + a1 @ w3.t() + a3 @ w1.t() + bias.double()
    errs = {}
    for nprod in (6, 9):
        C = torch.zeros(M, N, device=dev)
        Cf = torch.zeros(Mrb * Nrb * 1024, device=dev)
        hip.call("split3_gemm", Ap, Wp, Cf, C, N, bd, Mrb, Nrb, K, M, nprod, wm)
        torch.cuda.synchronize()
        Cc = C.cpu().double()
        errs[nprod] = (Cc - ref).abs().max().item()
        # tile-major element (m, n): ((m/32) Nrb + n/32) 1024 + ((m%32)/8) 256 + (n%32 + 32 (((m%32)/4)&1)) 4 + m%4
        t = Cf.cpu().view(Mrb, Nrb, 4, 2, 32, 4)                  # [rb][cb][q][half][col][r]: row = 8 q + 4 half + r
        Ct = t.permute(0, 2, 3, 5, 1, 4).reshape(Mrb * 32, Nrb * 32)[:M, :N]
        assert torch.equal(Ct, C.cpu()), "tile-major and row-major outputs hold the same values"
        if nprod == 6:
            assert (Cc - six).abs().max().item() < 3e-7 * scale, ((Cc - six).abs().max().item(), scale)
    # measured: 0.9-2.4 x the native product's error (K = 1024: one fp32 accumulator takes 384 MFMA results per output where the
    # native kernel's takes 512 and splits them over chunks); 6 and 9 products agree to the last digit -- the error is the fp32
    # accumulation's, not the dropped 2^-24 terms'
    assert errs[6] < max(3.0 * e_native, 3e-7 * scale), (errs, e_native, scale)
    assert errs[9] < max(3.0 * e_native, 3e-7 * scale), (errs, e_native, scale)


# (r06: the first, fifth and sixth shape take 256 x 256 tiles x slabs on s3_gemm_big_kernel<., 8> -- stage-1 training's weight gradients;
#  the others the 256 x 128 / 64 x 128 kernel)
@pytest.mark.parametrize("R,Co,Ci,ns", [(10240, 4096, 512, 8), (10240, 2048, 512, 10), (512, 64, 96, 2), (1024, 100, 33, 3),
                                        (10240, 2048, 512, 16), (10240, 4096, 1024, 4)])
def test_split3_weight_gradient_product_in_k_slabs(dev, R, Co, Ci, ns):
    """dW = dY^T X on piece products (stage-1 training with IMUNet.train_precision = "split3"): mmego_split3_cvt_t's pieces of a
    transpose are bit-identical to mmego_split3_cvt of the explicit transpose; the K-slab product (mmego_split3_gemm_slabs) + the
    streaming slab sum against float64 -- no worse than the native fp32 product (at K = 10 240 it is 5 x closer) --, ragged widths."""
    from mmego_amd import blocks, hip
    g = torch.Generator().manual_seed(R + Co + Ci)
    dg, x = torch.randn(R, Co, generator=g).to(dev), torch.randn(R, Ci, generator=g).to(dev)
    Cop, Cip = (Co + 31) // 32 * 32, (Ci + 31) // 32 * 32
    a, b = blocks.split3_cvt_t(dg), blocks.split3_cvt_t(x)
    assert torch.equal(a.view(torch.int16), blocks.split3_cvt(dg.t().contiguous()).view(torch.int16))
    ws = torch.empty(ns * Cop * Cip, device=dev)
    hip.call("split3_gemm_slabs", a, b, ws, Cop // 32, Cip // 32, R, 6, 0, ns)
    out = torch.empty(Cop, Cip, device=dev)
    hip.call("split3_slab_sum", ws, ns, Cop * Cip, out)
    assert torch.equal(out, ws.view(ns, Cop, Cip).sum(0)) or float((out - ws.view(ns, Cop, Cip).sum(0)).abs().max()) < 1e-3     # (slab order vs torch's tree)
    ref = dg.double().t() @ x.double()
    e3 = float((out[:Co, :Ci].double() - ref).abs().max())
    en = float(((dg.t() @ x).double() - ref).abs().max())
    assert e3 <= max(1.5 * en, 3e-7 * float(ref.abs().max())), (e3, en)
    assert float(out[Co:].abs().max()) == 0.0 if Cop > Co else True


def test_split3_transposed_pieces_of_shifted_rows(dev):
    """mmego_split3_cvt_t with (shift, T): column r of the transpose is row r + shift of the same T-row sequence, zero outside it -- the
    h_{t-1} (shift -1) and h_{t+1} (shift +1) operands of the recurrent weight gradients read from a layer's outputs in place."""
    from mmego_amd import blocks
    g = torch.Generator().manual_seed(5)
    Bn, T, H = 24, 20, 96
    out = torch.randn(Bn * T, 2 * H, generator=g).to(dev)
    for d, shift in ((0, -1), (1, 1)):
        src = out[:, d * H:(d + 1) * H]
        ref = torch.zeros(Bn, T, H, device=dev)
        if shift < 0:
            ref[:, 1:] = src.view(Bn, T, H)[:, :-1]
        else:
            ref[:, :-1] = src.view(Bn, T, H)[:, 1:]
        got = blocks.split3_cvt_t(src, shift=shift, T=T)
        want = blocks.split3_cvt(ref.view(Bn * T, H).t().contiguous())
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (d, shift)


@pytest.mark.parametrize("M,H,K", [(512, 512, 1024), (200, 512, 512), (64, 256, 256)])
def test_split3_projection_for_few_rows(dev, M, H, K):
    """mmego_split3_proj (the step kernel's product alone: IMU_Net's rnn_slow input projections, 512 rows x 4096 columns x K = 1024) against
    float64 and the native fp32 product; both directions, ragged row counts."""
    from mmego_amd import blocks, hip, ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(2, 4 * H, K, generator=g) * 0.04
    bias = torch.randn(8 * H, generator=g)
    ref = torch.cat((A.double() @ W[0].double().t(), A.double() @ W[1].double().t()), 1) + bias.double()
    scale = (A.double().abs() @ W[0].double().abs().t()).max().item()
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    native = torch.empty(M, 8 * H, device=dev)
    ops.linear_pair(Ad, Wd[0], Wd[1], bd[:4 * H], bd[4 * H:], native, 4 * H)
    e_native = (native.cpu().double() - ref).abs().max().item()
    blocked = lambda m: m.view(4, H // 32, 32, -1).permute(1, 0, 2, 3).reshape(4 * H, -1).contiguous()
    Ap = blocks.split3_cvt(Ad)
    Wp = [blocks.split3_cvt(blocked(Wd[d])) for d in range(2)]
    for nprod in (6, 9):
        C = torch.full((M, 8 * H), float("nan"), device=dev)
        hip.call("split3_proj", Ap, Wp[0], Wp[1], bd, C, C.stride(0), M, H, K, nprod)
        torch.cuda.synchronize()
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < max(3.0 * e_native, 3e-7 * scale), (nprod, err, e_native, scale)


@pytest.mark.parametrize("Bn,T,H,chains", [(512, 20, 512, True), (512, 20, 512, False), (200, 5, 512, True), (100, 5, 512, False),
                                           (64, 3, 256, False), (160, 4, 256, True)])
def test_split3_bilstm_stack_against_the_fp32_step_kernels(dev, monkeypatch, Bn, T, H, chains):
    """blocks.lstm_steps_forward_split3 (projection GEMMs + recurrent steps on split operands) against blocks.lstm_steps_forward (the
    fp32 path: gemm_tile / lstm_step kernels) on the same two-layer BiLSTM: outputs of the last layer at fp32 rounding -- the bench
    shape (512 rows x 20 samples, every workgroup full), ragged row counts and H = 256; both forms of the recurrence: two chains of
    single-direction launches on 16-unit workgroups (mmego_split3_step16, from 128 rows) and one both-direction launch per timestep
    on 32-unit workgroups (mmego_split3_step)."""
    from mmego_amd import blocks, ops
    monkeypatch.setattr(blocks, "SPLIT3_TWO_CHAINS", chains)
    torch.manual_seed(17)
    lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
    x = torch.randn(Bn * T, H, device=dev).relu_()
    ar32, ar3 = ops.Arena(dev), ops.Arena(dev)
    with torch.no_grad():
        want = blocks.lstm_steps_forward(ar32, "t", lstm, x, Bn, T).clone()
        got6 = blocks.lstm_steps_forward_split3(ar3, "t", lstm, x, Bn, T, nprod=6).clone()
        got9 = blocks.lstm_steps_forward_split3(ar3, "t", lstm, x, Bn, T, nprod=9).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(got6).all()
    e6, e9 = (got6 - want).abs().max().item(), (got9 - want).abs().max().item()
    assert e6 < 4e-6 and e9 < 4e-6, (e6, e9)                       # |h| <= 1: fp32 rounding through T steps and two layers
    assert blocks.seq_xcd_errors() == 0


def test_split3_projection_on_320x256_tiles_is_the_256x128_kernel_bit_for_bit(dev):
    """s3_gemm_big_kernel (split3.hip, r06: 320 x 256 tiles, LDS-DMA operands; mmego_split3_gemm's choice for rnn_fast's projections)
    against s3_gemm_kernel<4, 2> (wm = 4) on the same pieces: tile-major and row-major outputs bit for bit, both K; a shape the
    big tiles do not divide takes the old kernel either way."""
    from mmego_amd import blocks, hip
    g = torch.Generator().manual_seed(12)
    for M, N, K in ((10240, 4096, 512), (10240, 4096, 1024), (640, 512, 256)):
        A = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        Ap, Wp = blocks.split3_cvt(A), blocks.split3_cvt(W)
        out = {}
        for wm in (4, 10, 0):
            Cf, C = torch.full((M * N,), float("nan"), device=dev), torch.full((M, N), float("nan"), device=dev)
            hip.call("split3_gemm", Ap, Wp, Cf, C, N, bias, M // 32, N // 32, K, M, 6, wm)
            out[wm] = (Cf, C)
        assert torch.isfinite(out[10][1]).all()
        for wm in (10, 0):
            assert torch.equal(out[wm][0], out[4][0]) and torch.equal(out[wm][1], out[4][1]), (M, N, K, wm)
        ref = A[:64].double() @ W.double().t() + bias.double()
        assert float((out[10][1][:64].double() - ref).abs().max()) < 3e-5 * (K / 512) ** 0.5


def test_imu_forward_split3_at_the_fp32_bars(dev):
    """tests/test_hip_parity.py::test_imu_forward with IMUNet.precision = "split3": the reference's golden G7 (seeded IMUNet(15, 9, 512,
    2), R and t within 2e-5) and the CPU oracle at 64, 128 and 200 rnn_fast rows; switching back restores the fp32 path bit for bit."""
    from mmego_amd import nets
    g = golden("g7_imu.npz")
    T_ = lambda a: torch.tensor(np.asarray(a))
    imu = T_(g["imu"])
    torch.manual_seed(703)
    big = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
    with torch.no_grad():
        R32, t32 = big(imu.to(dev))
    big.precision = "split3"
    with torch.no_grad():
        R, t = big(imu.to(dev))
    assert torch.allclose(R.cpu(), T_(g["big.R"]), atol=2e-5) and torch.allclose(t.cpu(), T_(g["big.t"]), atol=2e-5)
    assert float((R - R32).abs().max()) < 2e-5 and float((t - t32).abs().max()) < 2e-5
    ob = on.IMUNet(15, 9, 512, 2, True, 0.1).eval()
    ob.load_state_dict({k: v.cpu() for k, v in big.state_dict().items()})
    torch.manual_seed(9)
    worst = 0.0
    for Bq, Tq in ((8, 8), (25, 8), (16, 8), (64, 8)):
        imu3 = torch.randn(Bq, Tq, 20, 15)
        with torch.no_grad():
            Ro, to_ = ob(imu3)
            Rh, th = big(imu3.to(dev))
        worst = max(worst, float((Rh.cpu() - Ro).abs().max()), float((th.cpu() - to_).abs().max()))
        assert torch.allclose(Rh.cpu(), Ro, atol=2e-5) and torch.allclose(th.cpu(), to_, atol=2e-5), (Bq, Tq, worst)
    big.precision = "fp32"
    with torch.no_grad():
        R32b, t32b = big(imu.to(dev))
    assert torch.equal(R32, R32b) and torch.equal(t32, t32b)
    big.precision = "split2"
    with pytest.raises(ValueError):
        big(imu.to(dev))


def test_split3_and_bf16_weight_pieces_follow_load_state_dict_in_eval_mode(dev):
    """the cached weight pieces of the split3 mode were keyed by layout only and dropped by weights_changed() alone, so
    load_state_dict() / an in-place copy AFTER an eval-mode split3 forward kept the old pieces.  Now keyed on every parameter's
    (_version, data_ptr) like the bf16 copies: forward -> load other weights (still in eval mode) -> forward must equal a FRESH net that
    was given those weights from the start, bit for bit, in both modes."""
    from mmego_amd import nets
    imu = torch.randn(16, 8, 20, 15, generator=torch.Generator().manual_seed(2)).to(dev)
    torch.manual_seed(41)
    a = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
    torch.manual_seed(42)
    other = {k: v.clone() for k, v in nets.IMUNet(15, 9, 512, 2, True, 0.1).state_dict().items()}
    for mode in ("split3", "bf16"):
        a.precision = mode
        with torch.no_grad():
            R0, t0 = [v.clone() for v in a(imu)]
        a.load_state_dict({k: v.to(dev) for k, v in other.items()})
        with torch.no_grad():
            R1, t1 = [v.clone() for v in a(imu)]
        fresh = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
        fresh.load_state_dict({k: v.to(dev) for k, v in other.items()})
        fresh.precision = mode
        with torch.no_grad():
            R2, t2 = fresh(imu)
        assert not torch.equal(R0, R1), mode
        assert torch.equal(R1, R2) and torch.equal(t1, t2), (mode, float((R1 - R2).abs().max()))
        with torch.no_grad():                                  # an in-place change of one weight tensor
            a.rnn_fast.w("weight_hh", 0, 0).mul_(0.5)
            fresh.rnn_fast.w("weight_hh", 0, 0).mul_(0.5)
            fresh.weights_changed()
            R3, R4 = a(imu)[0].clone(), fresh(imu)[0]
        assert torch.equal(R3, R4), mode
        torch.manual_seed(41)
        a.load_state_dict({k: v.to(dev) for k, v in nets.IMUNet(15, 9, 512, 2, True, 0.1).state_dict().items()})


def test_ul_step_at_bench_shape_with_split3_imu(dev):
    """tests/test_bench_shape.py::test_ul_step_at_bench_shape_against_oracle with both frozen IMU_Net forwards in the split3 mode: the
    SAME bars (losses 2e-5 rel., joints 1e-3 cm, every gradient 2e-4 of the largest, post-Adam parameters)."""
    import re

    import bench
    torch.set_num_threads(bench.host_cores())
    r = bench.ul_step_parity(dev, use_graph=True, imu_precision="split3")
    noise = re.compile(bench.NOISE_GRAD)

    for tag in ("upper", "lower"):
        assert r["loss_rel_err_" + tag] < 2e-5, (tag, r["loss_rel_err_" + tag])
        assert r[tag + "_cm"] < 1e-3, (tag, r[tag + "_cm"])
        scale = r["tensors"][tag]["scale"]
        for k, (eg, dp) in r["tensors"][tag]["per_param"].items():
            assert eg < 2e-4 * scale, (tag, k, eg, scale)
            if not noise.search(k):
                assert dp <= 6e-5 + 2e-6, (tag, k, dp)
        assert r["param_frac_moved_" + tag] < 0.05, (tag, r["param_frac_moved_" + tag])


def _ul_engine(dev, kind, precision, unguarded=False, graph=True):
    """A U+L engine of train_step.py at the benchmarked shape with both frozen IMU_Net forwards (and, for "bf16", the frozen Upper_Net
    of the Lower stage) in the given precision mode.  -> (engine, its body, su, sl)"""
    import bench
    from mmego_amd.train_step import ConcurrentStages, PipelinedStages, SharedImuStages, StageStep
    x, imu_in, body, target = [v.to(dev) for v in bench.synth_batch(1234, "cpu")]
    himu, hup, hlo, hfr = bench.build_hip_models(dev)
    himu_l = bench.clone_imu(himu, dev)
    himu.precision = himu_l.precision = precision
    if precision == "bf16":
        hfr.precision = "bf16"
    bench._lstm_dropout_off(hup, hlo)
    own = kind == "concurrent"
    su = StageStep("upper", hup, himu if own else None, lr=3e-5, use_graph=False)
    sl = StageStep("lower", hlo, himu_l if own else None, upper_frozen=hfr, lr=3e-5, use_graph=False)
    kw = {"unguarded": True} if unguarded else {}
    if kind == "concurrent":
        eng = ConcurrentStages([su, sl], use_graph=graph, **kw)
        run = eng._bodies
    elif kind == "shared":
        eng = SharedImuStages(himu, [su, sl], imu_in, use_graph=graph)
        eng.pair.unguarded = unguarded
        run = eng._body
    else:
        eng = PipelinedStages([su, sl], [himu, himu_l], imu_in, use_graph=graph, **kw)
        run = eng._body
    su.bind(x, imu_in, body, target)
    sl.bind(x, imu_in, body, target)
    if kind == "pipelined":
        eng.prime()
    return eng, run, su, sl


def test_no_kernel_can_run_beside_a_bf16_mfma_kernel(dev):
    """STRUCTURAL guard of the r06 co-residency finding (VERDICT r05 item 1 c / d): with a net in "split3" or
    "bf16" precision every engine of train_step.py (the timed two-stage engine, the IMU-shared one, the prefetch-pipelined one that
    main.py --train runs) must keep every launch ordered against every bf16-MFMA launch (the forwards on one chain; only all-fp32 bodies fork, behind it) -- read off the recorded launch / wait graph of the engine's own
    body (plan.StepPlan.record: nothing is launched): no launch of any segment may be unordered against a launch of a bf16-MFMA entry
    point (hip.is_bf16_mfma_entry: everything exported by split3.hip, bf16.hip, *_bf16.hip).  Controls: the same engines in fp32
    precision DO have parallel segments (and no bf16-MFMA launch), and `unguarded=True` (bench.py's comparison figure) is seen by the
    check -- so an engine that forgot the guard would fail here every time, not in 5 % of runs."""
    from mmego_amd import hip
    from mmego_amd.plan import StepPlan
    for kind in ("concurrent", "shared", "pipelined"):
        for precision in ("split3", "bf16"):
            eng, run, su, sl = _ul_engine(dev, kind, precision, graph=False)
            eng.step()                                       # (sizes the arenas: recording must not allocate what a launch reads)
            torch.cuda.synchronize()
            plan = StepPlan().record(run)
            names = [n for sg in plan.segments for n, _ in sg.calls]
            hot = [n for n in names if hip.is_bf16_mfma_entry(n)]
            assert len(hot) >= 20, (kind, precision, len(hot))                       # the forwards really run on the bf16 pipe
            assert plan.unordered_with(hip.is_bf16_mfma_entry) == [], (kind, precision)
            if precision == "bf16":                          # (the Lower body holds the frozen Upper_Net's bf16 kernels: nothing forks at all)
                assert len({sg.stream for sg in plan.segments}) == 1, (kind, precision)
            else:                                            # split3: only the fp32 bodies fork, BEHIND every bf16-MFMA launch
                assert len({sg.stream for sg in plan.segments}) == 2, (kind, precision)
            del eng, run, su, sl
        # control 1: fp32 -- branches exist, no bf16-MFMA launch
        eng, run, su, sl = _ul_engine(dev, kind, "fp32", graph=False)
        eng.step()
        torch.cuda.synchronize()
        plan = StepPlan().record(run)
        assert len({sg.stream for sg in plan.segments}) > 1, kind
        assert not any(hip.is_bf16_mfma_entry(n) for sg in plan.segments for n, _ in sg.calls), kind
        assert plan.unordered_with(lambda n: n == "lstm_step"), kind                # (the check sees the fp32 branches)
        del eng, run, su, sl
        # control 2: the guard switched off -- the check must report the pairs
        eng, run, su, sl = _ul_engine(dev, kind, "split3", unguarded=True, graph=False)
        eng.step()
        torch.cuda.synchronize()
        pairs = StepPlan().record(run).unordered_with(hip.is_bf16_mfma_entry)
        if kind == "shared":
            assert pairs == [], kind           # (its one IMU_Net forward precedes the fork of the two bodies: nothing beside it either way)
        else:
            assert any(a in ("split3_step", "split3_gemm") and not hip.is_bf16_mfma_entry(b) for a, b in pairs), (kind, pairs[:5])
        del eng, run, su, sl


def test_stage1_training_on_split_products_is_one_chain_too(dev):
    """The same structural check for the stage-1 trainer with train_precision = "split3" (imu_train.py: piece products beside the
    fp32 recurrent steps of the same step)."""
    import bench
    from mmego_amd import hip
    from mmego_amd.plan import StepPlan
    from mmego_amd.train_step import ImuStep
    himu = bench.build_hip_models(dev)[0]
    himu.train()
    himu.train_precision = "split3"
    g = torch.Generator().manual_seed(5)
    B, T = 64, 8
    imu_in = torch.randn(B, T, 20, 15, generator=g).to(dev)
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(dev)
    target = torch.randn(B, T, 21, 3, generator=g).to(dev)
    st = ImuStep(himu, use_graph=False)
    st.bind(imu_in, R, target)
    st.step()
    torch.cuda.synchronize()
    plan = StepPlan().record(st._body)
    assert sum(hip.is_bf16_mfma_entry(n) for sg in plan.segments for n, _ in sg.calls) >= 8
    assert plan.unordered_with(hip.is_bf16_mfma_entry) == []


def test_unguarded_concurrent_split3_steps_are_reproducible_bit_for_bit(dev):
    """The r05 / r06 reproducer as a regression test of the BUILD (no packed-fp32 instruction in any kernel):
    the arrangement that corrupted the Lower stage's gradients -- both IMU_Net forwards in split3 mode as concurrent graph branches
    beside the other stage's tail (`unguarded=True`), in the 32-unit form AND in the 16-unit two-chain form -- 10 fresh engines x 3
    steps each, both stages' gradient buffers bit-equal to the first engine's.  With the packed instructions in the library 22 of 60
    (32-unit) and 59 of 60 (two chains) such engines differed (profiles/r06_coexec/r06_fullstep_b.log): this test would then fail
    with probability > 0.99; without them 0 of 120."""
    import bench
    from mmego_amd import blocks
    for two in (True, False):
        ref = None
        with blocks.split3_two_chains(two):
            for it in range(10):
                eng, run, su, sl = _ul_engine(dev, "concurrent", "split3", unguarded=True)
                gs = []
                for _ in range(3):
                    eng.step()
                    torch.cuda.synchronize()
                    gs.append([st.net.flat().flat_g.detach().clone() for st in (su, sl)])
                if ref is None:
                    ref = gs
                for s_, (ga, gb) in enumerate(zip(gs, ref)):
                    for a, b, tag in zip(ga, gb, ("upper", "lower")):
                        assert torch.equal(a, b), (two, it, s_, tag, float((a - b).abs().max()))
                del eng, run, su, sl
    assert blocks.seq_xcd_errors() == 0


def test_concurrent_step_with_split3_imu_is_reproducible_bit_for_bit(dev):
    """Six U+L steps of the timed engine (ConcurrentStages, HIP graph) from the same seeded state with both IMU_Net forwards in the
    split3 mode: both stages' gradient buffers bit-equal every time.  (r05: with its stage branches side by side this differed in
    ~10 % of runs, the co-residency finding; since r06 the engine runs such a step as one chain -- see the two tests above -- and the
    library holds no packed-fp32 instruction.)"""
    import bench
    from mmego_amd import blocks
    from mmego_amd.train_step import ConcurrentStages, StageStep
    assert blocks.SPLIT3_TWO_CHAINS is False
    x, imu_in, body, target = [v.to(dev) for v in bench.synth_batch(1234, "cpu")]
    ref = None
    for it in range(6):
        himu, hup, hlo, hfr = bench.build_hip_models(dev)
        himu_l = bench.clone_imu(himu, dev)
        himu.precision = himu_l.precision = "split3"
        bench._lstm_dropout_off(hup, hlo)
        su = StageStep("upper", hup, himu, lr=3e-5, use_graph=True)
        sl = StageStep("lower", hlo, himu_l, upper_frozen=hfr, lr=3e-5, use_graph=True)
        su.bind(x, imu_in, body, target)
        sl.bind(x, imu_in, body, target)
        ConcurrentStages([su, sl], use_graph=True).step()
        torch.cuda.synchronize()
        g = [st.net.flat().flat_g.detach().clone() for st in (su, sl)]
        if ref is None:
            ref = g
        for a, b, tag in zip(g, ref, ("upper", "lower")):
            assert torch.equal(a, b), (it, tag, float((a - b).abs().max()))
        junk = torch.full((16 << 20,), float("nan"), device=dev)           # (the next nets land on used memory, as in a long session)
        del junk
    assert blocks.seq_xcd_errors() == 0
