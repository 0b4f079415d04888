"""bf16-operand / fp32-accumulate mode of the frozen IMU_Net forward (mmego_amd/csrc/bf16.hip; BASELINE config 5).

The checker is an emulation of the same arithmetic on the CPU: operands rounded to bf16 with torch's
round-to-nearest-even cast, products and sums in float64.  Tolerances (stated per test): the conversion is bit-exact;
a product differs from the emulation only by fp32 summation order; the recurrence re-rounds h_t to bf16 every step,
so a 1e-7 difference can flip one bf16 ulp (2^-8 relative) of an operand of the next step.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def test_cvt_bf16_is_round_to_nearest_even_bit_exact():
    from mmego_amd import blocks
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 64, generator=g) * torch.logspace(-30, 30, 64)
    x[0, :8] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1.00390625, 1.01171875, 3.3895e38])
    x[1, :4] = torch.tensor([1e-40, -1e-40, 1.0 + 2 ** -9, 1.0 + 2 ** -8 + 2 ** -9])      # denormals, ties
    xd = x.to(_dev())
    out = torch.empty((37, 64), dtype=torch.bfloat16, device=_dev())
    blocks.cvt_bf16(xd, out)
    want = x.to(torch.bfloat16)
    got = out.cpu()
    nan = torch.isnan(want.float())
    assert torch.equal(torch.isnan(got.float()), nan)
    assert torch.equal(got.view(torch.int16)[~nan], want.view(torch.int16)[~nan])
    # row-strided views on both sides
    big = torch.zeros((37, 96), dtype=torch.bfloat16, device=_dev())
    blocks.cvt_bf16(xd[:, 16:48], big[:, 32:64])
    assert torch.equal(big[:, 32:64].cpu().view(torch.int16)[~nan[:, 16:48]], want[:, 16:48].view(torch.int16)[~nan[:, 16:48]])
    assert float(big[:, :32].float().abs().sum()) == 0.0 and float(big[:, 64:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("M,N,K,relu", [(200, 192, 128, 0), (512, 4096, 512, 0), (1, 1, 64, 1), (300, 130, 1024, 1),
                                         (1120, 1024, 128, 0), (96, 160, 320, 1)])
def test_gemm_bf16_matches_fp64_product_of_rounded_operands(M, N, K, relu):
    from mmego_amd import blocks, hip
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dev = _dev()
    Ab = blocks.cvt_bf16(A.to(dev), torch.empty((M, K), dtype=torch.bfloat16, device=dev))
    Wb = blocks.cvt_bf16(W.to(dev), torch.empty((N, K), dtype=torch.bfloat16, device=dev))
    C = torch.full((M, N + 3), 7.0, device=dev)            # padded leading dimension: the pad must stay untouched
    Cb = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    tile_major = M % 32 == 0 and N % 32 == 0
    Cf = torch.zeros(M * N, device=dev) if tile_major else None
    hip.call("gemm_bf16", Ab, K, Wb, K, C, C.stride(0), Cb, Cb.stride(0), Cf, b.to(dev), M, N, K, relu)
    want = _bf(A) @ _bf(W).T + b.double()
    if relu:
        want = want.clamp_min(0)
    got = C[:, :N].cpu().double()
    # fp32 accumulation of K exact products of O(1/sqrt(K)) magnitude: error ~ K * 2^-24 * |partial sums| << 1e-4
    assert float((got - want).abs().max()) < 1e-4
    assert float(C[:, N:].min()) == 7.0
    assert torch.equal(Cb.cpu().view(torch.int16), C[:, :N].cpu().to(torch.bfloat16).view(torch.int16))
    if tile_major:
        # element (m, n) at ((m/32)*(N/32) + n/32)*1024 + ((m%32)/8)*256 + (n%32 + 32*(((m%32)/4)&1))*4 + m%4
        m = torch.arange(M).view(M, 1)
        n = torch.arange(N).view(1, N)
        off = ((m // 32) * (N // 32) + n // 32) * 1024 + ((m % 32) // 8) * 256 + (n % 32 + 32 * (((m % 32) // 4) & 1)) * 4 + m % 4
        assert torch.equal(Cf.cpu()[off], C[:, :N].cpu())


def _emulate_bilstm(x, W, H, Bn, T):
    """x [Bn*T, In] (rows b*T+t) float64 already bf16-rounded; W = per layer per direction (w_ih, w_hh, b_ih, b_hh)."""
    cur = x
    for layer in W:
        out = torch.zeros(Bn, T, 2 * H, dtype=torch.float64)
        xin = cur.view(Bn, T, -1)
        for d, (w_ih, w_hh, b_ih, b_hh) in enumerate(layer):
            xp = (xin @ _bf(w_ih).T).float() + (b_ih + b_hh)           # fp32 projection output, as on the device
            h = torch.zeros(Bn, H, dtype=torch.float64)
            c = torch.zeros(Bn, H, dtype=torch.float64)
            for s in range(T):
                t = s if d == 0 else T - 1 - s
                gates = xp[:, t].double() + _bf(h.float()) @ _bf(w_hh).T
                i, f, gg, o = gates.split(H, dim=1)
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                out[:, t, d * H:(d + 1) * H] = h
        cur = _bf(out.view(Bn * T, 2 * H).float())
    return out.view(Bn * T, 2 * H)


@pytest.mark.parametrize("Bn,T,H,In,fused", [(200, 5, 128, 64, False), (2100, 3, 64, 128, False), (64, 4, 512, 1024, False),
                                               (200, 3, 256, 64, False), (2050, 2, 256, 64, False),
                                               (2100, 3, 64, 128, True), (2050, 2, 256, 64, True), (130, 4, 128, 192, True),
                                               # multiples of 256 rows with H % 64 == 0: the 256 x 256-tile LDS-DMA form of the fused step
                                               (512, 3, 64, 128, True), (256, 4, 128, 192, True), (768, 2, 256, 64, True)])
def test_bilstm_bf16_forward_matches_emulation(Bn, T, H, In, fused, monkeypatch):
    """fused = the large-batch form (projection folded into the step kernel; default from 2048 rows), forced on or off here."""
    from mmego_amd import blocks, ops
    monkeypatch.setattr(blocks, "FUSED_MIN_ROWS", 1 if fused else 10 ** 9)
    torch.manual_seed(Bn)
    lstm = blocks.LstmParams(In, H, 2).to(_dev())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bn * T, In, generator=g)
    out = blocks.lstm_steps_forward_bf16(ops.Arena(_dev()), "t", lstm, x.to(_dev()), Bn, T)
    W = [[tuple(lstm.w(k, l, d).detach().cpu() for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for d in range(2)]
         for l in range(2)]
    want = _emulate_bilstm(_bf(x), W, H, Bn, T)
    got = out.cpu().double()
    err = (got - want).abs()
    # |h| < 1; a flipped bf16 ulp of one operand moves a gate by <= 2^-8 * |w| ~ 4e-4: allow 2e-3 at worst, 2e-5 typical
    assert float(err.max()) < 2e-3, float(err.max())
    assert float(err.mean()) < 2e-5, float(err.mean())


def test_imu_forward_bf16_mode_is_close_to_fp32_and_is_opt_in():
    from mmego_amd import nets
    torch.manual_seed(3)
    net = nets.IMUNet(15, 9, 512, 2).to(_dev()).eval()
    assert net.precision == "fp32"
    g = torch.Generator().manual_seed(2)
    imu = torch.randn(16, 8, 20, 15, generator=g).to(_dev())
    with torch.no_grad():
        R32, t32 = [v.clone() for v in net(imu)]
        net.precision = "bf16"
        Rb, tb = [v.clone() for v in net(imu)]
        Rb2, tb2 = net(imu)
        assert torch.equal(Rb, Rb2) and torch.equal(tb, tb2)          # deterministic
        net.precision = "fp32"
        R32b, _ = net(imu)
    assert torch.equal(R32, R32b)                                     # switching back restores the fp32 path bit for bit
    assert not torch.equal(R32, Rb)
    # bf16 operands (2^-9 relative rounding) through two BiLSTM stacks: head rotation entries move by ~1e-3
    assert float((R32 - Rb).abs().max()) < 2e-2
    assert float((t32 - tb).abs().max()) < 2e-2
    net.precision = "fp16"
    with pytest.raises(ValueError):
        net(imu)


@pytest.mark.parametrize("Bn", [4096, 4352, 4128])
def test_large_batch_fused_and_unfused_forms_agree(monkeypatch, Bn):
    """At a config-5-like size (4096 sequences, H = 512) the CPU emulation is too slow; instead the two device forms of the mode
    -- separate tile-major projection + step kernels, and the projection folded into the step -- must agree: same bf16 operand
    roundings, fp32 accumulation in a different order (plus the occasional flipped bf16 ulp of h_t that follows from it).
    4096 rows: the persistent 256 x 256-tile kernel on its XCD-contiguous walk (256 tiles); 4352 rows: 272 tiles, the plain walk
    with a ragged last round; 4128 rows (not a multiple of 256): the 128-row kernel with a partial last row block."""
    from mmego_amd import blocks, ops
    T, H, In = 3, 512, 512
    torch.manual_seed(11)
    lstm = blocks.LstmParams(In, H, 2).to(_dev())
    x = torch.randn(Bn * T, In, generator=torch.Generator().manual_seed(12)).to(_dev())
    outs = {}
    for fused in (False, True):
        monkeypatch.setattr(blocks, "FUSED_MIN_ROWS", 1 if fused else 10 ** 9)
        outs[fused] = blocks.lstm_steps_forward_bf16(ops.Arena(_dev()), "t", lstm, x, Bn, T).clone()
    err = (outs[True] - outs[False]).abs()
    assert float(err.max()) < 2e-3 and float(err.mean()) < 2e-5, (float(err.max()), float(err.mean()))
    assert float(outs[True].abs().mean()) > 1e-2          # not trivially zero


def test_fc_relu_bf16_fragments_and_the_imu_forward_that_uses_them(monkeypatch):
    """mmego_fc_relu_bf16_frag_tm (IMU_Net's fc1 written straight as the fused step's layer-0 operand): every element is the bf16
    rounding of relu(x . W^T + b) (fp32 sums in another order than the GEMM's: a value within half a bf16 ulp of a tie may round
    the other way), at its fragment-major position, rows past Bn zero; and the IMU_Net forward that uses it agrees with the one
    that stores the fp32 activation and converts it (nets._BF16_FUSED_FC1 = False) to the mode's bound."""
    from mmego_amd import blocks, hip, nets
    dev = _dev()
    Bn, T, Cin, H = 70, 3, 15, 128
    Bp = 96
    g = torch.Generator().manual_seed(5)
    X = torch.randn(Bn * T, Cin, generator=g)
    W = torch.randn(H, Cin, generator=g) * 0.3
    b = torch.randn(H, generator=g) * 0.1
    Y = torch.full((T, Bp * H), 7.0, dtype=torch.bfloat16, device=dev)
    hip.call("fc_relu_bf16_frag_tm", X.to(dev), Cin, W.to(dev), b.to(dev), Bn, T, Cin, H, Y, Bp, 1)
    got = Y.cpu().float().view(T, Bp * H)
    want = torch.relu(X.double() @ W.double().T + b.double()).view(Bn, T, H)
    r = torch.arange(Bp).view(Bp, 1)
    k = torch.arange(H).view(1, H)
    off = (((r // 32) * (H // 16) + k // 16) * 64 + ((k // 8) % 2) * 32 + r % 32) * 8 + k % 8        # frag_off (bf16.hip)
    for t in range(T):
        m = got[t][off]                                         # [Bp, H] in matrix order
        assert float(m[Bn:].abs().max()) == 0.0
        ref = want[:, t]
        err = (m[:Bn].double() - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all()), float(err.max())
        exact = m[:Bn] == ref.float().to(torch.bfloat16).float()
        assert float(exact.float().mean()) > 0.995
    # the forward that uses it
    torch.manual_seed(9)
    net = nets.IMUNet(15, 9, 128, 2).to(dev).eval()
    net.precision = "bf16"
    monkeypatch.setattr(blocks, "FUSED_MIN_ROWS", 1)
    imu = torch.randn(32, 8, 20, 15, generator=g).to(dev)         # 256 rows: the 256 x 256-tile fused step as well
    with torch.no_grad():
        R1, t1 = [v.clone() for v in net(imu)]
        monkeypatch.setattr(nets, "_BF16_FUSED_FC1", False)
        R0, t0 = [v.clone() for v in net(imu)]
    assert float((R1 - R0).abs().max()) < 2e-2 and float((t1 - t0).abs().max()) < 2e-2
    assert float((R1 - R0).abs().mean()) < 2e-3


@pytest.mark.parametrize("T,Bn,H", [(20, 70, 128), (5, 32, 512), (20, 64, 256), (1, 33, 128)])
def test_attention_pooling_from_bf16_fragments(T, Bn, H):
    """mmego_attn_pool_frag_bf16 (IMU_Net's pooling over the 20 samples read from the fused step's own bf16 h_t fragments, online
    softmax) against the fp32 pooling kernel on the same bf16-rounded values: pooled vectors and attention weights at fp32 rounding;
    ragged row counts (rows past Bn are padding), one timestep."""
    from mmego_amd import blocks, hip
    dev = _dev()
    g = torch.Generator().manual_seed(T * 1000 + Bn + H)
    Bp = (Bn + 31) // 32 * 32
    h = (torch.randn(T, 2, Bp, H, generator=g) * 1.5).to(torch.bfloat16)
    r, k = torch.arange(Bp)[:, None], torch.arange(H)[None, :]
    off = ((((r >> 5) * (H >> 4) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (r & 31)) * 8 + (k & 7)      # frag_off of bf16.hip
    hf = torch.empty(T, 2, Bp * H, dtype=torch.bfloat16)
    for t in range(T):
        for d in range(2):
            hf[t, d][off.flatten()] = h[t, d].flatten()
    lin = torch.nn.Linear(2 * H, 1)
    lin.weight.data.normal_(0.0, 0.2, generator=g)
    lin = lin.to(dev)
    # the fp32 kernel's input: rows (b * T + t), columns [direction 0 | direction 1]
    fast = torch.cat([h[:, 0], h[:, 1]], dim=2)[:, :Bn].float().permute(1, 0, 2).reshape(Bn * T, 2 * H).contiguous().to(dev)
    vec_ref, attn_ref = torch.empty(Bn, 2 * H, device=dev), torch.empty(Bn, T, device=dev)
    blocks.attn_pool_forward(fast, lin, Bn, T, 2 * H, vec_ref, attn_ref)
    vec, attn = torch.full((Bn, 2 * H), 7.0, device=dev), torch.full((Bn, T), 7.0, device=dev)
    assert hip.lib().mmego_attn_pool_frag_bf16_ok(H)
    hip.call("attn_pool_frag_bf16", hf.to(dev), T, Bp, Bn, H, lin.weight, lin.bias, vec, attn)
    torch.cuda.synchronize()
    assert float((attn - attn_ref).abs().max()) < 2e-6
    assert float((vec - vec_ref).abs().max()) < 2e-5 * max(1.0, float(vec_ref.abs().max()))
    assert abs(float(attn.sum(1).mean()) - 1.0) < 1e-5


def test_config5_full_size_bf16_forward():
    """BASELINE config 5 at its FULL size (B = 2048 sequences, T = 16 frames -> 32 768 rows x 20 samples through rnn_fast, the
    fused projection + recurrence step; 2048 x 16 through rnn_slow): IMUNet.precision = "bf16".  The CPU emulation cannot run
    this size, so the check is through size-independent properties: (i) every output is finite and R is a rotation;
    (ii) sequences are independent in eval mode, so the first 4 sequences of the big batch equal the same 4 run alone -- a
    different kernel family (64 rows: the direct step kernel + tile-major projections), i.e. the same bf16 operand roundings
    accumulated in another order, hence the mode's stated 2e-3 bound rather than equality; (iii) against the fp32 path on the
    same inputs the head rotation moves by < 2e-2 (bf16 operands, 2^-9 relative rounding, two BiLSTM stacks)."""
    from mmego_amd import nets
    dev = _dev()
    torch.manual_seed(17)
    net = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
    B, Tn = 2048, 16
    g = torch.Generator().manual_seed(18)
    imu = torch.zeros(B, Tn, 20, 15)
    imu[..., :9] = torch.linalg.qr(torch.randn(B, Tn, 20, 3, 3, generator=g))[0].reshape(B, Tn, 20, 9)
    imu[..., 9:12] = torch.randn(B, Tn, 20, 3, generator=g)
    imu[..., 12:] = 0.3 * torch.randn(B, Tn, 20, 3, generator=g)
    imu = imu.to(dev)
    with torch.no_grad():
        net.precision = "bf16"
        Rb, tb = [v.clone() for v in net(imu)]
        Rs, ts = [v.clone() for v in net(imu[:4].contiguous())]
        net.precision = "fp32"
        R32, t32 = [v.clone() for v in net(imu)]
    assert Rb.shape == (B, Tn, 3, 3) and tb.shape == (B, Tn, 3)
    assert bool(torch.isfinite(Rb).all()) and bool(torch.isfinite(tb).all())
    eye = torch.eye(3, device=dev).expand(B, Tn, 3, 3)
    assert float((Rb @ Rb.transpose(-1, -2) - eye).abs().max()) < 1e-4, "R is orthonormal (6-D -> rotation head)"
    assert float((Rb[:4] - Rs).abs().max()) < 2e-3 and float((tb[:4] - ts).abs().max()) < 2e-3, \
        (float((Rb[:4] - Rs).abs().max()), float((tb[:4] - ts).abs().max()))
    assert float((Rb - R32).abs().max()) < 2e-2 and float((tb - t32).abs().max()) < 2e-2, \
        (float((Rb - R32).abs().max()), float((tb - t32).abs().max()))
    assert float((Rb - R32).abs().mean()) < 2e-3


# ---- eval-mode ST-GCN block on bf16 operands (mmego_amd/csrc/gcn_bf16.hip) --------------------------------------------------------
@pytest.mark.parametrize("B,T,V,C", [(3, 16, 15, 128), (5, 8, 15, 64), (2, 16, 15, 32), (2, 20, 15, 64), (1, 3, 15, 128), (2, 40, 13, 32)])
def test_tconv_eval_bf16_matches_emulation(B, T, V, C):
    """mmego_tconv_eval_bf16 against the same arithmetic on the CPU: the (already bf16) input and the bf16-rounded weights
    multiplied and summed in float64, then bias -> BatchNorm affine -> + residual -> ReLU.  The kernel accumulates 9 C products per
    output in fp32 in another order: 2e-5 of the output scale at worst.  (2, 20, 15, 64) and (2, 40, 13, 32): T V > 256 rows, two
    and three row passes per sequence; (1, 3, 15, 128): a sequence shorter than the kernel's reach of 4 frames each way."""
    from mmego_amd import hip
    dev, taps = _dev(), 9
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + C)
    rows = B * T * V
    x = torch.relu(torch.randn(rows, C, generator=g)).to(torch.bfloat16)
    w = torch.randn(C, C, taps, 1, generator=g) / (taps * C) ** 0.5
    bias, res = torch.randn(C, generator=g) * 0.1, torch.randn(rows, C + 4, generator=g)
    post = torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5,
                        torch.randn(C, generator=g) * 0.1])
    assert hip.lib().mmego_tconv_eval_bf16_ok(T, V, C, C, taps)
    wp = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    hip.call("tconv_pack_bf16", w.to(dev), C, C, taps, wp)
    resd = res.to(dev)
    outs = {}
    for name, (pb, pp, pr, relu) in {"full": (bias, post, resd, 1), "plain": (None, None, None, 0)}.items():
        y = torch.full((rows, C + 8), 7.0, device=dev)
        hip.call("tconv_eval_bf16", x.to(dev), wp, None if pb is None else pb.to(dev), None if pp is None else pp.to(dev).contiguous(),
                 None if pr is None else pr[:, 2:], resd.stride(0), y[:, 4:4 + C], y.stride(0), relu, B, T, V, C, C, taps)
        torch.cuda.synchronize()
        assert torch.all(y[:, :4] == 7.0) and torch.all(y[:, 4 + C:] == 7.0)           # nothing outside the output columns
        outs[name] = y[:, 4:4 + C].cpu().double()
    xe = x.double().view(B, T, V, C).permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d(xe, _bf(w), None, padding=(taps // 2, 0)).permute(0, 2, 3, 1).reshape(rows, C)
    scale = float(conv.abs().max())
    err = float((outs["plain"] - conv).abs().max())
    assert err < 2e-5 * scale, (err, scale)
    full = torch.relu((conv + bias.double() - post[0].double()) * post[2].double() + post[3].double() + res[:, 2:2 + C].double())
    err = float((outs["full"] - full).abs().max())
    assert err < 2e-5 * max(scale, float(full.abs().max())), err


def _emulate_mix(x, in_state, AI, Wg, bg, Wres, bres, st0, st_r, F, V, cin, cout, K):
    """gcn_mix_eval_bf16 in float64 with the kernel's roundings: X (after data_bn's affine, fp32) is mixed with A . importance, the
    mixed copies and X itself are rounded to bf16, the weights are rounded to bf16, everything else is exact."""
    xf = x.view(F, V, cin).float()
    if in_state is not None:
        m, a, b = (in_state[i].view(V, cin) for i in (0, 2, 3))
        xf = (xf - m) * a + b
    xd = xf.double()
    y = torch.zeros(F, V, cout, dtype=torch.float64)
    for k in range(K):
        xk = torch.einsum("vw,fvc->fwc", AI[k].double(), xd).float()                 # (fp32 like the kernel, up to summation order)
        y += _bf(xk) @ _bf(Wg[k]).t()
    y += torch.einsum("kw,kc->wc", AI.double().sum(dim=1), bg.double().view(K, cout))
    yact = torch.relu((y - st0[0].double()) * st0[2].double() + st0[3].double())
    r = _bf(xf) @ _bf(Wres).t() + bres.double()
    rn = (r - st_r[0].double()) * st_r[2].double() + st_r[3].double()
    return yact.view(F * V, cout), rn.view(F * V, cout)


@pytest.mark.parametrize("cin,cout,F,K", [(3, 32, 19, 3), (32, 64, 64, 3), (64, 128, 37, 3), (64, 128, 8, 2), (3, 32, 2100, 2), (32, 64, 21, 2),
                                          (64, 128, 9, 1)])
def test_gcn_mix_eval_bf16_matches_emulation(cin, cout, F, K):
    """mmego_gcn_mix_eval_bf16 against _emulate_mix.  Differences: fp32 summation order, and -- rarely -- one bf16 ulp (2^-8
    relative) of a mixed operand whose fp32 sum lands on the other side of a rounding boundary, which moves an output by
    ~4e-3 x |w| x |x|: 3e-3 of the output scale at worst, 3e-5 on average; the bf16 einsum output within one bf16 ulp.
    F = 19 / 37: ragged last tile of 8 frames; 2100 frames: more tiles than workgroups (the persistent loop); K = 2 is the skeleton graph's
    partition count, K = 3 the reference's general case."""
    from mmego_amd import hip
    dev, V = _dev(), 15
    g = torch.Generator().manual_seed(cin * 100 + F)
    x = torch.randn(F * V, cin, generator=g)
    A = torch.rand(K, V, V, generator=g) * (torch.rand(K, V, V, generator=g) < 0.3).float()
    imp = torch.rand(K, V, V, generator=g) + 0.5
    Wg = torch.randn(K, cout, cin, generator=g) / (K * cin) ** 0.5
    bg = torch.randn(K * cout, generator=g) * 0.1
    Wres, bres = torch.randn(cout, cin, generator=g) / cin ** 0.5, torch.randn(cout, generator=g) * 0.1

    def state(C):
        return torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5,
                            torch.randn(C, generator=g) * 0.1]).contiguous()
    st0, st_r = state(cout), state(cout)
    in_state = state(V * cin) if cin == 3 else None
    assert hip.lib().mmego_gcn_mix_eval_bf16_ok(V, cin, cout, K)
    kd = ((K + 1) * cin + 15) // 16 * 16
    ksy, kr0 = (K * cin + 15) // 16, (K * cin) // 16
    Wcat = torch.zeros(cout, kd)
    Wcat[:, :K * cin] = Wg.permute(1, 0, 2).reshape(cout, K * cin)
    Wy = Wcat[:, :ksy * 16].clone()
    Wcat.zero_()
    Wcat[:, K * cin:(K + 1) * cin] = Wres
    Wr = Wcat[:, kr0 * 16:].clone()

    def frag(Wm):
        co, n = Wm.shape
        return Wm.to(torch.bfloat16).view(co // 32, 32, n // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().to(dev)
    AI = A * imp
    biasy = torch.einsum("kw,kc->wc", AI.sum(dim=1), bg.view(K, cout)).contiguous()
    yact = torch.zeros(F * V + 3, cout, dtype=torch.bfloat16, device=dev)
    rn = torch.full((F * V + 3, cout), 7.0, device=dev)
    hip.call("gcn_mix_eval_bf16", x.to(dev), None if in_state is None else in_state.to(dev), A.to(dev), imp.to(dev), frag(Wy), frag(Wr),
             biasy.to(dev), bres.to(dev), st0.to(dev), st_r.to(dev), yact, rn, F, V, cin, cout, K)
    torch.cuda.synchronize()
    assert torch.all(rn[F * V:] == 7.0) and torch.all(yact[F * V:].float() == 0.0)      # nothing behind the last frame
    want_y, want_r = _emulate_mix(x, in_state, AI, Wg, bg, Wres, bres, st0, st_r, F, V, cin, cout, K)
    got_y, got_r = yact[:F * V].cpu().double(), rn[:F * V].cpu().double()
    sr = float(want_r.abs().max())
    er = (got_r - want_r).abs()
    if in_state is None:
        assert float(er.max()) < 2e-5 * sr, (float(er.max()), sr)      # the residual product's operands are exact roundings of x
    else:       # x passes data_bn's affine in fp32 first (one fused multiply-add in the kernel): rarely the other side of a bf16 boundary
        assert float(er.max()) < 3e-3 * sr and float(er.mean()) < 2e-5 * sr, (float(er.max()), float(er.mean()), sr)
    sy = float(want_y.abs().max())
    ey = (got_y - _bf(want_y.float())).abs()
    assert float(ey.max()) < 3e-3 * sy + 2.0 ** -7 * sy, (float(ey.max()), sy)
    assert float(ey.mean()) < 2e-4 * sy, (float(ey.mean()), sy)
    # and within bf16 resolution of the emulation for nearly every element
    close = (got_y - want_y).abs() <= 2.0 ** -8 * want_y.abs() + 1e-4 * sy
    assert float(close.double().mean()) > 0.999


def test_lower_eval_bf16_mode_is_close_to_fp32_and_is_opt_in():
    """LowerNet.precision = "bf16": the eval forward's ST-GCN runs on gcn_bf16.hip; joints move by well under a millimetre per
    metre of skeleton, the mode is deterministic, switching back restores the fp32 path bit for bit, training ignores it."""
    from mmego_amd import nets
    dev = _dev()
    torch.manual_seed(21)
    lo = nets.LowerNet(64).to(dev).eval()
    assert lo.precision == "fp32"
    with torch.no_grad():
        for m in lo.modules():                                       # running statistics away from (0, 1): the folded affines matter
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
    g = torch.Generator().manual_seed(22)
    B, T, N = 6, 16, 128
    x = torch.randn(B, T, N, 6, generator=g).to(dev)
    up = (torch.randn(B, T, 15, 3, generator=g) * 0.3).to(dev)
    body = (torch.randn(B, 20, 3, generator=g) * 0.2).to(dev)
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(dev)
    t = (torch.randn(B, T, 3, generator=g) * 0.1).to(dev)

    def fwd():
        with torch.no_grad():
            l, q = lo(up.clone(), x.clone(), None, None, None, None, body, R, t)[:2]
        return l.clone(), q.clone()
    l32, q32 = fwd()
    lo.precision = "bf16"
    lb, qb = fwd()
    lb2, qb2 = fwd()
    assert torch.equal(lb, lb2) and torch.equal(qb, qb2)
    assert not torch.equal(l32, lb)
    assert float((l32 - lb).abs().max()) < 2e-2, float((l32 - lb).abs().max())
    assert float((q32 - qb).abs().max()) < 5e-2
    lo.precision = "fp32"
    l32b, q32b = fwd()
    assert torch.equal(l32, l32b) and torch.equal(q32, q32b)
    lo.precision = "fp16"
    with pytest.raises(ValueError):
        fwd()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_eval_packs_follow_the_weights_through_a_training_step(precision):
    """LowerNet's eval-mode packs (the bf16 ST-GCN pack `_gcn_bf16_pack`: weight fragments, folded BatchNorm states, bias
    tables; the fp32 path's re-packed temporal-conv weights `_tconv_packed`) were keyed on tensor._version, which the fused Adam and
    the train-mode BatchNorm kernels never bump (they write through raw pointers): the first evaluation froze them.  Eval -> one
    training step (forward, backward, Adam at a large lr) -> eval must CHANGE the output, and the second eval must equal a freshly
    constructed net that holds the same state (so nothing stale is left)."""
    from mmego_amd import nets
    from mmego_amd.params import FusedAdam
    dev = _dev()
    torch.manual_seed(41)
    lo = nets.LowerNet(64).to(dev)
    lo.precision = precision
    g = torch.Generator().manual_seed(42)
    B, T, N = 4, 8, 128
    x = torch.randn(B, T, N, 6, generator=g).to(dev)
    up = (torch.randn(B, T, 15, 3, generator=g) * 0.3).to(dev)
    body = (torch.randn(B, 20, 3, generator=g) * 0.2).to(dev)
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(dev)
    t = (torch.randn(B, T, 3, generator=g) * 0.1).to(dev)
    target = torch.randn(B, T, 8, 3, generator=g).to(dev)

    def ev(net):
        net.eval()
        with torch.no_grad():
            l, q = net(up.clone(), x.clone(), None, None, None, None, body, R, t)[:2]
        return l.clone(), q.clone()
    l0, q0 = ev(lo)
    lo.train()
    lo.lstm_dropout = 0.0
    l = lo(up.clone(), x.clone(), None, None, None, None, body, R, t)[0]
    (l - target).abs().sum().backward()
    FusedAdam(lo.flat(), lr=1e-2).step()
    torch.cuda.synchronize()
    l1, q1 = ev(lo)
    assert not torch.equal(l0, l1), "the evaluation after a training step still ran the old weights"
    fresh = nets.LowerNet(64).to(dev)
    fresh.precision = precision
    fresh.load_state_dict(lo.state_dict())
    l2, q2 = ev(fresh)
    assert torch.equal(l1, l2) and torch.equal(q1, q2), float((l1 - l2).abs().max())
    # ... and a step taken while the net is ALREADY in eval mode (no .train() call in between) is seen too
    lo.flat().flat_g.fill_(1e-3)
    FusedAdam(lo.flat(), lr=1e-2).step()
    torch.cuda.synchronize()
    l3, _ = ev(lo)
    fresh.load_state_dict(lo.state_dict())
    l4, _ = ev(fresh)
    assert not torch.equal(l1, l3) and torch.equal(l3, l4)


def test_upper_front_eval_bf16_against_emulation_and_fp32():
    """mmego_upper_front_eval_bf16 (front_bf16.hip) against the six stages written out in float64 with the kernel's roundings (folded
    weights and every stage's input rounded to bf16), and UpperNet.precision = "bf16" end to end: the transformed points written back
    into x are bit-identical to the fp32 launch, pooled features within bf16 resolution of the emulation, opt-in and reversible."""
    from mmego_amd import nets
    dev = _dev()
    torch.manual_seed(31)
    up = nets.UpperNet().to(dev).eval()
    with torch.no_grad():
        for m in up.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
                m.weight.uniform_(0.7, 1.3)
                m.bias.uniform_(-0.1, 0.1)
    g = torch.Generator().manual_seed(32)
    B, T, N = 3, 5, 128
    x0 = torch.randn(B, T, N, 6, generator=g)
    body = torch.randn(B, 20, 3, generator=g) * 0.2
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous()
    t = torch.randn(B, T, 3, generator=g) * 0.1
    h0 = torch.zeros(6, B, 64)

    def fwd():
        x = x0.clone().to(dev)
        with torch.no_grad():
            out = up(x, h0.to(dev), h0.clone().to(dev), body.to(dev), R.to(dev), t.to(dev))
        return x.cpu(), out[0].cpu(), up.arena("eval").get("vec", (B * T, 64)).cpu().clone()
    x32, l32, v32 = fwd()
    up.precision = "bf16"
    xb, lb, vb = fwd()
    xb2, lb2, vb2 = fwd()
    assert torch.equal(lb, lb2) and torch.equal(vb, vb2)
    assert torch.equal(x32, xb)                                       # Transform2H and its write-back: untouched by the mode
    assert not torch.equal(v32, vb)
    up.precision = "fp32"
    x32b, l32b, v32b = fwd()
    assert torch.equal(l32, l32b) and torch.equal(v32, v32b)
    # emulation of the pooled features from the transformed points
    pts = x32.view(B * T, N, 6).double()
    cur = _bf(pts.float())
    feats4 = cur[..., :4]

    def layer(a, conv, bn):
        s = bn.weight.detach().cpu() / torch.sqrt(bn.running_var.cpu() + bn.eps)
        Wf = (s[:, None] * conv.weight.detach().cpu().view(conv.weight.shape[0], -1)).float()
        bf = ((conv.bias.detach().cpu() - bn.running_mean.cpu()) * s + bn.bias.detach().cpu()).float()
        return torch.relu(a @ _bf(Wf).t() + bf.double())
    m0, m1 = up.module0, up.module1.gpointnet
    a = layer(cur, m0.conv1, m0.cb1)
    a = layer(_bf(a.float()), m0.conv2, m0.cb2)
    a = layer(_bf(a.float()), m0.conv3, m0.cb3)
    z = torch.cat([feats4, _bf(a.float())], dim=-1)
    gq = layer(z, m1.conv1, m1.cb1)
    gq = layer(_bf(gq.float()), m1.conv2, m1.cb2)
    gq = layer(_bf(gq.float()), m1.conv3, m1.cb3)                     # [F, N, 64] fp32 features (not rounded: pooled in fp32)
    sc = gq @ m1.attn.weight.detach().cpu().double().view(-1) + m1.attn.bias.detach().cpu().double()
    w = torch.softmax(sc, dim=1)
    want = (w.unsqueeze(-1) * gq).sum(dim=1)
    err = (vb.double() - want).abs()
    scale = float(want.abs().max())
    # an activation that lands on the other side of a bf16 boundary (fp32 vs float64 sums) moves a feature by 2^-8 of one term
    assert float(err.max()) < 5e-3 * scale and float(err.mean()) < 2e-4 * scale, (float(err.max()), float(err.mean()), scale)
    assert float((v32 - vb).abs().max()) < 5e-2 * scale
    assert float((l32 - lb).abs().max()) < 2e-2


def test_config5_full_size_upper_lower_bf16_forward():
    """The Upper_Net / Lower_Net half of BASELINE config 5 at its FULL size (B = 2048, T = 16, N = 256: 32 768 frames, 491 520 skeleton
    rows through the ST-GCN) with precision = "bf16", head pose given.  Size-independent properties: (i) outputs finite; (ii) sequences are
    independent in eval mode, so the first 3 sequences of the big batch equal the same 3 run alone up to what the fp32 products around
    the bf16 kernels do differently at another batch size (other tile shapes, other summation order; and a bf16 operand that lands on
    the other side of a rounding boundary because of it): 2e-3; (iii) against the fp32 path on the same inputs the joints move by
    < 3e-2 (5e-3 on average)."""
    from mmego_amd import nets
    dev = _dev()
    torch.manual_seed(41)
    up, lo = nets.UpperNet().to(dev).eval(), nets.LowerNet(64).to(dev).eval()
    with torch.no_grad():
        for m in list(up.modules()) + list(lo.modules()):
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.6, 1.4)
    B, T, N = 2048, 16, 256
    g = torch.Generator().manual_seed(42)
    x0 = (torch.randn(B, T, N, 6, generator=g) * 0.4).to(dev)
    # ONE skeleton for every sequence: the reference indexes the bone table by FRAME index modulo B (quirk Q2), so with different
    # skeletons per sequence a sub-batch would not see the bones the big batch gives the same frames
    body = (torch.randn(1, 20, 3, generator=g) * 0.2).expand(B, 20, 3).contiguous().to(dev)
    R = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(dev)
    t = (torch.randn(B, T, 3, generator=g) * 0.1).to(dev)

    def fwd(n):
        h0 = torch.zeros(6, n, 64, device=dev)
        with torch.no_grad():
            x = x0[:n].clone()
            u = up(x, h0, h0.clone(), body[:n].contiguous(), R[:n].contiguous(), t[:n].contiguous())[0]
            l = lo(u, x, None, None, None, None, body[:n].contiguous(), R[:n].contiguous(), t[:n].contiguous())[0]
        return u.clone(), l.clone()
    up.precision = lo.precision = "bf16"
    ub, lb = fwd(B)
    us, ls = fwd(3)
    up.precision = lo.precision = "fp32"
    u32, l32 = fwd(B)
    assert bool(torch.isfinite(ub).all()) and bool(torch.isfinite(lb).all())
    assert float((ub[:3] - us).abs().max()) < 2e-3 and float((lb[:3] - ls).abs().max()) < 2e-3, \
        (float((ub[:3] - us).abs().max()), float((lb[:3] - ls).abs().max()))
    du, dl = (ub - u32).abs(), (lb - l32).abs()
    assert float(du.max()) < 3e-2 and float(dl.max()) < 3e-2, (float(du.max()), float(dl.max()))
    assert float(du.mean()) < 5e-3 and float(dl.mean()) < 5e-3, (float(du.mean()), float(dl.mean()))


@pytest.mark.parametrize("rows,dims", [(1000, (6, 16, 32, 61)), (64, (28, 32, 48, 64)), (4099, (6, 8, 16, 24))])
def test_mlp3_eval_bf16_matches_emulation(rows, dims):
    """mmego_mlp3_eval_bf16 (mlp3_bf16.hip) against the three stages in float64 with the kernel's roundings (folded weights and every
    stage's input rounded to bf16; biases, ReLU, output fp32): 5e-3 of the output scale at worst (an activation on the other side of a
    bf16 boundary), 2e-4 on average.  Row counts with a ragged last 64-row tile, BasePointNet's / GlobalPointNet's / PointNet's widths."""
    from mmego_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, dims[0] + 2, generator=g)
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(3)]
    bn = [(torch.rand(dims[i + 1], generator=g) + 0.5, torch.randn(dims[i + 1], generator=g) * 0.1, torch.randn(dims[i + 1], generator=g) * 0.2,
           torch.rand(dims[i + 1], generator=g) + 0.5) for i in range(3)]
    eps = 1e-5
    xd = x.to(dev)
    y = torch.full((rows, dims[3] + 3), 7.0, device=dev)
    devt = [[t.to(dev) for t in b] for b in bn]
    bnp = torch.tensor([t.data_ptr() for b in devt for t in b], dtype=torch.int64)
    Wd, bd = [w.to(dev) for w in Ws], [b.to(dev) for b in bs]
    hip.call("mlp3_eval_bf16", xd[:, 1:], xd.stride(0), rows, dims[0], Wd[0], bd[0], dims[1], Wd[1], bd[1], dims[2], Wd[2], bd[2], dims[3],
             y[:, 2:], y.stride(0), bnp, eps, 1)
    torch.cuda.synchronize()
    # pre = 1: the first input column, in fp32, right in front of every row's outputs; nothing else outside them
    assert torch.all(y[:, 0] == 7.0) and torch.equal(y[:, 1], xd[:, 1]) and torch.all(y[:, 2 + dims[3]:] == 7.0)
    a = _bf(x[:, 1:1 + dims[0]])
    for i in range(3):
        gam, bet, mean, var = bn[i]
        sc = (gam / torch.sqrt(var + eps)).float()
        Wf = (sc[:, None] * Ws[i]).float()
        bf = ((bs[i] - mean) * sc + bet).float()
        a = torch.relu(a @ _bf(Wf).t() + bf.double())
        if i < 2:
            a = _bf(a.float())
    got = y[:, 2:2 + dims[3]].cpu().double()
    err = (got - a).abs()
    scale = float(a.abs().max())
    assert float(err.max()) < 5e-3 * scale and float(err.mean()) < 2e-4 * scale, (float(err.max()), float(err.mean()), scale)
