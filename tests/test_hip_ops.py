"""GPU parity of the unit operators (through the C ABI) against the CPU oracle.
fp32 MFMA is an exact-fp32 fma chain, so the tolerance is the fp32 re-association noise floor."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.build()
    from gecco_amd import hip_ops
    return hip_ops


def _rs(seed):
    return np.random.RandomState(seed)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _close(got, ref, tol=TOL):
    e = cpu_ref.rel_err(got.cpu(), ref)
    assert e[0] <= tol, e


@pytest.mark.parametrize("B,rows,K,Nout", [(2, 256, 128, 256), (3, 100, 36, 70), (2, 64, 384, 768), (1, 2048, 384, 384),
                                           (2, 200, 672, 128), (2, 64, 768, 384)])
def test_linear_fused(ops, B, rows, K, Nout):
    rs = _rs(B * 1000 + rows + K + Nout)
    A, W, b = _t(rs.randn(B, rows, K)), _t(rs.randn(Nout, K) / math.sqrt(K)), _t(rs.randn(Nout))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    R = _t(rs.randn(B, rows, Nout))
    alpha = _t(np.array(0.9))
    ref = R + cpu_ref.gaussian_activation(F.linear(A * pa[:, None] + po[:, None], W, b), alpha)
    out, stats = ops.linear(A.cuda(), W.cuda(), b.cuda(), (pa.cuda(), po.cuda()), alpha.cuda(), R.cuda(), want_stats=True)
    _close(out, ref)
    s = stats.cpu().double().sum(1)  # (B, 2, Nout)
    _close(s[:, 0], ref.double().sum(1), 1e-4)
    _close(s[:, 1], (ref.double() ** 2).sum(1), 1e-5)
    # plain variant: no prologue / bias / act / residual
    out2 = ops.linear(A.cuda(), W.cuda())
    _close(out2, F.linear(A, W))
    # in-place residual (C aliases residual) as the layer uses it
    Rc = R.cuda().clone()
    ops.linear(A.cuda(), W.cuda(), b.cuda(), residual=Rc, out=Rc)
    _close(Rc, R + F.linear(A, W, b))


@pytest.mark.parametrize("B,rows,K,Nout", [(2, 256, 128, 256), (1, 2048, 384, 768), (2, 300, 768, 384), (3, 64, 384, 768),
                                           (2, 64, 768, 384), (2, 100, 128, 200)])
def test_linear_split_bf16(ops, B, rows, K, Nout):
    """precision="bf16x3" on the unit operator: same fused contract, ~2^-16 per-product error."""
    rs = _rs(rows + K)
    A, W, b = _t(rs.randn(B, rows, K) * 3), _t(rs.randn(Nout, K) / math.sqrt(K)), _t(rs.randn(Nout))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    R = _t(rs.randn(B, rows, Nout))
    alpha = _t(np.array(1.1))
    pre = F.linear((A * pa[:, None] + po[:, None]).double(), W.double(), b.double())
    ref = R.double() + cpu_ref.gaussian_activation(pre, alpha.double())
    out, stats = ops.linear(A.cuda(), W.cuda(), b.cuda(), (pa.cuda(), po.cuda()), alpha.cuda(), R.cuda(), want_stats=True,
                            precision="bf16x3")
    e = cpu_ref.rel_err(out.cpu(), ref)
    assert e[0] < 1e-4, e
    _close(stats.cpu().double().sum(1)[:, 0], ref.sum(1), 2e-4)
    exact = ops.linear(A.cuda(), W.cuda(), b.cuda(), (pa.cuda(), po.cuda()), alpha.cuda(), R.cuda())
    e32 = cpu_ref.rel_err(exact.cpu(), ref)
    assert e32[0] < e[0]  # the exact-fp32 mode is (of course) closer; both are far inside 1e-3


@pytest.mark.parametrize("B,rows,K,n1,n2", [(2, 256, 128, 256, 128), (1, 2048, 384, 768, 384), (2, 200, 64, 96, 64), (3, 64, 384, 768, 384)])
@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 1e-4), ("fp16", 2e-3)])
def test_linear_pair(ops, B, rows, K, n1, n2, precision, tol):
    """kv_proj | q_proj in one launch: each half must equal its own linear (bit-for-bit in the same arithmetic)."""
    rs = _rs(rows + K + n1)
    A = _t(rs.randn(B, rows, K) * 2)
    W1, W2, b2 = _t(rs.randn(n1, K) / math.sqrt(K)), _t(rs.randn(n2, K) / math.sqrt(K)), _t(rs.randn(n2))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    An = (A * pa[:, None] + po[:, None]).double()
    c1, c2 = ops.linear_pair(A.cuda(), W1.cuda(), None, W2.cuda(), b2.cuda(), (pa.cuda(), po.cuda()), precision=precision)
    _close(c1, F.linear(An, W1.double()), tol)
    _close(c2, F.linear(An, W2.double(), b2.double()), tol)
    s1 = ops.linear(A.cuda(), W1.cuda(), None, (pa.cuda(), po.cuda()), precision=precision)
    s2 = ops.linear(A.cuda(), W2.cuda(), b2.cuda(), (pa.cuda(), po.cuda()), precision=precision)
    assert torch.equal(c1, s1) and torch.equal(c2, s2)


@pytest.mark.parametrize("B,rows,C,G,ctx", [(2, 256, 128, 32, 1), (3, 77, 64, 32, 1), (2, 64, 384, 32, 3), (2, 300, 672, 16, 0)])
def test_adagn(ops, B, rows, C, G, ctx):
    rs = _rs(rows + C)
    x = _t(rs.randn(B, rows, C) * 2 + 0.7)
    if ctx:
        t = _t(rs.randn(B, 1, ctx))
        p = {"scale.weight": _t(rs.randn(C, ctx) * .2), "scale.bias": _t(1 + .1 * rs.randn(C)),
             "bias.weight": _t(rs.randn(C, ctx) * .2), "bias.bias": _t(.1 * rs.randn(C))}
        ref = cpu_ref.adagn(x, t, p, "", G)
        params = [p[k].cuda() for k in ("scale.weight", "scale.bias", "bias.weight", "bias.bias")]
        got = ops.adagn(x.cuda(), t.cuda(), params, G)
    else:
        ref = cpu_ref.group_norm_bnc(x, G)
        got = ops.adagn(x.cuda(), None, None, G)
    _close(got, ref)


@pytest.mark.parametrize("B,rows,K,Nout", [(2, 256, 128, 256), (1, 2048, 384, 768), (2, 300, 768, 384), (3, 64, 384, 768),
                                           (2, 64, 768, 384), (2, 100, 128, 200), (1, 130, 32, 132)])
def test_linear_fp16(ops, B, rows, K, Nout):
    """precision="fp16": the fused contract with fp16-rounded operands; error ~2^-11 per operand."""
    rs = _rs(rows + K + 7)
    A, W, b = _t(rs.randn(B, rows, K) * 3), _t(rs.randn(Nout, K) / math.sqrt(K)), _t(rs.randn(Nout))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    R = _t(rs.randn(B, rows, Nout))
    alpha = _t(np.array(1.1))
    An = (A * pa[:, None] + po[:, None])
    pre = F.linear(An.double(), W.double(), b.double())
    ref = R.double() + cpu_ref.gaussian_activation(pre, alpha.double())
    out, stats = ops.linear(A.cuda(), W.cuda(), b.cuda(), (pa.cuda(), po.cuda()), alpha.cuda(), R.cuda(), want_stats=True,
                            precision="fp16")
    e = cpu_ref.rel_err(out.cpu(), ref)
    assert e[0] < 2e-3, e
    _close(stats.cpu().double().sum(1)[:, 0], ref.sum(1), 5e-3)
    # the arithmetic is exactly "round both operands to fp16, multiply exactly, accumulate in fp32"
    # (the kernel's affine is one fma: emulate it as the correctly rounded fp32 of the exact a*x+o)
    An16 = (A.double() * pa[:, None].double() + po[:, None].double()).float().half().double()
    emu = F.linear(An16, W.half().double(), b.double())
    plain = ops.linear(A.cuda(), W.cuda(), b.cuda(), (pa.cuda(), po.cuda()), precision="fp16")
    e = cpu_ref.rel_err(plain.cpu(), emu)
    assert e[1] <= 2e-6 and e[0] <= 2e-4, e   # rms at the fp32 accumulation floor; a rare fp16 tie may flip one operand


def test_fp16_stored_intermediates(ops):
    """The fp16-mode launches with fp16 TENSORS (what the network entry point issues) give the same bits as the unit
    operators that take fp32 tensors and round on the fly."""
    rs = _rs(77)
    B, N, Cc, H = 2, 256, 128, 8
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    a, o = _t(1 + 0.3 * rs.randn(B, Cc)).cuda(), _t(0.3 * rs.randn(B, Cc)).cuda()
    Wkv, Wq, bq = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(Cc, Cc) / 11).cuda(), _t(rs.randn(Cc) * .1).cuda()
    y16 = ops.affine_cast_f16(x, a, o)
    assert torch.equal(y16, torch.addcmul(o[:, None].double(), x.double(), a[:, None].double()).float().half())
    kv16, q16 = ops.linear_pair_f16io(y16, Wkv, None, Wq, bq)
    kv32, q32 = ops.linear_pair(x, Wkv, None, Wq, bq, (a, o), precision="fp16")
    assert torch.equal(kv16, kv32.half()) and torch.equal(q16, q32.half())
    ind = _t(rs.randn(1, H, 64, Cc // H)).cuda()
    assert torch.equal(ops.pool_attn_f16in(kv16, ind, H), ops.pool_attn(kv16.float(), ind, H, precision="fp16"))
    kvh = _t(rs.randn(B, 64, 2 * Cc)).cuda()
    att16 = ops.unpool_attn_f16io(q16, kvh, H)
    assert torch.equal(att16, ops.unpool_attn(q16.float(), kvh, H, precision="fp16").half())
    Wo, bo = _t(rs.randn(Cc, Cc) / 11).cuda(), _t(rs.randn(Cc) * .1).cuda()
    r1, r2 = x.clone(), x.clone()
    _, s1 = ops.linear_f16io(att16, Wo, bo, residual=r1, want_stats=True, out=r1)
    _, s2 = ops.linear(att16.float(), Wo, bo, residual=r2, want_stats=True, out=r2, precision="fp16")
    assert torch.equal(r1, r2) and torch.equal(s1, s2)
    alpha = _t(np.array(0.9)).cuda()
    W0, b0 = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(2 * Cc) * .1).cuda()
    h16 = ops.linear_f16io(y16, W0, b0, act_alpha=alpha, out_f16=True)
    h32 = ops.linear(x, W0, b0, (a, o), act_alpha=alpha, precision="fp16")
    assert torch.equal(h16, h32.half())


@pytest.mark.parametrize("B,N,Cc", [(2, 256, 128), (2, 128, 384), (1, 384, 256), (1, 128, 512)])
def test_linear_astat_matches_two_launch_form(ops, B, N, Cc):
    """The one-pass A-stationary kernel gives the bits of the cast pass + streaming fp16 GEMM (pair, and single + act)."""
    rs = _rs(N + Cc)
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    a, o = _t(1 + 0.3 * rs.randn(B, Cc)).cuda(), _t(0.3 * rs.randn(B, Cc)).cuda()
    Wkv, Wq, bq = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(Cc, Cc) / 11).cuda(), _t(rs.randn(Cc) * .1).cuda()
    y16 = ops.affine_cast_f16(x, a, o)
    kv_ref, q_ref = ops.linear_pair_f16io(y16, Wkv, None, Wq, bq)
    kv, q = ops.linear_astat_f16(x, (a, o), Wkv, None, Wq, bq)
    assert torch.equal(kv, kv_ref) and torch.equal(q, q_ref)
    alpha = _t(np.array(0.9)).cuda()
    W0, b0 = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(2 * Cc) * .1).cuda()
    h_ref = ops.linear_f16io(y16, W0, b0, act_alpha=alpha, out_f16=True)
    h = ops.linear_astat_f16(x, (a, o), W0, b0, act_alpha=alpha)
    assert torch.equal(h, h_ref)


@pytest.mark.parametrize("B,N,Cc,H", [(2, 256, 128, 8), (3, 384, 384, 8), (2, 128, 256, 4)])
def test_head_major_layout_is_a_pure_permutation(ops, B, N, Cc, H):
    """head_dim > 0: the A-stationary kernel stores "b n (g d) -> b g n d" of the row-major result (same bits), and
    the pool / unpool attention kernels give the same bits reading that layout (ragged N included: N = 384 has a last
    key tile the split does not fill)."""
    rs = _rs(N + Cc + H)
    hd = Cc // H
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    a, o = _t(1 + 0.3 * rs.randn(B, Cc)).cuda(), _t(0.3 * rs.randn(B, Cc)).cuda()
    Wkv, Wq, bq = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(Cc, Cc) / 11).cuda(), _t(rs.randn(Cc) * .1).cuda()
    kv, q = ops.linear_astat_f16(x, (a, o), Wkv, None, Wq, bq)
    kv_hm, q_hm = ops.linear_astat_f16(x, (a, o), Wkv, None, Wq, bq, head_dim=hd)
    assert kv_hm.shape == (B, 2 * H, N, hd) and q_hm.shape == (B, H, N, hd)
    assert torch.equal(kv_hm, kv.view(B, N, 2 * H, hd).permute(0, 2, 1, 3))
    assert torch.equal(q_hm, q.view(B, N, H, hd).permute(0, 2, 1, 3))
    ind = _t(rs.randn(1, H, 64, hd)).cuda()
    assert torch.equal(ops.pool_attn_f16in(kv_hm, ind, H, head_major=True), ops.pool_attn_f16in(kv, ind, H))
    kvh = _t(rs.randn(B, 64, 2 * Cc)).cuda()
    assert torch.equal(ops.unpool_attn_f16io(q_hm, kvh, H, head_major=True), ops.unpool_attn_f16io(q, kvh, H))


@pytest.mark.parametrize("B,N,Cc", [(2, 256, 128), (3, 384, 384), (2, 128, 256)])
def test_mlp_fused_matches_the_two_launch_form(ops, B, N, Cc):
    """x += mlp.2(act(mlp.0(AdaGN(x)))) in one launch against the A-stationary mlp.0 kernel followed by the fp16-A mlp.2
    kernel with residual: same rounding points, same k order, same order of the GroupNorm partial sums — the same bits."""
    rs = _rs(N + Cc)
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    a, o = _t(1 + 0.3 * rs.randn(B, Cc)).cuda(), _t(0.3 * rs.randn(B, Cc)).cuda()
    W0, b0 = _t(rs.randn(2 * Cc, Cc) / 11).cuda(), _t(rs.randn(2 * Cc) * .1).cuda()
    W2, b2 = _t(rs.randn(Cc, 2 * Cc) / 15).cuda(), _t(rs.randn(Cc) * .1).cuda()
    alpha = _t(np.array(0.9)).cuda()
    h16 = ops.linear_astat_f16(x, (a, o), W0, b0, act_alpha=alpha)
    ref, st_ref = ops.linear_f16io(h16, W2, b2, residual=x, want_stats=True)
    got, st = ops.mlp_fused_f16(x.clone(), (a, o), W0, b0, W2, b2, act_alpha=alpha, want_stats=True)
    assert torch.equal(got, ref)
    assert st_ref.shape == st.shape and torch.equal(st, st_ref)
    assert torch.equal(ops.mlp_fused_f16(x.clone(), (a, o), W0, b0, W2, b2, act_alpha=alpha)[0], ref)   # stats optional


@pytest.mark.parametrize("B,N,Cc,H", [(2, 256, 128, 8), (3, 384, 384, 8), (2, 128, 256, 8)])
def test_unpool_outproj_fused_gives_the_bits_of_the_two_launch_form(ops, B, N, Cc, H):
    """Unpool attention + out_proj + residual + GroupNorm partials in one launch against unpool_attn_f16io (head-major
    q) followed by the fp16-A out_proj GEMM: same fragments, same softmax, same roundings, same epilogue — same bits."""
    rs = _rs(N + Cc + 1)
    hd = Cc // H
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    q = (_t(rs.randn(B, H, N, hd)).cuda()).half()
    kvh = _t(rs.randn(B, 64, 2 * Cc)).cuda()
    W, bias = _t(rs.randn(Cc, Cc) / 13).cuda(), _t(rs.randn(Cc) * .1).cuda()
    att = ops.unpool_attn_f16io(q, kvh, H, head_major=True)
    ref, st_ref = ops.linear_f16io(att, W, bias, residual=x, want_stats=True)
    got, st = ops.unpool_outproj_f16(x.clone(), q, kvh, W, bias, H, want_stats=True)
    assert torch.equal(got, ref)
    assert st.shape == st_ref.shape and torch.equal(st, st_ref)
    assert torch.equal(ops.unpool_outproj_f16(x.clone(), q, kvh, W, None, H)[0],
                       ops.linear_f16io(att, W, None, residual=x))   # bias and stats optional


@pytest.mark.parametrize("B,N,Cc,H", [(2, 256, 128, 8), (3, 384, 384, 8), (2, 128, 256, 8), (5, 2048, 384, 8), (3, 512, 512, 8)])
def test_unpool_outproj_h8_fused_matches_the_two_launch_form(ops, B, N, Cc, H):
    """Mixed mode: unpool attention + h8 out_proj + residual + GroupNorm partials in ONE launch (unpool_outproj_h8.hip; reference
    models/set_transformer.py:70-75, 112, 164) against the two launches it replaces — the fp16 attention writing the h8 activation
    image, then gemm_h8_areg.hip: the same attention bits and the same hi / lo split of its output, so the results differ by the
    fp32 summation order of the product alone; and against float64 of what the image decodes to."""
    rs = _rs(N + Cc + 2)
    hd = Cc // H
    x = _t(rs.randn(B, N, Cc) * 2).cuda()
    q = (_t(rs.randn(B, H, N, hd)).cuda()).half()
    kvh = _t(rs.randn(B, 64, 2 * Cc)).cuda()
    W, bias = _t(rs.randn(Cc, Cc) / 13).cuda(), _t(rs.randn(Cc) * .1).cuda()
    img = ops.unpool_attn_h8img(q, kvh, H)
    ref, st_ref = ops.linear_h8_areg(img, W, bias, residual=x, want_stats=True)
    got, st = ops.unpool_outproj_h8(x.clone(), q, kvh, W, bias, H, want_stats=True)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 4e-6 * scale, (got - ref).abs().max().item() / scale
    att = ops.decode_h8_image(img).double()
    ref64 = x.double() + att @ W.double().t() + bias.double()
    assert (got.double() - ref64).abs().max().item() <= 3e-5 * scale
    assert st.shape == st_ref.shape
    g4 = got.double().reshape(B, N // 128, 128, Cc)
    assert (st[:, :, 0].double() - g4.sum(2)).abs().max().item() <= 1e-3
    assert (st[:, :, 1].double() - (g4 * g4).sum(2)).abs().max().item() <= 1e-5 * (g4 * g4).sum(2).max().item()
    # bias and statistics optional; a ready weight image gives the same bits; run to run reproducible
    lib = ops._lib.load()
    ws = torch.empty(lib.gecco_unpool_outproj_h8_wsplit_bytes(B, Cc, H), dtype=torch.uint8, device="cuda")
    a = ops.unpool_outproj_h8(x.clone(), q, kvh, W, None, H, wsplit=ws)[0]
    c = ops.unpool_outproj_h8(x.clone(), q, kvh, W, None, H, wsplit=ws, image_ready=True)[0]
    assert torch.equal(a, c)
    assert (a - ops.linear_h8_areg(img, W, None, residual=x)).abs().max().item() <= 4e-6 * scale
    assert torch.equal(ops.unpool_outproj_h8(x.clone(), q, kvh, W, bias, H)[0], got)


@pytest.mark.parametrize("B,rows,act,K", [(2, 256, "gauss", 384), (1, 128, "relu", 384), (3, 384, "none", 384), (5, 2048, "gauss", 384), (300, 128, "gauss", 384),
                                          (2, 256, "gauss", 256), (3, 384, "relu", 256), (300, 128, "gauss", 256), (1, 128, "none", 256),
                                          (2, 256, "gauss", 128), (3, 384, "none", 128), (600, 128, "gauss", 128), (1, 128, "relu", 128),
                                          (2, 256, "gauss", 512), (3, 384, "relu", 512), (300, 128, "gauss", 512), (1, 128, "none", 512)])
def test_mlp_fused_w_vs_float64(ops, B, rows, act, K):
    """"w2" mode: the point MLP of a layer in ONE launch with the hidden layer kept in registers (mlp_fused_w.hip;
    models/set_transformer.py:164-166, mlp.py:5-39, activation.py:17-24, normalization.py:36-44) against float64, stage by stage:
    mlp.0's pre-activations (two-term operands on both sides: the fp6 second terms leave ~1e-5), the output against a reference built
    from the kernel's OWN hidden layer rounded to fp16 (the second product alone: two-term weights, ~2e-5), and the whole MLP against
    the exact one (the hidden layer's dropped second term: ~2e-4 of the MLP's scale).  B = 300: more row tiles than CUs (blocks are
    persistent: several tiles per block, the weight stream wraps).  K = feature_dim: 384 (6 groups of 64: three per ring stage), 256 and 128 (two / one
    group per stage: other stage sizes, set counts and activation schedules of the same kernel template)."""
    Wd = 2 * K
    rs = _rs(B + rows + len(act) + abs(384 - K) + (7 if K > 384 else 0))
    x, W0, b0 = _t(rs.randn(B, rows, K)), _t(rs.randn(Wd, K) / math.sqrt(K)), _t(rs.randn(Wd) / math.sqrt(K))
    W2, b2 = _t(rs.randn(K, Wd) / math.sqrt(Wd)), _t(rs.randn(K) / math.sqrt(Wd))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    alpha = _t(np.array(0.9))
    kw = dict(act_alpha=alpha.cuda()) if act == "gauss" else dict(act="relu") if act == "relu" else {}
    xc, pro = x.cuda(), (pa.cuda(), po.cuda())
    dbg = torch.zeros(B, rows, Wd, device="cuda")
    got, st = ops.mlp_fused_w(xc, pro, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), want_stats=True, out=torch.empty_like(xc), dbg_u=dbg, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()

    def actf(u):
        return (torch.exp(-u * u / (2 * 0.9 ** 2)) - 0.7) / 0.28 if act == "gauss" else torch.relu(u) if act == "relu" else u
    nb = min(B, 6)   # float64 on the host: the first clouds
    y = torch.addcmul(po[:nb, None], x[:nb], pa[:nb, None]).double()   # fmaf, as the kernel forms it
    u = F.linear(y, W0.double(), b0.double())
    # the kernel's pre-activations carry the Gaussian activation's argument scale sqrt(log2(e) / 2) / |alpha| (folded into mlp.0's weights)
    us = dbg[:nb].cpu().double() / (0.84932180028801907 / 0.9 if act == "gauss" else 1.0)
    eu = cpu_ref.rel_err(us, u)
    hk = actf(us).half().double()
    mlp_own = F.linear(hk, W2.double(), b2.double())
    mlp_ref = F.linear(actf(u), W2.double(), b2.double())
    scale = mlp_ref.abs().max().item()
    d_own = ((got[:nb].cpu().double() - x[:nb].double()) - mlp_own).abs().max().item() / scale
    d_ref = ((got[:nb].cpu().double() - x[:nb].double()) - mlp_ref).abs().max().item() / scale
    print(f"w2 point MLP ({act}, B={B}, rows={rows}, d={K}): pre-activations {eu[0]:.2e}, second product {d_own:.2e}, whole MLP {d_ref:.2e} of its scale")
    assert eu[0] <= 4e-5, eu
    assert d_own <= 6e-5, d_own
    assert d_ref <= 5e-4, d_ref
    g4 = got.double().reshape(B, rows // 128, 128, K)
    assert (st[:, :, 0].double() - g4.sum(2)).abs().max().item() <= 1e-3 * max(1.0, got.abs().max().item())
    assert (st[:, :, 1].double() - (g4 * g4).sum(2)).abs().max().item() <= 1e-5 * (g4 * g4).sum(2).max().item()
    # in place = out of place; a ready weight stream gives the same bits; bias / statistics optional; reproducible
    lib = ops._lib.load()
    assert torch.equal(ops.mlp_fused_w(xc.clone(), pro, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), **kw)[0], got)
    ws = torch.empty(lib.gecco_mlp_fused_w_wsplit_bytes(K, Wd), dtype=torch.uint8, device="cuda")
    a = ops.mlp_fused_w(xc.clone(), pro, W0.cuda(), None, W2.cuda(), None, wsplit=ws, **kw)[0]
    c = ops.mlp_fused_w(xc.clone(), pro, W0.cuda(), None, W2.cuda(), None, wsplit=ws, image_ready=True, **kw)[0]
    assert torch.equal(a, c)


@pytest.mark.parametrize("K", [384, 256, 128, 512])
def test_mlp_fused_w_on_outlier_weights_and_activations(ops, K):
    """The fp6 second terms carry a block scale per lane and 64-k group (no fixed range to leave); the fp16 main terms saturate at
    +-3584 like every h8 operand (h8_scales.h): |w| = 8 entries, an outlier channel of AdaGN(x) (|y| ~ 500) and a weight matrix 100 x
    the usual scale stay finite and proportionate."""
    Wd, B, rows = 2 * K, 2, 256
    rs = _rs(77 + abs(384 - K) + (7 if K > 384 else 0))
    x, W0, b0 = _t(rs.randn(B, rows, K)), _t(rs.randn(Wd, K) / math.sqrt(K)), _t(rs.randn(Wd) / math.sqrt(K))
    W2, b2 = _t(rs.randn(K, Wd) / math.sqrt(Wd)), _t(rs.randn(K) / math.sqrt(Wd))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    W0[rs.randint(0, Wd, 40), rs.randint(0, K, 40)] = 8.0
    W2[rs.randint(0, K, 40), rs.randint(0, Wd, 40)] = -8.0
    pa[:, 7] = 400.0
    for wscale in (1.0, 100.0):
        got = ops.mlp_fused_w(x.cuda(), (pa.cuda(), po.cuda()), (W0 * wscale).cuda(), b0.cuda(), W2.cuda(), b2.cuda(), act="relu",
                              out=torch.empty_like(x.cuda()))[0]
        y = torch.addcmul(po[:, None], x, pa[:, None]).double().clamp(-3584, 3584)
        ref = x.double() + F.linear(torch.relu(F.linear(y, (W0 * wscale).double(), b0.double())).clamp(max=3584), W2.double(), b2.double())
        e = cpu_ref.rel_err(got.cpu().double(), ref)
        print(f"w2 point MLP, d={K}, outliers, weight scale {wscale}: {e[0]:.2e}")
        assert torch.isfinite(got).all() and e[0] <= 1e-3, e


def test_adagn_large_mean(ops):
    """E[x^2]-mean^2 cancellation: mean 50x the std must still be accurate (fp64 combine)."""
    rs = _rs(5)
    x = _t(rs.randn(2, 2048, 64) * 0.1 + 5.0)
    _close(ops.adagn(x.cuda(), None, None, 32), cpu_ref.group_norm_bnc(x.double(), 32).float(), 2e-3)


@pytest.mark.parametrize("B,N,C,H", [(2, 256, 128, 8), (2, 1000, 384, 8), (1, 64, 64, 8), (3, 33, 512, 8), (2, 2048, 256, 8)])
@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 1e-4), ("fp16", 2e-3)])
def test_pool_attn(ops, B, N, C, H, precision, tol):
    rs = _rs(N + C)
    y = _t(rs.randn(B, N, C))
    p = {"kv_proj.weight": _t(rs.randn(2 * C, C) / math.sqrt(C) * 2), "inducers": _t(rs.randn(1, H, 64, C // H)),
         "out_proj.weight": torch.eye(C)}
    ref = cpu_ref.attention_pool(y, p, "", H)
    KV = F.linear(y, p["kv_proj.weight"])
    got = ops.pool_attn(KV.cuda(), p["inducers"].cuda(), H, precision=precision)
    _close(got, ref, tol)


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 1e-4), ("fp16", 2e-3)])
def test_pool_attn_online_softmax_rescale(ops, precision, tol):
    """Force the running max to jump late in the key stream (rule: a rare branch needs its own test)."""
    B, N, C, H = 1, 1024, 128, 8
    rs = _rs(9)
    y = _t(rs.randn(B, N, C))
    ind = _t(rs.randn(1, H, 64, C // H))
    Wkv = _t(rs.randn(2 * C, C) / math.sqrt(C))
    KV = F.linear(y, Wkv)
    KV[0, 900, :C] = ind[0, :, 3, :].reshape(-1) * 6.0   # key 900 aligned with query 3 of every head
    KV[0, 17, :C] = ind[0, :, 5, :].reshape(-1) * 6.0
    p = {"inducers": ind}
    hd = C // H
    k = KV[..., :C].reshape(B, N, H, hd).permute(0, 2, 1, 3)
    v = KV[..., C:].reshape(B, N, H, hd).permute(0, 2, 1, 3)
    a = torch.softmax(ind @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    ref = (a @ v).permute(0, 2, 1, 3).reshape(B, 64, C)
    _close(ops.pool_attn(KV.cuda(), ind.cuda(), H, precision=precision), ref, tol)


@pytest.mark.parametrize("B,N,C,H", [(2, 256, 128, 8), (2, 1000, 384, 8), (1, 50, 64, 8), (2, 640, 512, 8), (2, 300, 256, 8)])
@pytest.mark.parametrize("precision,tol", [("fp32", TOL), ("bf16x3", 1e-4), ("fp16", 2e-3)])
def test_unpool_attn(ops, B, N, C, H, precision, tol):
    rs = _rs(N + C + 1)
    y, h = _t(rs.randn(B, N, C)), _t(rs.randn(B, 64, C))
    p = {"in_proj_weight": _t(rs.randn(3 * C, C) / math.sqrt(C) * 1.5), "in_proj_bias": _t(rs.randn(3 * C) * .1),
         "out_proj.weight": torch.eye(C), "out_proj.bias": torch.zeros(C)}
    ref = cpu_ref.mha_unpool(y, h, p, "", H)
    q = F.linear(y, p["in_proj_weight"][:C], p["in_proj_bias"][:C])
    kvh = F.linear(h, p["in_proj_weight"][C:], p["in_proj_bias"][C:])
    _close(ops.unpool_attn(q.cuda(), kvh.cuda(), H, precision=precision), ref, tol)


@pytest.mark.parametrize("B,N,C", [(3, 200, 128), (2, 2048, 384), (33, 2000, 384), (40, 1666, 128)])
def test_lift_lower_edm(ops, B, N, C):
    """(33, 2000, 384), (40, 1666, 128): >= 65536 rows — the 8-rows-per-lane-group form of the lowering kernel (W and the block's
    GroupNorm coefficients through LDS), with blocks that straddle samples and a ragged last block; its results equal the
    one-row-per-group form's (what a single sample runs) bit for bit."""
    rs = _rs(N)
    x = _t(rs.randn(B, N, 3) * 3)
    sigma = _t(np.exp(rs.uniform(np.log(.002), np.log(165), size=B)))
    Wl, bl = _t(rs.randn(C, 3)), _t(rs.randn(C))
    Wo, bo = _t(rs.randn(3, C) / math.sqrt(C)), _t(rs.randn(3))
    c_skip, c_out, c_in, c_noise = cpu_ref.edm_coeffs(sigma)
    coef = ops.edm_coeffs(sigma.cuda())
    _close(coef[: 4 * B].reshape(B, 4), torch.cat([c_skip, c_out, c_in, c_noise], -1).reshape(B, 4), 1e-6)
    _close(coef[4 * B:], c_noise.reshape(B), 1e-6)
    feat_ref = F.linear(c_in * x, Wl, bl)
    feat, stats = ops.lift(x.cuda(), coef, Wl.cuda(), bl.cuda(), want_stats=True)
    _close(feat, feat_ref)
    _close(stats.cpu().double().sum(1)[:, 0], feat_ref.double().sum(1), 1e-4)
    F_ref = F.linear(F.layer_norm(feat_ref, (C,), eps=1e-5), Wo, bo)
    out, raw = ops.lower_edm(feat, x.cuda(), coef, Wo.cuda(), bo.cuda(), want_raw=True)
    _close(raw, F_ref)
    _close(out, c_skip * x + c_out * F_ref)
    # GroupNorm(16) head variant (RayNetwork.output_proj)
    st = ops.col_stats(feat)
    a, o = ops.adagn_coeffs(st, N, None, None, 16)
    F2 = F.linear(cpu_ref.group_norm_bnc(feat_ref, 16), Wo, bo)
    got2 = ops.lower_edm(feat, None, None, Wo.cuda(), bo.cuda(), gn=(a, o))
    _close(got2, F2)
    if B * N >= 65536:
        for b in (0, B // 2, B - 1):
            o1, r1 = ops.lower_edm(feat[b:b + 1].contiguous(), x[b:b + 1].cuda(), coef.view(-1)[4 * b:4 * b + 4].contiguous(), Wo.cuda(), bo.cuda(),
                                   want_raw=True)
            assert torch.equal(o1[0], out[b]) and torch.equal(r1[0], raw[b])
            g1 = ops.lower_edm(feat[b:b + 1].contiguous(), None, None, Wo.cuda(), bo.cuda(), gn=(a[b:b + 1].contiguous(), o[b:b + 1].contiguous()))
            assert torch.equal(g1[0], got2[b])


@pytest.mark.parametrize("B,rows,K,Nout,act", [(2, 512, 384, 768, "gauss"), (1, 256, 256, 512, "relu"), (3, 256, 128, 256, "none"),
                                               (1, 2048, 384, 768, "gauss"), (2, 384, 512, 1024, "gauss")])
def test_linear_h8_image(ops, B, rows, K, Nout, act):
    """mlp.0 of the mixed mode (gemm_h8_astat.hip): fp16 main product + the two cross terms on the fp8 matrix instruction,
    AdaGN prologue, activation, output as the tiled split image — against float64 (models/set_transformer.py:164-166,
    models/mlp.py).  One-term fp16 operands would sit at ~3e-4 here; the cross terms bring the product to split-bf16 accuracy."""
    rs = _rs(B * 7 + rows + K + Nout)
    x, W, b = _t(rs.randn(B, rows, K)), _t(rs.randn(Nout, K) / math.sqrt(K)), _t(rs.randn(Nout) / math.sqrt(K))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    alpha = _t(np.array(0.9))
    u = F.linear((x.double() * pa[:, None].double() + po[:, None].double()), W.double(), b.double())
    if act == "gauss":
        ref = (torch.exp(-u * u / (2 * 0.9 ** 2)) - 0.7) / 0.28
    elif act == "relu":
        ref = torch.relu(u)
    else:
        ref = u
    kw = dict(act_alpha=alpha.cuda()) if act == "gauss" else dict(act="relu") if act == "relu" else {}
    img = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), **kw)
    got = ops.decode_split_image(img).cpu().double()
    e = cpu_ref.rel_err(got, ref)
    assert e[0] < (1e-4 if act == "gauss" else 3e-5), e   # the Gaussian epilogue uses the fast exp (as every kernel here): ~6e-5 max
    # the pre-activation itself, without prologue / bias: the product's own error
    img0 = ops.linear_h8_img(x.cuda(), None, W.cuda(), None)
    e0 = cpu_ref.rel_err(ops.decode_split_image(img0).cpu().double(), F.linear(x.double(), W.double()))
    assert e0[0] < 2e-5, e0
    # image-ready call: the same bits from the weight image the first call left in the scratch
    ws = torch.empty(Nout * K * 4, dtype=torch.uint8, device="cuda")
    a = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), wsplit=ws, **kw)
    c = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), wsplit=ws, image_ready=True, **kw)
    assert torch.equal(a, c) and torch.equal(a, img)


def _h8_outlier_case(rs, B, rows, K, Wd, wmax, ymax):
    """Operands a trained checkpoint may hold: weights ~ N(0, 1 / K) with a few entries up to +-wmax, activations of unit scale with
    an outlier channel reaching +-ymax and isolated large values; a bias that puts an outlier channel into the hidden layer too."""
    x = _t(rs.randn(B, rows, K))
    x[:, :, 7] *= ymax / 4.0
    x[0, 5, 11] = ymax
    x[-1, rows - 3, K - 2] = -ymax
    W0 = _t(rs.randn(Wd, K) / math.sqrt(K))
    W0[:, 7] *= 0.05
    W0[3, 20], W0[Wd - 1, 0], W0[17, K - 1] = wmax, -wmax, wmax / 2
    b0 = _t(rs.randn(Wd) * 0.1)
    b0[9], b0[40] = 0.8 * ymax, -0.5 * ymax
    W2 = _t(rs.randn(K, Wd) / math.sqrt(Wd))
    W2[5, 9], W2[K - 1, Wd - 1], W2[100, 300] = -wmax, wmax, wmax
    return x, W0, b0, W2


@pytest.mark.parametrize("wmax,ymax,bar", [(1.0, 4.0, 3e-5), (8.0, 500.0, 3e-5), (14.0, 448.0, 3e-5), (100.0, 3000.0, 1.5e-3), (1000.0, 1e5, 2e-3)])
def test_h8_products_on_outlier_weights_and_activations(ops, wmax, ymax, bar):
    """The h8 products (fp16 main product + fp8 cross terms) hold their accuracy on weights up to |w| = 14 and activations up to
    |y| = 448 (csrc/h8_scales.h: the fp8 operands' power-of-two scales; round 3's covered |w| <= 1.75, |y| <= 56 and turned NaN above
    |y| = 448); beyond that the cross terms saturate and the product degrades towards one-term fp16 accuracy — gradually, never a
    NaN; beyond |y| = 3584 the operand itself is clamped: finite, and equal to the product of the clamped operand.  mlp.0 -> h8
    image -> mlp.2 with identity activation (the unbounded case; models/mlp.py, set_transformer.py:164-166) against float64."""
    B, rows, K, Wd = 2, 256, 384, 768
    rs = _rs(int(wmax) + int(ymax) % 1000)
    x, W0, b0, W2 = _h8_outlier_case(rs, B, rows, K, Wd, wmax, ymax)
    img = ops.linear_h8_img(x.cuda(), None, W0.cuda(), b0.cuda(), kind=2)
    u_ref = F.linear(x.double().clamp(-3584, 3584), W0.double(), b0.double()).clamp(-3584, 3584)
    u = ops.decode_h8_image(img).cpu()
    assert torch.isfinite(u).all()
    out = ops.linear_h8_areg(img, W2.cuda(), None).cpu().double()
    assert torch.isfinite(out).all()
    e0 = cpu_ref.rel_err(u.double(), u_ref)
    e2 = cpu_ref.rel_err(out, F.linear(u.double(), W2.double()))   # mlp.2's product alone, on the image's own values
    print(f"h8 outliers |w| <= {wmax}, |y| <= {ymax}: mlp.0 {e0[0]:.2e}, mlp.2 {e2[0]:.2e} (bar {bar})")
    assert e0[0] < bar and e2[0] < bar, (e0, e2)


@pytest.mark.parametrize("K,Nout", [(384, 768), (256, 512), (128, 256), (512, 1024)])
def test_h6_cross_terms_hold_the_h8_accuracy(ops, K, Nout):
    """Option "h6" (default on): mlp.0's two cross terms as fp6 (e2m3) x fp6 with one E8M0 scale per lane and 64-k group — the scale
    blocks of v_mfma_scale_f32_32x32x64_f8f6f4 — instead of fp8 with fixed power-of-two scales (gemm_h8_astat_kernel<.., F6>; half the
    matrix cycles for those terms).  Against float64 on plain and on outlier operands it holds the fp8 form's accuracy (the cross terms
    are 2^-12 of the product: 3 mantissa bits inside a block whose scale follows its own maximum suffice), it really is another
    arithmetic (bits differ from "h6" = 0), and the image-ready call gives the per-call bits."""
    B, rows = 2, 384
    rs = _rs(K + Nout)
    x, W, b = _t(rs.randn(B, rows, K) * np.exp(rs.uniform(-3, 3, size=(B, rows, 1))).astype(np.float32)), _t(rs.randn(Nout, K) / math.sqrt(K)), \
        _t(rs.randn(Nout) / math.sqrt(K))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    ref = F.linear(x.double() * pa[:, None].double() + po[:, None].double(), W.double(), b.double())
    out, err = {}, {}
    try:
        for on in (0, 1):
            ops.set_option("h6", on)
            img = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), kind=2)
            out[on] = ops.decode_h8_image(img).cpu()
            err[on] = cpu_ref.rel_err(out[on].double(), ref)
        ws = torch.empty(Nout * K * 4, dtype=torch.uint8, device="cuda")
        a = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), kind=2, wsplit=ws)
        c = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W.cuda(), b.cuda(), kind=2, wsplit=ws, image_ready=True)
        assert torch.equal(a, c) and torch.equal(a, img)
    finally:
        ops.set_option("h6", -1)
    print(f"K={K}: h8 {err[0][0]:.2e} / {err[0][1]:.2e}, h6 {err[1][0]:.2e} / {err[1][1]:.2e} (max-rel / rel-L2)")
    assert err[1][0] < 3e-5 and err[1][1] < 1.3 * err[0][1] + 2e-6, err
    assert not torch.equal(out[0], out[1])


@pytest.mark.parametrize("B,T,Cc,G", [(3, 128, 384, 32), (1, 64, 512, 32), (8, 128, 256, 32), (2, 72, 128, 16)])
def test_adagn_coeffs_channel_parts_on_few_samples_with_many_tiles(ops, B, T, Cc, G):
    """AdaGN coefficients from the per-tile partial sums (models/normalization.py:36-44) where few samples carry many row tiles (a cached
    `upsample` evaluation: 8 clouds x 128 tiles): `adagn_coeffs_launch` cuts the channels into whole-group parts across blocks.  Against
    float64 from the same partials, and against the one-block-per-sample form (the same samples inside a batch of 64: its grid alone fills
    the chip) to fp32 rounding of the double sums."""
    rs = _rs(B + T + Cc)
    rows = 128 * T
    part = rs.randn(B, T, 128, Cc) * np.exp(rs.uniform(-1, 1, size=(B, 1, 1, Cc))) + rs.randn(B, 1, 1, Cc)
    st = _t(np.stack([part.sum(2), (part ** 2).sum(2)], axis=2))                     # (B, T, 2, C): {sum x, sum x^2} per tile
    t = _t(rs.randn(B, 1))
    sw, sb, bw, bb = _t(rs.randn(Cc, 1) * 0.3), _t(1 + 0.1 * rs.randn(Cc)), _t(rs.randn(Cc, 1) * 0.3), _t(0.1 * rs.randn(Cc))
    a, o = ops.adagn_coeffs(st.cuda(), rows, t.cuda(), (sw.cuda(), sb.cuda(), bw.cuda(), bb.cuda()), G)
    s64 = st.double().sum(1)                                                          # (B, 2, C)
    cpg = Cc // G
    g1, g2 = s64[:, 0].reshape(B, G, cpg).sum(2), s64[:, 1].reshape(B, G, cpg).sum(2)
    n = float(rows * cpg)
    mean = g1 / n
    rstd = 1.0 / torch.sqrt((g2 / n - mean * mean).clamp_min(0) + 1e-5)
    mean_c, rstd_c = mean.repeat_interleave(cpg, 1), rstd.repeat_interleave(cpg, 1)
    s = t.double() @ sw.double().t() + sb.double()
    z = t.double() @ bw.double().t() + bb.double()
    assert cpu_ref.rel_err(a.cpu().double(), s * rstd_c)[0] < 2e-6 and cpu_ref.rel_err(o.cpu().double(), z - s * mean_c * rstd_c)[0] < 2e-6
    # the same samples as the first of a batch of 64: one block per sample
    rep = lambda v: torch.cat([v] + [v[:1]] * (64 - B)).cuda()   # noqa: E731
    a2, o2 = ops.adagn_coeffs(rep(st), rows, rep(t), (sw.cuda(), sb.cuda(), bw.cuda(), bb.cuda()), G)
    assert cpu_ref.rel_err(a2[:B].cpu(), a.cpu())[0] < 1e-6 and cpu_ref.rel_err(o2[:B].cpu(), o.cpu())[0] < 1e-6


def test_h6_block_scales_on_degenerate_blocks(ops):
    """The fp6 cross terms' block scales on blocks a trained or pruned network may hold: all-zero weight blocks and rows, all-zero
    activation rows, one element 1e8 times its block's others, magnitudes down at 1e-30 — finite everywhere, the zero blocks exactly
    neutral (rows of zero weights give the bias, zero activations give the bias), the rest within the h8 bar."""
    B, rows, K, Nout = 1, 256, 384, 256
    rs = _rs(77)
    x = _t(rs.randn(B, rows, K))
    x[0, 10:20] = 0.0                       # all-zero activation rows
    x[0, 30, :] = 1e-30                     # tiny magnitudes
    x[0, 40, 5] = 300.0                     # one dominant element in its block
    W = _t(rs.randn(Nout, K) / math.sqrt(K))
    W[7] = 0.0                              # an all-zero weight row
    W[:, 64:128] = 0.0                      # all-zero 64-k groups in every row
    W[20, 200] = 50.0
    W[21, 201:232] *= 1e-8
    b = _t(rs.randn(Nout) * 0.1)
    ref = F.linear(x.double(), W.double(), b.double())
    try:
        ops.set_option("h6", 1)
        img = ops.linear_h8_img(x.cuda(), None, W.cuda(), b.cuda(), kind=2)
    finally:
        ops.set_option("h6", -1)
    got = ops.decode_h8_image(img).cpu().double()
    assert torch.isfinite(got).all()
    tol = 2e-5 * float(b.abs().max())                                   # the h8 image's own rounding of the stored value (fp16 + fp8 lo)
    assert float((got[0, 10:20] - b.double()).abs().max()) <= tol      # zero rows: the bias
    assert float((got[0, :, 7] - b[7].double()).abs().max()) <= tol   # zero weight row
    e = cpu_ref.rel_err(got, ref)
    assert e[0] < 3e-5, e


@pytest.mark.parametrize("B,rows,K,hd", [(2, 256, 384, 48), (1, 128, 128, 16), (2, 384, 256, 32), (1, 256, 512, 64), (2, 256, 384, 0)])
def test_linear_kvq_f16(ops, B, rows, K, hd):
    """kv_proj | q_proj of the mixed mode on the 64-column-tile kernel (gemm_kvq_astat_kernel): fp16(y) x fp16(W) everywhere, the V
    columns with the fp8 second weight term; head-major and row-major outputs; against float64 on the SAME rounded operands
    (so the bar is fp32 accumulation + the fp16 rounding of the result), and the V columns measurably closer to the
    unrounded-weight product than one-term weights are (models/set_transformer.py:49-52, 65-70)."""
    rs = _rs(B + rows + K + hd)
    C = K
    x = _t(rs.randn(B, rows, K))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    Wkv, Wq, bq = _t(rs.randn(2 * C, K) / math.sqrt(K)), _t(rs.randn(C, K) / math.sqrt(K)), _t(rs.randn(C) / math.sqrt(K))
    y16 = (x * pa[:, None] + po[:, None]).half().double()          # the kernel's A operand: one fma, one rounding
    kv, q = ops.linear_kvq_f16(x.cuda(), (pa.cuda(), po.cuda()), Wkv.cuda(), None, Wq.cuda(), bq.cuda(), lo=(C, 2 * C), head_dim=hd)
    if hd == 48:
        # head dim 48: the stream deals columns to the tiles head-aligned and a head's (32 rows, 48) slab leaves as three contiguous
        # 1 KiB stores (option "kvqperm", default on) — a column's dot product does not depend on its place in a tile: same bits
        try:
            ops.set_option("kvqperm", 0)
            kv0, q0 = ops.linear_kvq_f16(x.cuda(), (pa.cuda(), po.cuda()), Wkv.cuda(), None, Wq.cuda(), bq.cuda(), lo=(C, 2 * C), head_dim=hd)
        finally:
            ops.set_option("kvqperm", -1)
        assert torch.equal(kv, kv0) and torch.equal(q, q0)
    if hd:
        kv = kv.permute(0, 2, 1, 3).reshape(B, rows, 2 * C)        # "b g n d -> b n (g d)"
        q = q.permute(0, 2, 1, 3).reshape(B, rows, C)
    kv, q = kv.cpu().double(), q.cpu().double()
    Wh = Wkv.half().double()
    ref_k = y16 @ Wh[:C].T
    ref_q = y16 @ Wq.half().double().T + bq.double()
    ref_v2 = y16 @ Wkv[C:].double().T                              # two-term weights ~ the unrounded weights
    ref_v1 = y16 @ Wh[C:].T
    tol = 2.0 ** -10                                               # fp16 result rounding (relative to the value) + accumulation
    assert cpu_ref.rel_err(kv[..., :C], ref_k)[0] < tol
    assert cpu_ref.rel_err(q, ref_q)[0] < tol
    # compare before the output rounding hides it: mean absolute distance to the two candidates
    d2, d1 = (kv[..., C:] - ref_v2).abs().mean(), (kv[..., C:] - ref_v1).abs().mean()
    assert cpu_ref.rel_err(kv[..., C:], ref_v2)[0] < tol and d2 <= d1, (d2, d1)
    # one segment only (the cached evaluation's q projection), image-ready second call: same bits
    ws = torch.empty(C * K * 2, dtype=torch.uint8, device="cuda")
    a = ops.linear_kvq_f16(x.cuda(), (pa.cuda(), po.cuda()), Wq.cuda(), bq.cuda(), head_dim=hd, wsplit=ws)
    c = ops.linear_kvq_f16(x.cuda(), (pa.cuda(), po.cuda()), Wq.cuda(), bq.cuda(), head_dim=hd, wsplit=ws, image_ready=True)
    qq = a.permute(0, 2, 1, 3).reshape(B, rows, C) if hd else a
    assert torch.equal(a, c) and cpu_ref.rel_err(qq.cpu().double(), ref_q)[0] < tol


@pytest.mark.parametrize("B,rows,K,Wd", [(2, 256, 384, 768), (1, 128, 128, 256), (2, 384, 256, 512), (2, 256, 512, 1024)])
def test_linear_h8_areg_chain(ops, B, rows, K, Wd):
    """The point MLP of the mixed mode as its two h8 launches (models/set_transformer.py:164-166, models/mlp.py): mlp.0 writes the
    h8 activation image (fp16 hi + fp8 lo), mlp.2 (gemm_h8_areg.hip) loads it into registers.  The image against float64 of
    mlp.0; mlp.2 against float64 on the image's own values (its product alone) and the whole chain against float64."""
    rs = _rs(B + rows + K)
    x, W0, b0 = _t(rs.randn(B, rows, K)), _t(rs.randn(Wd, K) / math.sqrt(K)), _t(rs.randn(Wd) / math.sqrt(K))
    W2, b2 = _t(rs.randn(K, Wd) / math.sqrt(Wd)), _t(rs.randn(K) / math.sqrt(Wd))
    pa, po = _t(1 + 0.3 * rs.randn(B, K)), _t(0.3 * rs.randn(B, K))
    alpha = _t(np.array(0.9))
    u = F.linear((x.double() * pa[:, None].double() + po[:, None].double()), W0.double(), b0.double())
    hid = (torch.exp(-u * u / (2 * 0.9 ** 2)) - 0.7) / 0.28
    img = ops.linear_h8_img(x.cuda(), (pa.cuda(), po.cuda()), W0.cuda(), b0.cuda(), act_alpha=alpha.cuda(), kind=2)
    dec = ops.decode_h8_image(img).cpu()
    e = cpu_ref.rel_err(dec, hid)
    assert e[0] < 1e-4, e                                        # fast exp in the epilogue, as for the split image
    out, stats = ops.linear_h8_areg(img, W2.cuda(), b2.cuda(), residual=x.cuda(), want_stats=True)
    ref2 = x.double() + F.linear(dec, W2.double(), b2.double())  # the consumer's own product on exactly the operand it read
    e2 = cpu_ref.rel_err(out.cpu().double(), ref2)
    assert e2[0] < 1e-5, e2
    ref = x.double() + F.linear(hid, W2.double(), b2.double())
    e3 = cpu_ref.rel_err(out.cpu().double(), ref)
    assert e3[0] < 5e-5, e3
    s = stats.cpu().double().sum(1)
    _close(s[:, 0], ref2.sum(1), 1e-4)
    _close(s[:, 1], (ref2 ** 2).sum(1), 1e-5)
    # in place on the residual stream, no bias, image-ready call
    ws = torch.empty(K * Wd * 4, dtype=torch.uint8, device="cuda")
    xc = x.cuda().clone()
    ops.linear_h8_areg(img, W2.cuda(), None, residual=xc, out=xc, wsplit=ws)
    xd = x.cuda().clone()
    ops.linear_h8_areg(img, W2.cuda(), None, residual=xd, out=xd, wsplit=ws, image_ready=True)
    assert torch.equal(xc, xd)
    assert cpu_ref.rel_err(xc.cpu().double(), x.double() + F.linear(dec, W2.double()))[0] < 1e-5
