"""The forward attention against the DATA: the w1 kernel shifts every row's scores by M' = min(b, m_s + 64) -- b = |q| max|k|, m_s = the row's
maximum over 64 keys spread over the sequence -- instead of carrying a running maximum; strips with a row it cannot represent (true maximum > ~176 log2 units above
the sampled one, or |M'| > 1024) are flagged and redone by the online-softmax kernel in the same call; a layer with more than half its strips flagged switches to the
all-online entry point for good (ops.AttnFwdPolicy).  Operands: tools/attn_data.py::trained_like_qkv (QK-norm gains, outlier channels, one matched key per query,
sink keys).  Checked: results against fp64 whatever path runs (weights far above 1 in the shifted loop included), the flag accounting, the switch, and -- at the
headline shape -- the TIME against the same launch on bench.py's N(0,1) operands, with the cliff that remains stated as numbers."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_tol import lse2_tol  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from videogpa_amd import ops as o
    return o


def _ref(q, k, v):
    c = q.shape[-1] ** -0.5
    qs = (q.float() * (c * 1.4426950408889634)).to(torch.bfloat16).double()            # the kernel's contract: q arrives pre-scaled and rounded
    s2 = qs @ k.double().transpose(-1, -2)
    p = torch.exp2(s2 - s2.max(-1, keepdim=True).values)
    w = p / p.sum(-1, keepdim=True)
    lse_ref = s2.max(-1).values + torch.log2(p.sum(-1))
    return w @ v.double(), lse_ref, lse2_tol(w, lse_ref)


@pytest.mark.parametrize("kw,expect_redo", [(dict(gain=1.0), "none"), (dict(gain=2.0), "none"), (dict(gain=2.5), "any"), (dict(gain=6.0), "all"),
                                            (dict(gain=1.0, sink_norm=10.0), "any")])
def test_forward_is_exact_on_trained_like_data_whatever_path_runs(ops, kw, expect_redo):
    from attn_data import trained_like_qkv
    B, H, S = 1, 3, 2304
    q, k, v, stats = trained_like_qkv(B, H, S, seed=3, **kw)
    pol = ops.AttnFwdPolicy()
    o, lse = ops.attention_fwd_raw(q, k, v, policy=pol)
    o_ref, lse_ref, lse_tol = _ref(q, k, v)
    got = o.view(B, S, H, 64).permute(0, 2, 1, 3).double()
    assert torch.isfinite(got).all() and torch.isfinite(lse).all()
    err = (got - o_ref).abs()
    assert bool((err <= 0.02 + 0.008 * o_ref.abs()).all()), (kw, float(err.max()), stats)
    # lse2 is log2 of the sum of the bf16-rounded weights the PV product uses (tests/attn_tol.py: <= 5.7e-3 on a one-hot row, ~ 2e-3 / sqrt(n) over n keys)
    assert bool(((lse.double() - lse_ref).abs() <= lse_tol).all()), (kw, float(((lse.double() - lse_ref).abs() / lse_tol).max()), float((lse.double() - lse_ref).abs().max()))
    f = pol.redo_fraction
    assert f is not None and 0.0 <= f <= 1.0
    if expect_redo == "none":
        assert f == 0.0 and pol.mode == "bound", (kw, f, stats)
    if expect_redo == "all":
        assert f == 1.0 and pol.mode == "online", (kw, f, stats)
        # the all-online entry point gives the same bits as the redo pass did
        o2, lse2 = ops.attention_fwd_raw(q, k, v, policy=pol)
        assert torch.equal(o2, o) and torch.equal(lse2, lse)
        assert pol.calls == 2 and pol.switched_at == 0


def test_forward_time_against_the_data_at_the_headline_shape(ops):
    """B = 2, H = 48, S = 17 776: the launch bench.py times, on operands shaped like a trained QK-normed model's (tools/attn_data.py).  Asserted:
      * down to a row entropy of ~0.6 bits (QK-norm gain 3 with 3 x outlier channels: scores spread over +-230 log2 units, every query matched to one key) and with
        10 x sink keys at gain 1, (next to) NO strip is redone (<= 0.2 %; measured 0 - 1 of 6 720) and the launch takes <= 1.25 x its time on bench.py's N(0,1) operands (measured 0.97-1.11 x over four boxes; the cliff this guards
        against is 2.5 x, and three launches per arm at the power cap scatter by several per cent);
      * beyond that (gain 4: 0.3 bits; gain 2 with 10 x sinks) strips get flagged -- the documented cliff: what the policy runs is the cheaper of the two forms
        (within 20 %) and stays under 3 x (measured 1.4-1.9 x at gain 4, 2.42-2.50 x on the all-online entry) (round 5's kernel was there from gain 2.5 on: profiles/r06a_attn_trained_like.txt)."""
    from attn_data import trained_like_qkv
    B, H, S = 2, 48, 17776

    def ms(q, k, v, mode, n=3):
        pol = ops.AttnFwdPolicy(mode=mode, fixed=True)
        ops.attention_fwd_raw(q, k, v, policy=pol)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.attention_fwd_raw(q, k, v, policy=pol)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, S, 64, generator=g, device="cuda").to(torch.bfloat16) for _ in range(3))
    base = ms(q, k, v, "bound")
    del q, k, v
    report = {"randn_bound_ms": base}
    fails = []
    for name, kw, fast in (("gain1", dict(gain=1.0), True), ("gain2", dict(gain=2.0), True), ("gain3", dict(gain=3.0), True), ("gain1_sink10", dict(gain=1.0, sink_norm=10.0), True),
                           ("gain4", dict(gain=4.0), False), ("gain2_sink10", dict(gain=2.0, sink_norm=10.0), False)):
        q, k, v, stats = trained_like_qkv(B, H, S, **kw)
        pol = ops.AttnFwdPolicy()
        ops.attention_fwd_raw(q, k, v, policy=pol)
        t = ms(q, k, v, pol.mode)
        rec = {"redo_fraction": pol.redo_fraction, "mode": pol.mode, "ms": t, "over_randn": t / base, "row_entropy_bits": stats["row_entropy_bits_mean"],
               "gap_log2": stats["gap_bound_minus_rowmax_mean"]}
        if fast:
            if not (pol.redo_fraction <= 0.002 and pol.mode == "bound" and t <= 1.25 * base):      # sinks: 1 strip of 6 720 flagged on one box (the sample missed a row's sink)
                fails.append((name, rec))
        else:
            other = ms(q, k, v, "online" if pol.mode == "bound" else "bound")
            rec["other_form_ms"] = other
            if not (t <= 1.20 * other and t <= 3.0 * base):
                fails.append((name, rec))
        report[name] = rec
        del q, k, v
    print(report)
    assert not fails, fails
