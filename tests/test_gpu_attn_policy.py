"""The forward attention against the DATA: the w1 kernel shifts scores by M' = b - min(60, b / 2), b = |q| max|k| (no running maximum); strips
whose rows it cannot represent are flagged and redone by the online-softmax kernel in the same call; a layer whose flagged fraction exceeds 5 % switches to the
all-online entry point for good (ops.AttnFwdPolicy).  Operands: tools/attn_data.py::trained_like_qkv (QK-norm gains, outlier channels, one matched key per query,
sink keys).  Checked: results against fp64 whatever path runs (weights above 1 in the bound-shifted loop included), the flag accounting, the switch, and -- at the
headline shape -- the TIME: what the policy ends up running is within 10 % of the same launch on bench.py's N(0,1) operands for every distribution tried."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


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
    return (p @ v.double()) / p.sum(-1, keepdim=True), (s2.max(-1).values + torch.log2(p.sum(-1)))


@pytest.mark.parametrize("kw,expect_redo", [(dict(gain=1.0), "none"), (dict(gain=2.0), "none"), (dict(gain=2.5), "any"), (dict(gain=6.0), "all"),
                                            (dict(gain=1.0, sink_norm=10.0), "any")])
def test_forward_is_exact_on_trained_like_data_whatever_path_runs(ops, kw, expect_redo):
    from attn_data import trained_like_qkv
    B, H, S = 1, 3, 2304
    q, k, v, stats = trained_like_qkv(B, H, S, seed=3, **kw)
    pol = ops.AttnFwdPolicy()
    o, lse = ops.attention_fwd_raw(q, k, v, policy=pol)
    o_ref, lse_ref = _ref(q, k, v)
    got = o.view(B, S, H, 64).permute(0, 2, 1, 3).double()
    assert torch.isfinite(got).all() and torch.isfinite(lse).all()
    err = (got - o_ref).abs()
    assert bool((err <= 0.02 + 0.008 * o_ref.abs()).all()), (kw, float(err.max()), stats)
    assert bool(((lse.double() - lse_ref).abs() <= 2e-3 + 2e-5 * lse_ref.abs()).all()), (kw, float((lse.double() - lse_ref).abs().max()))
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


def test_policy_keeps_the_forward_within_ten_percent_of_its_randn_time_at_the_headline_shape(ops):
    """B = 2, H = 48, S = 17 776: the launch bench.py times.  For each distribution: the mode the policy picks after its first call, timed; asserted <= 1.10 x the
    bound-shifted kernel on N(0,1) operands.  The cliff the policy removes is reported too: the bound-shifted call when every strip is redone (two sweeps)."""
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
    for name, kw in (("gain1", dict(gain=1.0)), ("gain2", dict(gain=2.0)), ("gain3", dict(gain=3.0)), ("gain4", dict(gain=4.0)), ("gain2_sink10", dict(gain=2.0, sink_norm=10.0))):
        q, k, v, stats = trained_like_qkv(B, H, S, **kw)
        pol = ops.AttnFwdPolicy()
        ops.attention_fwd_raw(q, k, v, policy=pol)
        t = ms(q, k, v, pol.mode)
        report[name] = {"redo_fraction": pol.redo_fraction, "mode": pol.mode, "ms": t, "over_randn": t / base, "row_entropy_bits": stats["row_entropy_bits_mean"],
                        "gap_log2": stats["gap_bound_minus_rowmax_mean"]}
        if pol.mode == "online":
            report[name]["bound_with_redo_ms"] = ms(q, k, v, "bound")
        assert t <= 1.10 * base, (name, t, base, report)
        del q, k, v
    print(report)
