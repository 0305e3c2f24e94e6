"""The layer above the C-ABI keeps no process state that a forward mutates:

  * the vendor-GEMM rows-per-call choice is scoped to one model forward (ops.gemm_rows_per_call) and recorded on the autograd nodes, so a process that
    holds a CogVideoX and a Wan model (a scorer next to a trainer, two trainers) runs each exactly as a fresh process would;
  * "precise delta" is a per-model attribute handed down per call;
  * vendor GEMMs run past an operand's logical row count ONLY over buffers ops._empty_rows made (tagged), and never write outside them: every slack region
    is filled with NaN, a canary region sits behind it, a cfg2-shaped block runs forward + backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _cog_step(pm, x, txt, t, dy):
    for p in pm.parameters():
        p.grad = None
    y = pm(x, encoder_hidden_states=txt, timestep=t).sample
    y.backward(dy)
    return y.detach().clone(), {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}


def _wan_step(pm, x, t, ctx, L, gout):
    for p in pm.parameters():
        p.grad = None
    out = pm(x, t=t, context=ctx, seq_len=L)
    sum((o * g).sum() for o, g in zip(out, gout)).backward()
    return [o.detach().clone() for o in out], {n: p.grad.clone() for n, p in pm.named_parameters() if p.grad is not None}


def test_two_model_families_in_one_process_do_not_see_each_other():
    """CogVideoX step, Wan step (two samples: its GEMMs run one call per sample), CogVideoX again, Wan again -- inside one process, the Wan model under block
    recompute the second time is NOT what is compared (same settings both times): every output and every adapter gradient of a repeat is bit-identical to
    the first run, and outside a forward the scoped setting reads 0."""
    import test_gpu_model as tm
    import test_gpu_wan_model as tw
    from videogpa_amd import ops
    cfg, _, _, cog = tm._setup(b_std=0.05)
    x, txt, t = tm._inputs(cfg, B=2, seed=21)
    dy = torch.randn(x.shape, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
    cog.train()
    wan, _, _ = tw._build()
    wx, wt, wctx, L, gout = tw._inputs()
    seen = []
    orig = ops._linear_rows

    def spy(x2, W, bias, ext=False, split=0):
        seen.append((x2.shape[0], split))
        return orig(x2, W, bias, ext, split)
    ops._linear_rows = spy
    try:
        a1 = _cog_step(cog, x.cuda(), txt.cuda(), t.cuda(), dy)
        n_cog = len(seen)
        assert all(sp == 0 for _, sp in seen)
        w1 = _wan_step(wan, wx, wt, wctx, L, gout)
        wan_calls = seen[n_cog:]
        assert any(sp == L and M == 2 * L for M, sp in wan_calls), wan_calls          # forward AND backward GEMMs of the two-sample batch ran per sample
        assert sum(sp == L and M == 2 * L for M, sp in wan_calls) >= 8
        assert ops.current_gemm_rows() == 0                                           # nothing left behind
        n2 = len(seen)
        a2 = _cog_step(cog, x.cuda(), txt.cuda(), t.cuda(), dy)
        assert all(sp == 0 for _, sp in seen[n2:])                                   # the CogVideoX model never sees the Wan model's choice
        w2 = _wan_step(wan, wx, wt, wctx, L, gout)
    finally:
        ops._linear_rows = orig
    assert torch.equal(a1[0], a2[0]) and all(torch.equal(a, b) for a, b in zip(w1[0], w2[0]))
    for first, again in ((a1[1], a2[1]), (w1[1], w2[1])):
        assert first.keys() == again.keys() and len(first) > 0
        for n in first:
            assert torch.equal(first[n], again[n]), n


def test_precise_delta_is_a_per_model_setting():
    """two CogVideoX models in one process, one with the textbook backward: each keeps what it was told (the residual tensor exists only for the first)"""
    import test_gpu_model as tm
    from videogpa_amd import ops
    cfg, _, _, a = tm._setup(b_std=0.05)
    _, _, _, b = tm._setup(b_std=0.05)
    b.get_base_model().set_precise_delta(None)
    x, txt, t = tm._inputs(cfg, B=1, seed=5)
    kinds = []
    orig = ops.attention_bwd_raw

    def spy(*args, **kw):
        kinds.append(None if kw.get("o_res") is None else kw["o_res"].dtype)
        return orig(*args, **kw)
    ops.attention_bwd_raw = spy
    try:
        for m in (a, b, a):
            m.train()
            m(x.cuda(), encoder_hidden_states=txt.cuda(), timestep=t.cuda()).sample.float().square().mean().backward()
    finally:
        ops.attention_bwd_raw = orig
    L = cfg.num_layers
    assert kinds == [torch.uint8] * L + [None] * L + [torch.uint8] * L, kinds


def test_gemm_row_slack_is_only_ever_the_buffers_made_for_it():
    """One CogVideoX-5B-width block (D = 3072, 48 heads, LoRA r = 64) at cfg2's row count (2 x 17 776 = 35 552 rows -> the vendor GEMMs cover 35 840 /
    36 864), forward + backward, with ops._empty_rows replaced by a version that (i) fills every slack row with NaN and (ii) puts 8 canary rows behind the
    slack: outputs and gradients finite, canaries untouched.  And a caller's own view of a wider tensor is never run past its shape."""
    from videogpa_amd import ops
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.transformer import COGVIDEOX_5B, CogVideoXTransformer3DModel
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = CogVideoXTransformer3DModel(**dict(COGVIDEOX_5B, num_layers=1))
    finally:
        torch.set_default_dtype(prev)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02)
    pm = get_peft_model(model, LoraConfig(r=64, lora_alpha=128, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    with torch.no_grad():
        for n, p in pm.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 1e-2)
    pm.train()
    made, CANARY = [], 0x3C00
    orig = ops._empty_rows

    def poisoned(shape, dtype, device, wide=False):
        shape = tuple(int(v) for v in shape)
        rows = 1
        for v in shape[:-1]:
            rows *= v
        Mp = ops.gemm_rows(rows, wide=wide)
        if Mp == rows or dtype != torch.bfloat16:
            return orig(shape, dtype, device, wide)
        w = shape[-1]
        t = torch.empty((Mp + 8) * w, dtype=dtype, device=device)          # ONE storage: logical rows | slack rows | 8 canary rows
        t[rows * w: Mp * w] = float("nan")
        t[Mp * w:].view(torch.int16).fill_(CANARY)
        full = t.new_empty(0).set_(t.untyped_storage(), 0, ((Mp + 8) * w,), (1,))      # a second handle on the whole storage, for the check below
        t = t.resize_(shape)                              # like the real _empty_rows: the result is its own base, with more storage than its shape
        t._vgpa_rows = (rows, Mp, w)
        made.append((full, rows, Mp, w))
        return t
    ops._empty_rows = poisoned
    calls = []
    lin = ops._linear_rows

    def spy(x2, W, bias, ext=False, split=0):
        calls.append((x2.shape[0], ops._slack_rows(x2)))
        return lin(x2, W, bias, ext, split)
    ops._linear_rows = spy
    try:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (0.7 * torch.randn(2, 13, 16, 60, 90, generator=g, device="cuda")).bfloat16()
        txt = (0.2 * torch.randn(2, 226, 4096, generator=g, device="cuda")).bfloat16()
        y = pm(x, encoder_hidden_states=txt, timestep=torch.tensor([400, 400], device="cuda")).sample
        y.float().square().mean().backward()
        torch.cuda.synchronize()
    finally:
        ops._empty_rows, ops._linear_rows = orig, lin
    assert torch.isfinite(y).all()
    grads = [p.grad for n, p in pm.named_parameters() if p.requires_grad]
    assert len(grads) == 8 and all(gr is not None and torch.isfinite(gr).all() for gr in grads) and all(float(gr.abs().max()) > 0 for gr in grads)
    assert len(made) >= 6 and sum(room > M for M, room in calls) >= 6, (len(made), calls)          # the padded path really ran, forward and backward
    for full, rows, Mp, w in made:
        assert (full[Mp * w:].view(torch.int16) == CANARY).all(), (rows, Mp, w)
    # a view of somebody else's buffer is not a slack buffer, whatever room its storage has
    big = torch.zeros(40000, 3072 + 64, dtype=torch.bfloat16, device="cuda")
    assert ops._slack_rows(big[:35552, :3072]) == 35552 and ops._slack_rows(big[:35552]) == 35552
    own = ops._empty_rows((35552, 3072), torch.bfloat16, "cuda")
    assert ops._slack_rows(own) == 35840 and ops._slack_rows(own[:1000]) == 1000 and ops._slack_rows(own.view(2, 17776, 3072).view(-1, 3072)) == 35840
