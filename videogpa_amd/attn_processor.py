"""The narrowest drop-in seam when the REAL diffusers package is present ("attention operator hook"):
diffusers' `Attention.set_processor(P)` with `P.__call__(attn, hidden_states, encoder_hidden_states, attention_mask=None,
image_rotary_emb=None) -> (hidden_states, encoder_hidden_states)` -- the protocol of `CogVideoXAttnProcessor2_0`, which is what
`CogVideoXTransformer3DModel` (train/CogVideoX-5B/03_train.py:101,134-151) calls inside every block.

`MI355XCogVideoXAttnProcessor` keeps diffusers' module tree, weights and the PEFT wrappers a `get_peft_model` call put on
`to_q / to_k / to_v / to_out.0`, and replaces only the arithmetic of the attention layer: one fused-QKV hipBLASLt GEMM with the
LoRA adapters riding as extra K, QK-norm (+ 3D RoPE on the video tokens when `image_rotary_emb` is given) and the full 3D
attention as the hand-written gfx950 kernels, output projection.  Text tokens come first in the concatenated sequence, exactly
as the upstream processor builds it.

    from videogpa_amd.attn_processor import install
    install(transformer)            # every block: blk.attn1.set_processor(MI355XCogVideoXAttnProcessor())

It works on any module that carries those attributes (duck-typed); tests install it on this package's own blocks.
"""
import torch

from .transformer import AttentionCore


class MI355XCogVideoXAttnProcessor:
    def __init__(self):
        self._cores = {}

    def _core(self, attn):
        c = self._cores.get(id(attn))
        if c is None or c.mod is not attn:
            c = AttentionCore(attn)
            self._cores[id(attn)] = c
        return c

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("CogVideoX runs unmasked full attention; attention_mask is not supported by the MI355X kernels")
        if not hidden_states.is_cuda or hidden_states.dtype != torch.bfloat16:
            raise RuntimeError("MI355XCogVideoXAttnProcessor needs bf16 tensors on the GPU (no CPU fallback)")
        text_len = encoder_hidden_states.size(1)
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1).contiguous()
        rope = None
        if image_rotary_emb is not None:
            rope = (image_rotary_emb[0].float().contiguous(), image_rotary_emb[1].float().contiguous())
        out = self._core(attn).forward(x, text_len, rope)
        drop = attn.to_out[1] if len(attn.to_out) > 1 else None
        if drop is not None:
            out = drop(out)
        enc, hid = out.split([text_len, out.size(1) - text_len], dim=1)
        return hid, enc


def install(transformer):
    """Put the MI355X processor on every `attn1` of a CogVideoX transformer (diffusers' or this package's)."""
    n = 0
    for blk in transformer.transformer_blocks:
        blk.attn1.set_processor(MI355XCogVideoXAttnProcessor())
        n += 1
    return n
