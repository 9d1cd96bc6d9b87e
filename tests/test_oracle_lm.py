"""The oracle's minimal LLaMA/Mistral forward against the installed transformers implementation (CPU fp32):
prefill logits, cached decode with a 2-D key mask (zeros on visual-token columns), GQA."""
import numpy as np
import pytest
import torch

from oracle.lm_ref import KVCache, LMConfig, lm_hidden, lm_logits, random_weights

torch.set_grad_enabled(False)


def _hf(cfg: LMConfig, w, kind):
    from transformers import LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM
    kw = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
              num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
              head_dim=cfg.head_dim, max_position_embeddings=512, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
              tie_word_embeddings=False)
    if kind == "llama":
        m = LlamaForCausalLM(LlamaConfig(attention_bias=False, mlp_bias=False, **kw))
    else:
        m = MistralForCausalLM(MistralConfig(sliding_window=None, **kw))
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected
    return m.eval()


@pytest.mark.parametrize("kind,heads,kv", [("llama", 2, 2), ("mistral", 4, 2)])
def test_lm_ref_matches_transformers(kind, heads, kv):
    cfg = LMConfig(300, 128 * heads, 256, 2, heads, kv, 128, 1e-5, 10000.0 if kind == "llama" else 1e6)
    w = random_weights(cfg, 3, 0.05)
    hf = _hf(cfg, w, kind)
    gen = torch.Generator().manual_seed(0)
    T0 = 12
    emb = torch.randn(T0, cfg.hidden_size, generator=gen)
    cache = KVCache()
    mine = lm_logits(cfg, w, lm_hidden(cfg, w, emb, torch.arange(T0), cache))
    out = hf(inputs_embeds=emb[None], use_cache=True)
    np.testing.assert_allclose(mine.numpy(), out.logits[0].float().numpy(), rtol=2e-4, atol=2e-5)
    # one cached decode step with zeros in the 2-D mask on "visual" columns 2..6 (what the reference feeds, llava.py:350-359)
    x = torch.randn(1, cfg.hidden_size, generator=gen)
    km = torch.ones(T0 + 1, dtype=torch.long)
    km[2:7] = 0
    mine2 = lm_logits(cfg, w, lm_hidden(cfg, w, x, torch.tensor([T0]), cache, km))
    out2 = hf(inputs_embeds=x[None], attention_mask=km[None], past_key_values=out.past_key_values, use_cache=True,
              position_ids=torch.tensor([[T0]]))
    np.testing.assert_allclose(mine2.numpy(), out2.logits[0].float().numpy(), rtol=2e-4, atol=2e-5)
    assert cache.length == T0 + 1
