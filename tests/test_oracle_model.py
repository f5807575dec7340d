"""CPU checks of the oracle's model-level bookkeeping that no kernel-level test sees: the ring KV state of causal sliding-window
attention layers (restated from mixer/attention/state.rs:16-106, 174-236 + mode.rs:66-78 + core/single_pass.rs:44-73)."""
import numpy as np

from oracle import oracle as O
from uzu_amd import synthetic as S


def run(cfg, prompt, steps):
    om = O.OracleModel(S.build_model(cfg))
    tok, lg = om.prefill(prompt, True)
    toks, logits = [tok], [lg]
    for _ in range(steps):
        tok, lg = om.forward([tok], True)
        toks.append(tok)
        logits.append(lg)
    om.close()
    return toks, logits


def test_ring_state_that_never_wraps_equals_the_full_cache_bit_for_bit():
    """A window at least as long as the whole sequence: the sliding-window mask never removes a key and the ring never wraps, but every
    new row still takes the ring path (suffix region at row `window`, kv_token_offset = window, ring_length = accepted tokens,
    encode_accept copying suffix rows into ring slots) and the keys are visited in the same logical order -- so logits must be
    BIT-identical to the Full-cache model.  Pins the ring bookkeeping (physical prefix, suffix position, copy destinations)."""
    prompt = S.synthetic_prompt(130, 1024)
    full = run(S.tiny_llama(max_context_length=400, seed=47), prompt, 6)
    ring = run(S.tiny_llama(sliding_windows=[300], max_context_length=400, seed=47), prompt, 6)
    assert full[0] == ring[0]
    for a, b in zip(full[1], ring[1]):
        assert np.array_equal(a, b)


def test_wrapped_ring_sees_exactly_the_window():
    """window 48 < prompt 130: the ring wraps inside the prefill chunk (only the last 48 suffix rows survive encode_accept) and the
    decode steps see ring_offset != 0.  Cross-check against the SAME model run with the prompt fed in two ways that must agree: one
    130-token chunk, and 13 chunks of 10 tokens (different suffix lengths => different copy lists and ring offsets per pass; the
    visible key set per query is the same 48-token window, summed in a different physical order: logits agree to bf16 noise, tokens
    identical) -- and differ from the full-attention model."""
    cfg = S.tiny_llama(sliding_windows=[48], max_context_length=400, seed=47)
    prompt = S.synthetic_prompt(130, cfg.vocab_size)
    one = run(cfg, prompt, 4)
    om = O.OracleModel(S.build_model(cfg))
    for i in range(0, 120, 10):
        om.forward(prompt[i:i + 10])
    tok, lg = om.forward(prompt[120:130], True)
    toks, logits = [tok], [lg]
    for _ in range(4):
        tok, lg = om.forward([tok], True)
        toks.append(tok)
        logits.append(lg)
    om.close()
    f = lambda b: (np.asarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    for a, b in zip(one[1], logits):
        assert np.abs(f(a) - f(b)).max() <= 0.25 * f(a).std()  # the model-level logit tolerance of the GPU tests (a few bf16 ulps after three bf16 layers; measured 0.075-0.133)
    assert one[0] == toks
    full = run(S.tiny_llama(max_context_length=400, seed=47), prompt, 4)
    diff = np.abs(f(full[1][0]) - f(one[1][0])).max() / f(full[1][0]).std()
    assert diff > 0.4, f"a 48-token window over a 130-token prompt must change the logits (max difference {diff:.3f} sigma)"
