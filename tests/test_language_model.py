"""Text front-end (uzu_amd/language_model.py; engine/language_model/mod.rs:57-130): tokenizer.json next to config.json and
model.safetensors, generation_config stop tokens and sampling defaults."""
import json
import os

import numpy as np
import pytest

from uzu_amd import language_model as LM
from uzu_amd import loader as L
from uzu_amd import synthetic as S

CORPUS = ["the quick brown fox jumps over the lazy dog", "uzu runs language models on the device", "int4 weights bf16 activations f32 accumulators",
          "a ring of keys and values slides over the context", "greedy decode picks the arg max and ties go to the lowest index"] * 20


def make_model_dir(tmp_path, cfg, stop_token_ids=(), **gen):
    """A model directory in the reference's layout with a BPE tokenizer.json trained offline by the `tokenizers` library."""
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers
    bundle = S.build_model(cfg)
    os.makedirs(str(tmp_path), exist_ok=True)
    d = str(tmp_path / "model")
    L.save_model_dir(bundle, d)
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=min(cfg.vocab_size, 600), special_tokens=["<unk>", "<eos>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    tok.save(os.path.join(d, "tokenizer.json"))
    with open(os.path.join(d, "config.json")) as f:
        c = json.load(f)
    c["generation_config"].update(stop_token_ids=list(stop_token_ids), **gen)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(c, f)
    return d, bundle


def test_tokenizer_round_trip_and_errors(tmp_path):
    d, bundle = make_model_dir(tmp_path, S.tiny_llama(), stop_token_ids=[1], temperature=0.7, top_k=40)
    tok = LM.load_tokenizer(d)
    text = "the quick brown fox runs language models"
    ids = tok.encode(text).ids
    assert len(ids) > 3 and max(ids) < bundle.vocab_size and tok.decode(ids) == text
    g = LM.read_generation_config(d)
    assert g.stop_token_ids == [1] and g.temperature == 0.7 and g.top_k == 40 and g.top_p is None and g.stochastic
    os.remove(os.path.join(d, "tokenizer.json"))
    with pytest.raises(LM.TokenizerError, match="tokenizer.json not found"):
        LM.load_tokenizer(d)
    with open(os.path.join(d, "tokenizer.json"), "w") as f:
        f.write("{not json")
    with pytest.raises(LM.TokenizerError):
        LM.load_tokenizer(d)


@pytest.mark.gpu
def test_generate_text_matches_the_oracle_and_honours_stop_tokens(hip_ctx, tmp_path):
    """text -> ids (tokenizer.json) -> prefill + chained decode on the HIP engine -> text: the token ids are the oracle's for the same
    prompt ids; with one of the generated tokens declared a stop token the stream ends in front of it; with sampling parameters in
    generation_config the stream is stochastic and reproducible per seed."""
    from oracle import oracle as O
    cfg = S.tiny_llama(seed=45, max_context_length=256)
    d, bundle = make_model_dir(tmp_path, cfg)
    lm = LM.LanguageModel(hip_ctx, d, max_context_length=256)
    prompt = "the quick brown fox jumps over the lazy dog and the ring of keys"
    ids = lm.encode(prompt)
    res = lm.generate(prompt, max_tokens=12, chunk=5)
    om = O.OracleModel(bundle)
    tok = om.prefill(np.asarray(ids, np.uint32))
    want = [tok]
    for _ in range(11):
        tok = om.forward([tok])
        want.append(tok)
    om.close()
    assert res.token_ids == want and res.prompt_tokens == len(ids) and res.stopped_on is None
    assert res.text == lm.decode(want) and res.prefill_tokens_per_s > 0 and res.decode_tokens_per_s > 0
    lm.close()
    # stop token = the 5th generated token: four tokens come out
    stop = want[4]
    first_hit = want.index(stop)
    d2, _ = make_model_dir(tmp_path / "b", cfg, stop_token_ids=[stop])
    lm2 = LM.LanguageModel(hip_ctx, d2, max_context_length=256)
    res2 = lm2.generate(ids, max_tokens=12, chunk=3)
    assert res2.token_ids == want[:first_hit] and res2.stopped_on == stop
    lm2.close()
    # sampling defaults from generation_config: stochastic, reproducible per seed, greedy override
    d3, _ = make_model_dir(tmp_path / "c", cfg, temperature=30.0, top_k=50)
    lm3 = LM.LanguageModel(hip_ctx, d3, max_context_length=256)
    a, b, c = lm3.generate(ids, max_tokens=10, seed=7), lm3.generate(ids, max_tokens=10, seed=7), lm3.generate(ids, max_tokens=10, seed=8)
    assert a.token_ids == b.token_ids and a.token_ids != c.token_ids
    assert lm3.generate(ids, max_tokens=10, greedy=True).token_ids == want[:10]
    lm3.close()
