"""Model-directory loader (uzu_amd/loader.py) against the reference's on-disk rules (SURVEY.md Appendix B;
engine/language_model/mod.rs:57-130, parameters/safetensors_metadata.rs:40-127, parameters/loader.rs:162-250,
encodable_block/weight_matrix.rs:101-162, the #[uzu_config] strictness of crates/backend-uzu-macros/src/uzu_config.rs)."""
import dataclasses
import json
import os

import numpy as np
import pytest

from uzu_amd import desc as D
from uzu_amd import loader as L
from uzu_amd import synthetic as S

import oracle.oracle as O


def bundles_equal(a, b, path="bundle"):
    """Deep comparison of two ModelBundles (arrays bit-exact, scalars equal)."""
    if dataclasses.is_dataclass(a):
        assert type(a) is type(b), path
        for f in dataclasses.fields(a):
            if f.name in ("_keep", "name", "max_context_length"):
                continue
            bundles_equal(getattr(a, f.name), getattr(b, f.name), f"{path}.{f.name}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            bundles_equal(x, y, f"{path}[{i}]")
    elif isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        assert a is not None and b is not None, path
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), path
    elif isinstance(a, float):
        assert np.float32(a) == np.float32(b), f"{path}: {a} != {b}"
    else:
        assert a == b, f"{path}: {a!r} != {b!r}"


def variants():
    rng = np.random.default_rng(3)
    yield "tiny-qwen", S.tiny_qwen()
    yield "tiny-llama-zp", S.tiny_llama()
    yield "int8-symmetric-yarn", S.tiny_llama(bits=8, method=D.QUANT_SCALE_SYMMETRIC, tied_embeddings=True,
                                              rope=D.RopeConfig(kind=D.ROPE_YARN, head_dim=64, max_sequence_length=8192, base=10000.0, scaling_factor=4.0,
                                                                original_context_length=1024, beta_fast=32.0, beta_slow=1.0, truncate=False))
    yield "rht-qwen", S.tiny_qwen(rht=True)   # HybridSpec InputOutput linears: weights.quantized.* + weights.incoherence_signs.*
    yield "rht-llama-int8", S.tiny_llama(rht=True, bits=8, method=D.QUANT_SCALE_BIAS)
    yield "longrope", S.tiny_llama(rope=D.RopeConfig(kind=D.ROPE_LONGROPE, head_dim=64, max_sequence_length=8192, base=10000.0, scaling_factor=8.0,
                                                     original_context_length=1024, short_factor=rng.uniform(1.0, 1.2, 32).astype(np.float32),
                                                     long_factor=rng.uniform(1.0, 6.0, 32).astype(np.float32)))
    # the Gemma-family layer / decoder options (transformer_layer.rs:61-184, decoder.rs:68-99): two RoPE configurations, sandwich norms with
    # post-layer scalars, value normalisation, two layers sharing an earlier layer's KV state, per-layer embeddings; + embedding norm, no first pre-mixer norm
    yield "tiny-gemma", S.tiny_gemma(ple_dim=32)
    yield "tiny-gemma-embedding-norm", S.tiny_gemma(ple_dim=0, embedding_norm=True, first_layer_without_pre_mixer_norm=True, kv_sharing={4: 1})


@pytest.mark.parametrize("name,cfg", list(variants()), ids=[n for n, _ in variants()])
def test_round_trip_is_lossless_and_the_oracle_agrees(tmp_path, name, cfg):
    bundle = S.build_model(cfg)
    if name == "tiny-llama-zp":  # Linear biases (has_qkv_biases / has_out_biases / has_up_biases): bf16 [n] next to the weights tree
        lw = bundle.layers[0]
        for lin in (lw.qkv_projection, lw.out_projection, lw.up_projection, lw.down_projection):
            lin.out_biases = S.f32_to_bf16_bits(np.linspace(-0.1, 0.1, lin.n).astype(np.float32))
        for l in bundle.layers[1:]:  # the config flags are per layer
            for lin in (l.qkv_projection, l.out_projection, l.up_projection, l.down_projection):
                lin.out_biases = S.f32_to_bf16_bits(np.zeros(lin.n, np.float32))
    L.save_model_dir(bundle, str(tmp_path))
    loaded = L.load_model_dir(str(tmp_path), max_context_length=bundle.max_context_length)
    bundles_equal(bundle, loaded)
    prompt = S.synthetic_prompt(12, cfg.vocab_size)
    t0, l0 = O.OracleModel(bundle).prefill(prompt, True)
    t1, l1 = O.OracleModel(loaded).prefill(prompt, True)
    assert t0 == t1 and np.array_equal(l0, l1)


def test_rht_linears_change_the_forward_and_are_refused_where_unsupported(tmp_path):
    """HybridSpec (config/weight_matrix/hybrid_spec.rs): only {no adapter, block 32, input_output} maps onto RHTLinearWrapper
    (linear/mod.rs:118-136); the sign vectors are part of the forward (another model without them)."""
    cfg = S.tiny_llama(rht=True)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(9, cfg.vocab_size)
    _, with_rht = O.OracleModel(bundle).prefill(prompt, True)
    plain = S.build_model(S.tiny_llama())
    _, without = O.OracleModel(plain).prefill(prompt, True)
    assert not np.array_equal(with_rht, without)
    d = str(tmp_path)
    L.save_model_dir(bundle, d)
    path = os.path.join(d, "model.safetensors")
    st = L.SafeTensors(path)
    tensors = {k: (st.index[k][0], st.tensor(k, st.index[k][1], st.index[k][0])) for k in st.index}
    key = "decoder.transformer.layers.0.mlp.down_projection.weights.spec"
    for field, value, what in (("incoherence_processing_mode", "input", "incoherence processing"), ("incoherence_block_size", 64, "incoherence processing")):
        meta = dict(st.metadata)
        meta[key] = json.dumps({**json.loads(meta[key]), field: value})
        L.write_safetensors(path, tensors, meta)
        with pytest.raises(L.UnsupportedModelError, match=what):
            L.load_model_dir(d)
    # a LowRankSpec adapter selects QLoRALinearWrapper (linear/mod.rs:133-158): its tensors must be there, the rank positive, the base quantized
    meta = dict(st.metadata)
    meta[key] = json.dumps({**json.loads(meta[key]), "adapter_spec": {"type": "LowRankSpec", "rank": 8, "layout": "output_input"}})
    L.write_safetensors(path, tensors, meta)
    with pytest.raises(L.ModelFormatError, match="adapter.down_projection' not found"):
        L.load_model_dir(d)
    meta[key] = json.dumps({**json.loads(meta[key]), "adapter_spec": {"type": "LowRankSpec", "rank": 0, "layout": "output_input"}})
    L.write_safetensors(path, tensors, meta)
    with pytest.raises(L.ModelFormatError, match="adapter rank must be positive"):
        L.load_model_dir(d)
    with pytest.raises(NotImplementedError):  # tensor-parallel shards of RHT linears are not planned yet
        bundle.layers[0].up_projection.rows(0, 64)


def test_written_file_is_a_valid_safetensors_file_for_an_independent_reader(tmp_path):
    """The `safetensors` package (the format's own implementation) reads what write_safetensors wrote, and SafeTensors reads
    what the package wrote: header length, JSON header, __metadata__, dtypes, offsets."""
    st_np = pytest.importorskip("safetensors.numpy")
    from safetensors import safe_open
    bundle = S.build_model(S.tiny_llama())
    L.save_model_dir(bundle, str(tmp_path))
    path = os.path.join(str(tmp_path), "model.safetensors")
    ours = L.SafeTensors(path)
    with safe_open(path, framework="np") as f:
        assert set(f.keys()) == set(ours.index)
        assert f.metadata() == ours.metadata
        key = "decoder.transformer.layers.0.pre_mixer_norm.scales"
        assert np.array_equal(f.get_tensor(key), ours.tensor(key, (256,), "F32"))
        key = "decoder.transformer.layers.0.mixer.qkv_projection.weights.weights"
        want = f.get_tensor(key)
        assert np.array_equal(want, ours.tensor(key, want.shape, "U8"))
    other = os.path.join(str(tmp_path), "theirs.safetensors")
    a, b = np.arange(12, dtype=np.float32).reshape(3, 4), np.arange(6, dtype=np.uint8)
    st_np.save_file({"x.y": a, "z": b}, other, metadata={"x.spec": '{"type":"MLXSpec"}'})
    theirs = L.SafeTensors(other)
    assert np.array_equal(theirs.tensor("x.y", (3, 4), "F32"), a) and np.array_equal(theirs.tensor("z", (6,), "U8"), b)
    assert theirs.spec("x.spec") == {"type": "MLXSpec"}
    theirs.assert_all_consumed()


def edit_config(dirpath, fn):
    p = os.path.join(dirpath, "config.json")
    with open(p) as f:
        cfg = json.load(f)
    fn(cfg)
    with open(p, "w") as f:
        json.dump(cfg, f)


def test_strictness_matches_the_reference(tmp_path):
    bundle = S.build_model(S.tiny_qwen())
    d = str(tmp_path)
    layer0 = lambda c: c["decoder_config"]["transformer_config"]["layer_configs"][0]

    def fresh():
        L.save_model_dir(bundle, d)

    fresh()  # every field is required: an Option may be null but not absent (strict_serde::required)
    edit_config(d, lambda c: layer0(c).pop("kv_source_layer_index"))
    with pytest.raises(L.ModelFormatError, match="kv_source_layer_index is missing"):
        L.load_model_dir(d)
    fresh()  # deny_unknown_fields
    edit_config(d, lambda c: layer0(c)["mixer_config"].update(window=7))
    with pytest.raises(L.ModelFormatError, match="unknown field"):
        L.load_model_dir(d)
    fresh()  # tagged families: "type" must name a member
    edit_config(d, lambda c: layer0(c)["mlp_config"]["activation"].update(type="Swish"))
    with pytest.raises(L.ModelFormatError, match="type is 'Swish'"):
        L.load_model_dir(d)
    fresh()  # well-formed but outside the hot path: loud, not silent
    # (mixture-of-experts MLPs load since round 6 -- tests/test_oracle_moe.py; a DenseMLPConfig's fields under that tag are a format error, and a Mamba mixer stays refused)
    edit_config(d, lambda c: layer0(c)["mlp_config"].update(type="MixtureOfExpertsConfig"))
    with pytest.raises(L.ModelFormatError, match="expert_config is missing"):
        L.load_model_dir(d)
    fresh()
    edit_config(d, lambda c: layer0(c)["mixer_config"].update(type="Mamba2Config"))
    with pytest.raises(L.UnsupportedModelError, match="Mamba2 mixer"):
        L.load_model_dir(d)
    fresh()
    # sliding windows load since round 3 (ring KV state in the engine); a non-positive window is a format error, sinks need their tensor
    edit_config(d, lambda c: c["decoder_config"]["transformer_config"]["layer_configs"][3]["mixer_config"].update(sliding_window_size=128))
    assert L.load_model_dir(d).layers[3].sliding_window_size == 128
    edit_config(d, lambda c: c["decoder_config"]["transformer_config"]["layer_configs"][3]["mixer_config"].update(sliding_window_size=0))
    with pytest.raises(L.ModelFormatError, match="sliding_window_size must be positive"):
        L.load_model_dir(d)
    fresh()
    edit_config(d, lambda c: c["decoder_config"]["transformer_config"]["layer_configs"][3]["mixer_config"].update(has_sinks=True))
    with pytest.raises(L.ModelFormatError, match="sinks' not found"):
        L.load_model_dir(d)
    fresh()
    edit_config(d, lambda c: c["decoder_config"]["transformer_config"]["layer_configs"][3]["mixer_config"].update(logit_soft_cap=30.0))
    with pytest.raises(L.UnsupportedModelError, match="soft-capping"):
        L.load_model_dir(d)


def test_tensor_validation_matches_the_reference(tmp_path):
    """validate(shape, dtype) per leaf (loader.rs:162-178), spec from __metadata__, every tensor consumed (loader.rs:230-250)."""
    cfg = S.tiny_llama()
    bundle = S.build_model(cfg)
    d = str(tmp_path)
    L.save_model_dir(bundle, d)
    path = os.path.join(d, "model.safetensors")
    st = L.SafeTensors(path)
    tensors = {k: (st.index[k][0], st.tensor(k, st.index[k][1], st.index[k][0])) for k in st.index}
    meta = dict(st.metadata)

    extra = dict(tensors)
    extra["decoder.transformer.layers.0.mixer.unused"] = ("F32", np.zeros(4, np.float32))
    L.write_safetensors(path, extra, meta)
    with pytest.raises(L.ModelFormatError, match="never read"):
        L.load_model_dir(d)

    wrong = dict(tensors)
    key = "decoder.transformer.layers.1.mlp.down_projection.weights.scales"
    wrong[key] = ("BF16", np.ascontiguousarray(tensors[key][1][:, :-1]))
    L.write_safetensors(path, wrong, meta)
    with pytest.raises(L.ModelFormatError, match="expected BF16"):
        L.load_model_dir(d)

    nospec = dict(meta)
    del nospec["decoder.transformer.layers.2.mixer.out_projection.weights.spec"]
    L.write_safetensors(path, tensors, nospec)
    with pytest.raises(L.ModelFormatError, match="metadata entry"):
        L.load_model_dir(d)

    layout = dict(meta)  # a linear must be OutputInput (weight_matrix.rs:113-117)
    k2 = "decoder.transformer.layers.0.mlp.up_projection.weights.spec"
    layout[k2] = json.dumps({**json.loads(meta[k2]), "layout": "input_output"})
    L.write_safetensors(path, tensors, layout)
    with pytest.raises(L.ModelFormatError, match="expected output_input layout"):
        L.load_model_dir(d)

    with open(path, "r+b") as f:  # truncated header
        f.write((10 ** 12).to_bytes(8, "little"))
    with pytest.raises(L.ModelFormatError, match="header length"):
        L.load_model_dir(d)


def test_gemma_family_options_reach_the_config_and_inconsistent_ones_are_refused(tmp_path):
    """What save_model_dir writes for the options is what the reference's config structs name (TransformerLayerConfig::{ple_config,
    has_post_layer_scalar, kv_source_layer_index, rope_config}, AttentionConfig::{normalize_values, is_kv_sharing}, DecoderConfig::
    {ple_model_config, embedding_norm_config}); the reference's construction errors are load errors here."""
    bundle = S.build_model(S.tiny_gemma(ple_dim=32, embedding_norm=True))
    L.save_model_dir(bundle, str(tmp_path))
    cfg_path = os.path.join(str(tmp_path), "config.json")
    cfg = json.load(open(cfg_path))
    dec = cfg["decoder_config"]
    layers = dec["transformer_config"]["layer_configs"]
    assert dec["ple_model_config"]["num_layers"] == 5 and dec["ple_model_config"]["ple_dim"] == 32 and dec["embedding_norm_config"] is not None
    assert [l["kv_source_layer_index"] for l in layers] == [None, None, None, 0, 1]
    assert [l["mixer_config"]["is_kv_sharing"] for l in layers] == [False, False, False, True, True]
    assert all(l["has_post_layer_scalar"] and l["ple_config"]["ple_dim"] == 32 for l in layers)
    assert layers[0]["rope_config"]["type"] == "UnscaledRoPEConfig" and layers[1]["rope_config"]["type"] == "LinearScalingRoPEConfig"
    assert layers[0]["mixer_config"]["normalize_values"] is True
    loaded = L.load_model_dir(str(tmp_path))
    assert loaded.ropes is not None and [l.rope_index for l in loaded.layers] == [0, 1, 0, 0, 1]
    assert loaded.layers[3].qkv_projection.n == 4 * 64 and loaded.layers[3].kv_source_layer_index == 0

    def broken(mutate, match, exc=L.ModelFormatError):
        c = json.loads(json.dumps(cfg))
        mutate(c["decoder_config"])
        json.dump(c, open(cfg_path, "w"))
        with pytest.raises(exc, match=match):
            L.load_model_dir(str(tmp_path))
    lc = lambda d, i: d["transformer_config"]["layer_configs"][i]
    broken(lambda d: lc(d, 3)["mixer_config"].update(is_kv_sharing=False), "is_kv_sharing")
    broken(lambda d: lc(d, 3).update(kv_source_layer_index=4), "not an earlier attention layer")
    broken(lambda d: lc(d, 4).update(kv_source_layer_index=3), "not an earlier attention layer")
    broken(lambda d: lc(d, 4).update(kv_source_layer_index=0), "differ in sliding window")
    broken(lambda d: lc(d, 1).update(post_mlp_norm_config=None), "post-layer scalar")
    broken(lambda d: lc(d, 1).update(pre_mixer_norm_config=None), "pre_mixer_norm_config")
    broken(lambda d: d["ple_model_config"].update(num_layers=4), "num_layers")
    broken(lambda d: d.update(ple_model_config=None), "without decoder_config.ple_model_config")
    broken(lambda d: lc(d, 2)["ple_config"].update(ple_dim=16), "ple_dim")


def test_kv_sharing_layer_with_a_key_norm_config_loads_like_the_reference(tmp_path):
    """lalamo still writes `key_norm_config` on KV-sharing layers (the TODO at mixer/attention/mod.rs:134): the reference drops the key / value norms of such a
    layer and never opens `key_norm.scales`.  A config that carries the entry therefore loads -- and, exactly as in the reference, a checkpoint that ALSO holds the
    tensor is refused, because LanguageModel::load ends with assert_all_tensors_validated (engine/language_model/mod.rs:98, parameters/loader.rs:230-250: every key
    of the index must have been validated), which lists the unread `...mixer.key_norm.scales`."""
    import numpy as np
    bundle = S.build_model(S.tiny_gemma(ple_dim=0))
    d = str(tmp_path)
    L.save_model_dir(bundle, d)
    cfg_path = os.path.join(d, "config.json")
    cfg = json.load(open(cfg_path))
    layers = cfg["decoder_config"]["transformer_config"]["layer_configs"]
    sharing = [i for i, l in enumerate(layers) if l["mixer_config"]["is_kv_sharing"]]
    assert sharing and all(layers[i]["mixer_config"]["key_norm_config"] is None for i in sharing)
    donor = next(l["mixer_config"]["key_norm_config"] or l["mixer_config"]["query_norm_config"] for l in layers if not l["mixer_config"]["is_kv_sharing"])
    assert donor is not None
    for i in sharing:
        layers[i]["mixer_config"]["key_norm_config"] = donor
    json.dump(cfg, open(cfg_path, "w"))
    loaded = L.load_model_dir(d)  # the config entry alone is ignored (key norm dropped under sharing)
    assert all(not loaded.layers[i].key_norm.present for i in sharing)
    bundles_equal(bundle, loaded)
    # ... with the tensor in the file as well: the reference's load fails with UnvalidatedTensors; so does this one
    st_path = os.path.join(d, "model.safetensors")
    raw = open(st_path, "rb").read()
    hlen = int.from_bytes(raw[:8], "little")
    header = json.loads(raw[8:8 + hlen])
    data = raw[8 + hlen:]
    hd = loaded.layers[sharing[0]].head_dim
    key = f"decoder.transformer.layers.{sharing[0]}.mixer.key_norm.scales"
    header[key] = {"dtype": "F32", "shape": [hd], "data_offsets": [len(data), len(data) + 4 * hd]}
    data += np.ones(hd, dtype="<f4").tobytes()
    hj = json.dumps(header).encode()
    hj += b" " * (-len(hj) % 8)
    open(st_path, "wb").write(len(hj).to_bytes(8, "little") + hj + data)
    with pytest.raises(L.ModelFormatError, match="never read"):
        L.load_model_dir(d)
