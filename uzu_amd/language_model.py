"""Text front-end of the engine: `Engine::load_language_model` + the greedy / stochastic generate loop.

Mirrors crates/backend-uzu/src/engine/language_model/mod.rs:57-130: a model directory holds `config.json`, `model.safetensors` and
`tokenizer.json`; the tokenizer is the HuggingFace `tokenizers` format, read by the same library (the reference links the Rust crate
`tokenizers` 0.22, this module the Python binding of the same code base).  Everything numeric stays in uzu_amd.engine.HipModel; this
module only turns text into token ids and back, applies `generation_config` (stop tokens, sampling defaults: config/model/
language_model.rs) and reports the reference's two throughput figures (chat/token.rs:393-406: prefill t/s = prompt tokens / time to
first token, decode t/s = generated tokens / (t_last - t_first)).

Not on the hot path: no kernel, no device memory.  Chat templates / tool-call parsing (crates/nagare) stay out of scope.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np

from . import loader
from .backend import Context
from .engine import MODEL_DEFAULT, HipModel


class TokenizerError(RuntimeError):
    """tokenizer.json missing or unreadable (EngineLoadLanguageModelError::Tokenizer)."""


@dataclass
class GenerationConfig:
    """config.json `generation_config` (config/model/language_model.rs): stop tokens and sampling defaults."""
    stop_token_ids: List[int] = field(default_factory=list)
    temperature: Optional[float] = None
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    min_p: Optional[float] = None

    @property
    def stochastic(self) -> bool:
        return any(v is not None for v in (self.temperature, self.top_k, self.top_p, self.min_p))


@dataclass
class GenerateResult:
    text: str
    token_ids: List[int]
    prompt_tokens: int
    stopped_on: Optional[int]          # the stop token that ended the stream (not part of token_ids), or None (max_tokens reached)
    prefill_tokens_per_s: float        # tokens_input / (t_first_token - t_prefill_start)
    decode_tokens_per_s: float         # tokens_output / (t_last_token - t_first_token); 0 for a single token


def load_tokenizer(model_dir: str):
    """`Tokenizer::from_file(model_path.join("tokenizer.json"))` (mod.rs:72)."""
    path = os.path.join(model_dir, "tokenizer.json")
    if not os.path.exists(path):
        raise TokenizerError(f"{path} not found: a model directory holds config.json, model.safetensors and tokenizer.json")
    try:
        from tokenizers import Tokenizer
    except ImportError as exc:  # pragma: no cover -- the image ships tokenizers 0.22
        raise TokenizerError(f"the `tokenizers` package is required for the text front-end: {exc}") from None
    try:
        return Tokenizer.from_file(path)
    except Exception as exc:  # noqa: BLE001 -- tokenizers raises a plain Exception with the serde message
        raise TokenizerError(f"{path}: {exc}") from None


def read_generation_config(model_dir: str) -> GenerationConfig:
    with open(os.path.join(model_dir, "config.json"), "r", encoding="utf-8") as f:
        g = json.load(f).get("generation_config") or {}
    return GenerationConfig(stop_token_ids=[int(t) for t in (g.get("stop_token_ids") or [])], temperature=g.get("temperature"), top_k=g.get("top_k"),
                            top_p=g.get("top_p"), min_p=g.get("min_p"))


class LanguageModel:
    """config.json + model.safetensors + tokenizer.json -> text in, text out on the HIP engine."""

    def __init__(self, ctx: Context, model_dir: str, max_context_length: Optional[int] = None, flags: int = MODEL_DEFAULT):
        self.ctx = ctx
        self.tokenizer = load_tokenizer(model_dir)
        self.generation_config = read_generation_config(model_dir)
        self.bundle = loader.load_model_dir(model_dir, max_context_length=max_context_length)
        vocab = self.tokenizer.get_vocab_size(with_added_tokens=True)
        if vocab > self.bundle.vocab_size:
            raise TokenizerError(f"tokenizer.json has {vocab} tokens, the model's embedding {self.bundle.vocab_size} rows")
        self.model = HipModel(ctx, self.bundle, flags)

    def close(self):
        self.model.close()

    # ---- text <-> ids
    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        return list(self.tokenizer.encode(text, add_special_tokens=add_special_tokens).ids)

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        return self.tokenizer.decode([int(t) for t in ids], skip_special_tokens=skip_special_tokens)

    # ---- generation
    def generate(self, prompt: Union[str, Sequence[int]], max_tokens: int = 64, seed: Optional[int] = None, greedy: Optional[bool] = None,
                 chunk: int = 16) -> GenerateResult:
        """Prefill the prompt (chunks of <= 1024 tokens inside the engine), then chained decode until a stop token of generation_config
        or `max_tokens`.  Sampling: greedy unless generation_config carries sampling parameters (then stochastic with `seed`, default 0;
        `greedy=True` overrides).  Decode runs `chunk` graph replays per host round trip; tokens behind a stop token are discarded (the
        reference checks after every token: the text is the same, the engine ran up to chunk - 1 steps more)."""
        ids = self.encode(prompt) if isinstance(prompt, str) else [int(t) for t in prompt]
        if not ids:
            raise ValueError("empty prompt")
        g = self.generation_config
        stochastic = g.stochastic and not greedy
        self.model.reset()
        self.model.set_sampling(seed=(seed or 0), temperature=g.temperature, top_k=g.top_k, top_p=g.top_p, min_p=g.min_p) if stochastic else self.model.set_sampling(None)
        stops = set(g.stop_token_ids)
        t0 = time.perf_counter()
        first = self.model.prefill(np.asarray(ids, dtype=np.uint32))
        t_first = time.perf_counter()
        out: List[int] = []
        stopped: Optional[int] = None
        tok = int(first)
        t_last = t_first
        room = self.bundle.max_context_length - len(ids)
        while True:
            if tok in stops:
                stopped = tok
                break
            out.append(tok)
            if len(out) >= max_tokens or len(out) >= room:
                break
            n = min(chunk, max_tokens - len(out), room - len(out))
            toks, _ = self.model.decode(n)
            t_last = time.perf_counter()
            done = False
            for t in toks[:-1]:
                t = int(t)
                if t in stops:
                    stopped, done = t, True
                    break
                out.append(t)
                if len(out) >= max_tokens:
                    done = True
                    break
            if done:
                break
            tok = int(toks[-1])
        decode_s = t_last - t_first
        return GenerateResult(text=self.decode(out), token_ids=out, prompt_tokens=len(ids), stopped_on=stopped,
                              prefill_tokens_per_s=len(ids) / max(t_first - t0, 1e-9), decode_tokens_per_s=(len(out) - 1) / decode_s if len(out) > 1 and decode_s > 0 else 0.0)
