"""Host side of the DFlash tree speculator (crates/backend-uzu/src/speculators/dflash_tfm.rs) and of the speculative leg of
``LanguageModelStream`` (engine/language_model/stream/stream.rs:299-318 prefill taps, 380-470 accept, 551-628 propose + verify).

The device work sits behind two duck-typed objects (the parity tests drive a CPU checker through the same host code and compare the tries and token
streams):

  target   prefill(tokens) -> token, verify_tree(token_ids, nodes[, seeds]) -> sampled tokens, accept(indices), hidden_features() -> [rows, d] per tapped
           layer of the LAST pass, context_length                        (uzu_amd.engine.HipModel)
  drafter  accept(features, indices), draft(target, token, depth) -> (draft_hidden, logits, tokens), block_size, target_layer_ids, context_length
                                                                         (uzu_amd.engine.HipDrafter)

Same names and meaning as the reference:
  DFlashTfmTreeShape / DFlashTfmTreeConstructionMethod     dflash_tfm.rs:57-72
  DFlashTfmSpeculator.propose_tree                         dflash_tfm.rs:133-343  (Argmax construction; the Weaver construction needs the Weaver block's weights)
  ProposalNode                                             encodable_block/weaver.rs (token_id, depth, logprob, child_indices)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .trie import FlatTrie, PRng, TrieNode


class InvalidTreeShape(ValueError):
    """DFlashTreeError::InvalidTreeShape (dflash_tfm.rs:28-39)"""


@dataclass
class ProposalNode:
    token_id: int
    depth: int
    logprob: float = 0.0
    child_indices: List[int] = field(default_factory=list)


@dataclass
class TreeShape:
    """DFlashTfmTreeShape (dflash_tfm.rs:66-72); construction_method "argmax" (a chain of the drafter's greedy tokens)."""
    tree_budget: int
    max_tree_depth: int = 16
    dflash_depth_override: Optional[int] = None
    construction_method: str = "argmax"


def build_trie(nodes: Sequence[ProposalNode], root_position: int, prng: PRng, tree_budget: int) -> TrieNode:
    """recursive_build + prune_to_budget (dflash_tfm.rs:296-342, without the grammar leg): node seeds = prng.derive(root_position + depth)."""
    def build(index: int) -> TrieNode:
        n = nodes[index]
        t = TrieNode(n.token_id, prng.derive(root_position + n.depth), n.logprob)
        for c in n.child_indices:
            t.add(build(c))  # "tree children are selected without replacement": a duplicate raises, as the reference's expect() would panic
        return t
    trie = build(0)
    trie.prune_to_budget(tree_budget)
    return trie


class DFlashSpeculator:
    """DFlashTfmSpeculator (dflash_tfm.rs:74-131) over a drafter object."""

    def __init__(self, drafter):
        self.drafter = drafter

    def has_weaver(self) -> bool:
        return False

    def hidden_feature_layer_indices(self) -> List[int]:
        return list(self.drafter.target_layer_ids)

    def encode_accept(self, target_features, accepted_indices):
        self.drafter.accept(target_features, accepted_indices)

    def propose_tree(self, target, target_output_token: int, shape: TreeShape, prng: PRng) -> TrieNode:
        if shape.tree_budget < 2:
            raise AssertionError("tree budget needs at least a root and one draft token")  # dflash_tfm.rs:144
        block_size = self.drafter.block_size
        dflash_depth = block_size if shape.dflash_depth_override is None else shape.dflash_depth_override
        if not (2 <= dflash_depth <= block_size):
            raise InvalidTreeShape(f"dflash depth {dflash_depth} is outside 2..={block_size}")
        root_position = self.drafter.context_length
        if shape.construction_method != "argmax":
            raise InvalidTreeShape("weaver tree construction requires a speculator with weaver weights")
        if shape.tree_budget > dflash_depth:
            raise InvalidTreeShape(f"argmax chain of {shape.tree_budget} nodes needs {shape.tree_budget - 1} draft rows, dflash depth is {dflash_depth}")
        chain_length = shape.tree_budget - 1
        nodes = [ProposalNode(int(target_output_token), 0, 0.0, [1])]
        _hidden, _logits, tokens = self.drafter.draft(target, int(target_output_token), dflash_depth)
        # Sampling over the first chain_length lookahead rows, Greedy (dflash_tfm.rs:181-203)
        for depth, token in enumerate([int(t) for t in tokens[:chain_length]], start=1):
            nodes.append(ProposalNode(token, depth, 0.0, [depth + 1] if depth < chain_length else []))
        return build_trie(nodes, root_position, prng, shape.tree_budget)


class SpeculativeStream:
    """The speculative leg of LanguageModelStream: prefill with the target's hidden-feature taps feeding the drafter (stream.rs:299-318), then rounds of
    propose_tree -> one verify pass over the linearised tree -> FlatTrie::accept -> TransformerState / speculator encode_accept (stream.rs:380-470,551-628)."""

    def __init__(self, target, speculator: DFlashSpeculator, seed: int = 0, speculation_batch: int = 16, prefill_chunk: int = 1024):
        self.target, self.speculator, self.prng = target, speculator, PRng(seed)
        self.speculation_batch, self.prefill_chunk = speculation_batch, prefill_chunk
        self.tokens: List[int] = []
        self.rounds = self.proposed = self.accepted = 0
        self.tries: List[FlatTrie] = []  # every round's linearised tree (parity tests compare them)

    def prefill(self, prompt: Sequence[int]) -> int:
        prompt = np.asarray(prompt, dtype=np.uint32)
        token = None
        for s in range(0, prompt.size, self.prefill_chunk):  # one pass per chunk; every chunk's rows are accepted by the drafter
            chunk = prompt[s:s + self.prefill_chunk]
            token = self.target.prefill(chunk)
            self.speculator.encode_accept(self.target.hidden_features(), np.arange(chunk.size, dtype=np.uint32))
        self.tokens = [int(token)]
        return int(token)

    def round(self) -> List[int]:
        """One speculation round; returns the tokens it emitted (>= 1: the token sampled at the last accepted node is always new)."""
        budget = min(self.speculation_batch, self.speculator.drafter.block_size)
        trie = self.speculator.propose_tree(self.target, self.tokens[-1], TreeShape(tree_budget=budget, max_tree_depth=16), self.prng)
        flat = trie.linearize()
        self.tries.append(flat)
        sampled = self.target.verify_tree(flat.token_ids(), flat.nodes(), flat.token_seeds())
        full = flat.accept(sampled)
        indices = np.array([i for i, _, _ in full], dtype=np.uint32)
        feats = self.target.hidden_features()  # the tree pass's rows: taken BEFORE the accept moves the target on
        self.target.accept(indices)
        self.speculator.encode_accept(feats, indices)
        out = [int(t) for _, _, t in full]
        self.tokens += out
        self.rounds += 1
        self.proposed += len(flat) - 1
        self.accepted += len(full) - 1
        return out

    def generate(self, count: int) -> List[int]:
        while len(self.tokens) < count + 1:
            self.round()
        return self.tokens[1:count + 1]
