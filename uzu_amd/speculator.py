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
    """DFlashTfmTreeShape (dflash_tfm.rs:57-72); construction_method "argmax" (a chain of the drafter's greedy tokens) or "weaver" (rounds, expand_per_round,
    expand_width: DFlashTfmTreeConstructionMethod::Weaver)."""
    tree_budget: int
    max_tree_depth: int = 16
    dflash_depth_override: Optional[int] = None
    construction_method: str = "argmax"
    rounds: int = 16
    expand_per_round: int = 4
    expand_width: int = 4


TREE_FIELDS = {"token_id": 0, "parent_slot": 1, "depth": 2, "path_logprob_bits": 3, "edge_logprob_bits": 4, "valid": 5}                       # TreeIdx (gpu_types/weaver.rs:23-35)
FRONTIER_FIELDS = {"token_id": 0, "parent_slot": 1, "depth": 2, "path_logprob_bits": 3, "edge_logprob_bits": 4, "path_score_key": 5, "active": 6}  # FrontierIdx (:5-17)


def read_nodes(packed_tree: np.ndarray, frontier: np.ndarray) -> List[ProposalNode]:
    """EncodedWeaverTree::read_nodes (encodable_block/weaver.rs:60-113): the valid tree slots in slot order, then the active frontier slots, as ProposalNodes whose
    child lists follow that order."""
    packed_tree, frontier = np.asarray(packed_tree, np.uint32), np.asarray(frontier, np.uint32)
    tree_slot_count, frontier_capacity = packed_tree.shape[1], frontier.shape[1]
    bits_to_f32 = lambda b: float(np.array([b], np.uint32).view(np.float32)[0])
    slot_to_index = [None] * tree_slot_count
    nodes: List[ProposalNode] = []
    for slot in range(tree_slot_count):
        if packed_tree[TREE_FIELDS["valid"], slot] == 0:
            continue
        parent_slot = int(np.int32(packed_tree[TREE_FIELDS["parent_slot"], slot]))
        parent = None
        if parent_slot >= 0:
            parent = slot_to_index[parent_slot]
            assert parent is not None, f"tree slot {slot} names padding slot {parent_slot} as its parent"
        slot_to_index[slot] = len(nodes)
        if parent is not None:
            nodes[parent].child_indices.append(len(nodes))
        nodes.append(ProposalNode(int(packed_tree[TREE_FIELDS["token_id"], slot]), int(packed_tree[TREE_FIELDS["depth"], slot]),
                                  bits_to_f32(packed_tree[TREE_FIELDS["edge_logprob_bits"], slot]), []))
    for slot in range(frontier_capacity):
        if frontier[FRONTIER_FIELDS["active"], slot] == 0:
            continue
        parent = slot_to_index[int(frontier[FRONTIER_FIELDS["parent_slot"], slot])]
        assert parent is not None, f"frontier slot {slot} names a padding slot as its parent"
        nodes[parent].child_indices.append(len(nodes))
        nodes.append(ProposalNode(int(frontier[FRONTIER_FIELDS["token_id"], slot]), int(frontier[FRONTIER_FIELDS["depth"], slot]),
                                  bits_to_f32(frontier[FRONTIER_FIELDS["edge_logprob_bits"], slot]), []))
    return nodes


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

    def __init__(self, drafter, weaver=None):
        """`weaver`: the Weaver object of a speculator whose checkpoint carries weaver weights (encode_tree(target, target_hidden, draft_hidden, logits, depth_seeds,
        root_token_id, shape) -> (packed_tree, frontier) | None; max_depth)"""
        self.drafter, self.weaver = drafter, weaver

    def has_weaver(self) -> bool:
        return self.weaver is not None

    def hidden_feature_layer_indices(self) -> List[int]:
        return list(self.drafter.target_layer_ids)

    def encode_accept(self, target_features, accepted_indices):
        self.drafter.accept(target_features, accepted_indices)

    def propose_tree(self, target, target_output_token: int, shape: TreeShape, prng: PRng, target_output_norm=None) -> TrieNode:
        """`target_output_norm`: bf16 bits [1, target model_dim], the target's output-norm row of the position that sampled `target_output_token`
        (ForwardPassChaining's output_norm, stream.rs:466-476,558-569): the Weaver construction's prefix row 0"""
        if shape.tree_budget < 2:
            raise AssertionError("tree budget needs at least a root and one draft token")  # dflash_tfm.rs:144
        block_size = self.drafter.block_size
        dflash_depth = block_size if shape.dflash_depth_override is None else shape.dflash_depth_override
        if not (2 <= dflash_depth <= block_size):
            raise InvalidTreeShape(f"dflash depth {dflash_depth} is outside 2..={block_size}")
        root_position = self.drafter.context_length
        if shape.construction_method == "weaver":
            return self._propose_weaver(target, target_output_token, shape, prng, dflash_depth, root_position, target_output_norm)
        if shape.construction_method != "argmax":
            raise InvalidTreeShape(f"unknown construction method {shape.construction_method!r}")
        if shape.tree_budget > dflash_depth:
            raise InvalidTreeShape(f"argmax chain of {shape.tree_budget} nodes needs {shape.tree_budget - 1} draft rows, dflash depth is {dflash_depth}")
        chain_length = shape.tree_budget - 1
        nodes = [ProposalNode(int(target_output_token), 0, 0.0, [1])]
        _hidden, _logits, tokens = self.drafter.draft(target, int(target_output_token), dflash_depth)
        # Sampling over the first chain_length lookahead rows, Greedy (dflash_tfm.rs:181-203)
        for depth, token in enumerate([int(t) for t in tokens[:chain_length]], start=1):
            nodes.append(ProposalNode(token, depth, 0.0, [depth + 1] if depth < chain_length else []))
        return build_trie(nodes, root_position, prng, shape.tree_budget)


    def _propose_weaver(self, target, target_output_token, shape, prng, dflash_depth, root_position, target_output_norm) -> TrieNode:
        """DFlashTfmTreeConstructionMethod::Weaver (dflash_tfm.rs:224-292)"""
        from .desc import WeaverTreeShape
        if self.weaver is None:
            raise AssertionError("weaver tree construction requires a speculator with weaver weights")
        # `max_depth` counts the root; the weaver's `max_depth` counts edges
        if shape.max_tree_depth < 2 or shape.max_tree_depth > self.weaver.max_depth + 1:
            raise InvalidTreeShape(f"tree max_depth {shape.max_tree_depth} is outside 2..={self.weaver.max_depth + 1}")
        if shape.max_tree_depth > dflash_depth:
            raise InvalidTreeShape(f"tree of max_depth {shape.max_tree_depth} needs {shape.max_tree_depth - 1} draft rows, dflash depth is {dflash_depth}")
        draft_hidden, logits, _tokens = self.drafter.draft(target, int(target_output_token), dflash_depth, want_outputs=True) if _wants_outputs(self.drafter) else \
            self.drafter.draft(target, int(target_output_token), dflash_depth)
        depth_seeds = [prng.derive(root_position + depth) for depth in range(self.weaver.max_depth)]
        wshape = WeaverTreeShape(shape.tree_budget, shape.max_tree_depth, dflash_depth, shape.rounds, shape.expand_per_round, shape.expand_width)
        assert target_output_norm is not None, "the Weaver construction needs the target's output-norm row"
        out = self.weaver.encode_tree(target, target_output_norm, draft_hidden, logits, depth_seeds, int(target_output_token), wshape)
        if out is None:
            raise InvalidTreeShape("invalid Weaver tree input")  # WeaverEncodeError::InvalidTreeInput
        return build_trie(read_nodes(*out), root_position, prng, shape.tree_budget)


def _wants_outputs(drafter) -> bool:
    import inspect
    return "want_outputs" in inspect.signature(drafter.draft).parameters


class SpeculativeStream:
    """The speculative leg of LanguageModelStream: prefill with the target's hidden-feature taps feeding the drafter (stream.rs:299-318), then rounds of
    propose_tree -> one verify pass over the linearised tree -> FlatTrie::accept -> TransformerState / speculator encode_accept (stream.rs:380-470,551-628)."""

    def __init__(self, target, speculator: DFlashSpeculator, seed: int = 0, speculation_batch: int = 16, prefill_chunk: int = 1024, weaver_shape=None):
        """`weaver_shape` = (rounds, expand_per_round, expand_width): the Weaver construction of stream.rs:576-581 (16, 4, 4 there) when the speculator has a weaver"""
        self.target, self.speculator, self.prng = target, speculator, PRng(seed)
        self.weaver_shape = weaver_shape if weaver_shape is not None else ((16, 4, 4) if speculator.has_weaver() else None)
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
        self.output_norm = self.target.final_hidden_rows()[-1:] if self.speculator.has_weaver() else None  # the prefill's sampled row
        return int(token)

    def round(self) -> List[int]:
        """One speculation round; returns the tokens it emitted (>= 1: the token sampled at the last accepted node is always new)."""
        if self.speculator.has_weaver():  # stream.rs:567-590: tree_budget = the speculation batch, max_tree_depth 16 (capped by what the block / the weaver allow)
            r, e, wdt = self.weaver_shape
            depth = min(16, self.speculator.drafter.block_size, self.speculator.weaver.max_depth + 1)
            shape = TreeShape(tree_budget=self.speculation_batch, max_tree_depth=depth, construction_method="weaver", rounds=r, expand_per_round=e, expand_width=wdt)
        else:
            shape = TreeShape(tree_budget=min(self.speculation_batch, self.speculator.drafter.block_size), max_tree_depth=16)
        trie = self.speculator.propose_tree(self.target, self.tokens[-1], shape, self.prng, self.output_norm)
        flat = trie.linearize()
        self.tries.append(flat)
        sampled = self.target.verify_tree(flat.token_ids(), flat.nodes(), flat.token_seeds())
        full = flat.accept(sampled)
        indices = np.array([i for i, _, _ in full], dtype=np.uint32)
        feats = self.target.hidden_features()  # the tree pass's rows: taken BEFORE the accept moves the target on
        if self.speculator.has_weaver():  # the output-norm row of the last accepted node (stream.rs:466-476)
            self.output_norm = self.target.final_hidden_rows()[int(indices[-1]):int(indices[-1]) + 1]
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
