"""Speculation trie on the host (crates/backend-uzu/src/trie.rs, encodable_block/batch_topology.rs).

A speculator proposes a TREE of continuations; one forward pass scores every node (attention under the trie mask, DeltaNet layers
through tree-verify), ``FlatTrie.accept`` walks the sampled tokens from the root and returns the accepted root path, and the
sequence state takes exactly that path (``HipModel.accept``).  Same names and meaning as the reference:

  TrieNode(token, seed, logprob).add / get / linearize   trie.rs:25-170
  TrieNode.prune_to_budget / flat                       trie.rs:93-156
  PRng(seed).derive(index)                              encodable_block/sampling/prng.rs (the per-position sampling seeds of the nodes)
  FlatTrie.token_ids / nodes / accept                   trie.rs:186-305  (nodes = token_subtrie_ranges: {start, end, height})
  parents(nodes)                                        batch_topology.rs:11-37
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


_M64 = (1 << 64) - 1


class PRng:
    """encodable_block/sampling/prng.rs: the 64-bit finaliser (MurmurHash3's fmix64) of seed + index -- the seed of the draw at a position
    (stream.rs:248-258) and of a trie node (trie.rs:147-151).  The device derives the same numbers (csrc/k_sampling.hip::derive_seed)."""

    def __init__(self, seed: int):
        self.seed = int(seed) & _M64

    def derive(self, index: int) -> int:
        h = (self.seed + int(index)) & _M64
        h ^= h >> 33
        h = (h * 0xff51afd7ed558ccd) & _M64
        h ^= h >> 33
        h = (h * 0xc4ceb9fe1a85ec53) & _M64
        h ^= h >> 33
        return h


class DuplicateTokenId(ValueError):
    """trie.rs:12-16 TrieError::DuplicateTokenId"""


class TrieNode:
    def __init__(self, token: int, seed: int = 0, logprob: float = 0.0):
        self.token, self.seed, self.logprob = int(token), int(seed), float(logprob)
        self.next: List["TrieNode"] = []

    def add(self, node: "TrieNode") -> int:
        if any(n.token == node.token for n in self.next):
            raise DuplicateTokenId(f"child with token id {node.token} is already present")
        self.next.append(node)
        return len(self.next) - 1

    def get(self, token: int) -> Optional["TrieNode"]:
        for n in self.next:
            if n.token == token:
                return n
        return None

    def node_count(self) -> int:
        return 1 + sum(n.node_count() for n in self.next)

    def prune_to_budget(self, budget: int):
        """Keep the `budget` nodes with the largest cumulative log-probabilities (trie.rs:93-138): a stable descending sort of the DFS order
        (ties keep the earlier node, i.e. a parent before its child), then every dropped node goes with its whole subtree."""
        assert budget > 0, "budget must keep at least the root"
        logprobs: List[float] = []

        def collect(node: "TrieNode", parent: np.float32):
            lp = np.float32(parent + np.float32(node.logprob))
            logprobs.append(lp)
            for child in node.next:
                collect(child, lp)

        collect(self, np.float32(0.0))
        if budget >= len(logprobs):
            return
        order = sorted(range(len(logprobs)), key=lambda i: -float(logprobs[i]))  # (stable, like slice::sort_by; total_cmp: no NaNs here)
        kept = [False] * len(logprobs)
        for i in order[:budget]:
            kept[i] = True
        cursor = [0]

        def prune(node: "TrieNode"):
            cursor[0] += 1
            children = []
            for child in node.next:
                index = cursor[0]
                prune(child)
                if kept[index]:
                    children.append(child)
            node.next = children

        prune(self)

    @classmethod
    def flat(cls, tokens: Sequence[int], prefix_length: int = 0, prng: Optional[PRng] = None) -> "TrieNode":
        """A chain (trie.rs:140-156): node i carries prng.derive(prefix_length + i); without a prng the seeds stay 0 (greedy verification
        draws none)."""
        assert len(tokens) > 0, "need seed node"
        seed = (lambda i: prng.derive(prefix_length + i)) if prng is not None else (lambda i: 0)
        root = cls(tokens[0], seed(0))
        leaf = root
        for i, t in enumerate(tokens[1:], start=1):
            leaf.add(cls(t, seed(i)))
            leaf = leaf.next[0]
        return root

    def linearize(self) -> "FlatTrie":
        """DFS pre-order; a node's subtree is the contiguous index range [start, end] (trie.rs:154-172)."""
        nodes: List[TrieNode] = []
        ranges: List[List[int]] = []
        heights: List[int] = []

        def walk(node: "TrieNode", height: int):
            index = len(nodes)
            nodes.append(node)
            ranges.append([index, index])
            heights.append(height)
            for child in node.next:
                walk(child, height + 1)
            ranges[index][1] = len(nodes) - 1

        walk(self, 0)
        return FlatTrie(nodes, ranges, heights)


class FlatTrie:
    def __init__(self, nodes: List[TrieNode], ranges: List[List[int]], heights: List[int]):
        self._nodes, self._ranges, self._heights = nodes, ranges, heights

    def __len__(self) -> int:
        return len(self._nodes)

    def index(self, node: TrieNode) -> Optional[int]:
        """Position of THIS node object in the DFS order (trie.rs:262-267: pointer identity, not equality of the fields)."""
        return next((i for i, n in enumerate(self._nodes) if n is node), None)

    def token_ids(self) -> np.ndarray:
        return np.array([n.token for n in self._nodes], dtype=np.uint32)

    def token_seeds(self) -> np.ndarray:
        return np.array([n.seed for n in self._nodes], dtype=np.uint64)

    def nodes(self) -> np.ndarray:
        """token_subtrie_ranges: uint32 [n, 3] = {trie_start, trie_end, height} (gpu_types/trie.rs)."""
        return np.array([[r[0], r[1], h] for r, h in zip(self._ranges, self._heights)], dtype=np.uint32).reshape(len(self), 3)

    def parents(self) -> np.ndarray:
        return parents(self.nodes())

    def is_flat(self) -> bool:
        return all(h == i for i, h in enumerate(self._heights))

    def accept(self, sampled_tokens: Sequence[int]) -> List[Tuple[int, int, int]]:
        """-> [(node index, the node's input token, the token sampled at that node)] along the accepted root path: follow the
        sampled token while the tree proposed it (trie.rs:271-305, without the grammar leg)."""
        current = self._nodes[0]
        accepted = []
        while True:
            index = next(i for i, n in enumerate(self._nodes) if n is current)
            sampled = int(sampled_tokens[index])
            accepted.append((index, current.token, sampled))
            nxt = current.get(sampled)
            if nxt is None:
                return accepted
            current = nxt


def parents(nodes: np.ndarray) -> np.ndarray:
    """BatchTopology::new (batch_topology.rs:11-37): parent index per node from the heights of a DFS-ordered trie (-1 = the root,
    whose parent is the accepted context)."""
    nodes = np.asarray(nodes, dtype=np.uint32).reshape(-1, 3)
    out = np.empty(len(nodes), dtype=np.int32)
    stack: List[int] = []
    for index, (_, _, height) in enumerate(nodes):
        assert height <= len(stack), f"trie node {index} at height {height} has no parent in DFS order"
        del stack[int(height):]
        out[index] = stack[-1] if stack else -1
        stack.append(index)
    return out
