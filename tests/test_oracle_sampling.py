"""CPU tests of the oracle's sampler: Philox4x32-10 against the published Random123 known-answer vectors, the
reference's own unit test of the uniform mapping, and the full UnifiedSampling kernel (grammar bitmask, temperature,
top-k / top-p / min-p, Gumbel-max) against independent NumPy re-derivations and distributional checks.

References: BU/encodable_block/sampling/gumbel.rs:1-81 (+ tests/unit/encodable_block/sampling/gumbel_test.rs),
BU/cpu/kernel/sampling/unified_sampling.rs:13-99.  The HIP path implements the greedy specialisation only (DESIGN.md
scope row a11); this file pins the oracle for the stochastic specialisations ahead of the kernel.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import bf16, f32
from oracle import oracle as O


def philox(ctr, key):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    O.lib().orc_philox4x32_10(c, k, o)
    return list(o)


def test_philox4x32_10_random123_known_answers():
    """kat_vectors of the Random123 distribution (Salmon, Moraes, Dror, Shaw, SC'11), philox4x32 with 10 rounds."""
    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_unit_interval_stays_open_like_the_reference_test():
    """gumbel_test.rs::unit_interval_stays_open: the extremes map to 2^-24 and 1 - 2^-24, never to 0 or 1."""
    lib = O.lib()
    lib.orc_unit_interval.restype = C.c_float
    assert lib.orc_unit_interval(C.c_uint32(0xFFFFFFFF)) == 1.0 - 2.0 ** -24
    assert lib.orc_unit_interval(C.c_uint32(0)) == 2.0 ** -24
    assert lib.orc_unit_interval(C.c_uint32(255)) == 2.0 ** -24
    assert lib.orc_unit_interval(C.c_uint32(0x80000000)) == 0.5


def uniform(key, offset, word):
    lib = O.lib()
    lib.orc_uniform_float.restype = C.c_float
    return lib.orc_uniform_float(C.c_uint64(key), C.c_uint32(offset), C.c_uint32(word))


def test_uniform_float_is_the_selected_philox_word():
    key = 0x0123456789ABCDEF
    for offset in (0, 1, 77, 2 ** 31):
        words = philox([offset, 0, 0, 0], [key & 0xFFFFFFFF, key >> 32])
        for w in range(4):
            assert uniform(key, offset, w) == np.float32(max(words[w] >> 8, 1)) * np.float32(2.0 ** -24)


@pytest.mark.parametrize("vocab", [1, 1000, 4096, 5000, 248320])
def test_revidx_gives_every_logit_its_own_counter_word(vocab):
    """(offset, word) pairs are distinct over the vocabulary, so no two logits share a random number (gumbel.rs:66-81)."""
    lib = O.lib()
    off, word = C.c_uint32(), C.c_uint32()
    seen = set()
    step = 1 if vocab <= 5000 else 7  # the big vocabulary: every 7th index plus both ends keeps the test fast
    idx = sorted(set(range(0, vocab, step)) | {vocab - 1})
    for i in idx:
        lib.orc_revidx(C.c_uint32(i), C.c_uint32(vocab), C.byref(off), C.byref(word))
        per_thread = -(-vocab // 4096)
        assert word.value == (i // 1024) % 4 and off.value == per_thread * (i % 1024) + (i // 1024) // 4
        seen.add((off.value, word.value))
    assert len(seen) == len(idx)


def sample(logits_bits, seeds=None, bitmask=None, temperature=None, top_k=None, top_p=None, min_p=None):
    batch, vocab = logits_bits.shape
    out = np.zeros(batch, np.uint32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint64) if seeds is not None else None
    bm = np.ascontiguousarray(bitmask, dtype=np.uint32) if bitmask is not None else None
    O.lib().orc_unified_sampling(
        C.c_void_p(logits_bits.ctypes.data), O.BF16, C.c_void_p(out.ctypes.data),
        C.c_void_p(sd.ctypes.data) if sd is not None else None, C.c_void_p(bm.ctypes.data) if bm is not None else None,
        int(temperature is not None), C.c_float(temperature or 0.0), int(top_k is not None), C.c_uint32(top_k or 0),
        int(top_p is not None), C.c_float(top_p or 0.0), int(min_p is not None), C.c_float(min_p or 0.0), vocab, batch)
    return out


def test_greedy_specialisation_ties_bitmask_and_filters():
    rng = np.random.default_rng(0)
    logits = bf16(rng.normal(0, 3, (6, 70)))
    logits[1, 5] = logits[1, 40] = bf16(np.array([50.0], np.float32))[0]  # tie: the lower index wins
    want = np.array([int(np.argmax(f32(r))) for r in logits], np.uint32)
    assert want[1] == 5
    assert np.array_equal(sample(logits), want)
    # greedy is invariant under temperature and under every filter (the maximum always survives the cuts)
    assert np.array_equal(sample(logits, temperature=0.7, top_k=3, top_p=0.5, min_p=0.2), want)
    # grammar bitmask: a cleared bit removes the token; [batch, ceil(vocab / 32)] words
    mask = np.full((6, 3), 0xFFFFFFFF, np.uint32)
    for b, t in enumerate(want):
        mask[b, t // 32] &= ~np.uint32(1 << (t % 32))
    got = sample(logits, bitmask=mask)
    second = np.array([int(np.argsort(-f32(r).astype(np.float64), kind="stable")[1]) for r in logits], np.uint32)
    second[1] = 40  # the tie partner is next
    assert np.array_equal(got, second)


def test_stochastic_top_k_1_is_argmax_and_seeds_are_independent():
    rng = np.random.default_rng(1)
    row = bf16(rng.normal(0, 2, (1, 300)))
    logits = np.repeat(row, 64, axis=0)
    seeds = rng.integers(0, 2 ** 63, 64, dtype=np.uint64)
    assert np.all(sample(logits, seeds=seeds, top_k=1) == np.argmax(f32(row[0])))
    a, b = sample(logits, seeds=seeds, temperature=1.0), sample(logits, seeds=seeds, temperature=1.0)
    assert np.array_equal(a, b) and len(set(a.tolist())) > 5  # deterministic per seed, different across seeds
    same = sample(logits, seeds=np.full(64, seeds[0]), temperature=1.0)
    assert len(set(same.tolist())) == 1 and same[0] == a[0]  # a row's draw depends on its seed only


def kept_set(logits_f32, temperature, top_k, top_p, min_p):
    """float64 re-derivation of the one-pass cut (unified_sampling.rs:56-77)."""
    x = logits_f32.astype(np.float64) / (temperature or 1.0)
    order = sorted(range(x.size), key=lambda i: (-x[i], i))
    p = np.exp(x - x[order[0]])
    p /= p.sum()
    kept, mass = [], 0.0
    for rank, i in enumerate(order):
        if (top_k is not None and rank >= top_k) or (top_p is not None and mass >= top_p) or (min_p is not None and x[i] < x[order[0]] + np.log(min_p)):
            break
        kept.append(i)
        mass += p[i]
    return kept, p


@pytest.mark.parametrize("temperature,top_k,top_p,min_p", [(None, 5, None, None), (0.8, None, 0.7, None), (1.3, None, None, 0.08), (0.9, 6, 0.9, 0.02)])
def test_filters_keep_exactly_the_rederived_set_and_sample_it_proportionally(temperature, top_k, top_p, min_p):
    rng = np.random.default_rng(2)
    vocab, draws = 24, 6000
    row = bf16(rng.normal(0, 1.5, (1, vocab)))
    kept, p = kept_set(f32(row[0]), temperature, top_k, top_p, min_p)
    seeds = rng.integers(0, 2 ** 63, draws, dtype=np.uint64)
    got = sample(np.repeat(row, draws, axis=0), seeds=seeds, temperature=temperature, top_k=top_k, top_p=top_p, min_p=min_p)
    counts = np.bincount(got, minlength=vocab)
    assert set(np.nonzero(counts)[0]) <= set(kept), "a token outside the kept set was sampled"
    q = p[kept] / p[kept].sum()  # Gumbel-max over the kept logits = softmax restricted to them
    for i, qi in zip(kept, q):
        sigma = np.sqrt(draws * qi * (1 - qi))
        assert abs(counts[i] - draws * qi) <= 5 * sigma + 1, (i, counts[i], draws * qi)


def test_gumbel_noise_has_the_right_moments():
    lib = O.lib()
    lib.orc_gumbel_float.restype = C.c_float
    g = np.array([lib.orc_gumbel_float(C.c_uint64(12345), C.c_uint32(o), C.c_uint32(w)) for o in range(5000) for w in range(4)], np.float64)
    assert abs(g.mean() - 0.5772156649) < 0.03 and abs(g.var() - np.pi ** 2 / 6) < 0.08 and np.isfinite(g).all()


def test_sampling_matches_committed_golden():
    """tests/golden/sampling.json (make_sampling_golden.py): tokens for fixed logits / seeds / filter settings."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_sampling_golden as G
    gold = json.load(open(os.path.join(here, "golden", "sampling.json")))
    assert (gold["vocab"], gold["batch"]) == (G.VOCAB, G.BATCH)
    logits, seeds, mask = G.inputs()
    for case in gold["cases"]:
        assert G.draw(case, logits, seeds, mask) == case["tokens"], case["name"]
    drawn = {c["name"]: c["tokens"] for c in gold["cases"]}
    assert drawn["stochastic"] != drawn["temperature"] and len(set(drawn["stochastic"])) > 6  # the cases really differ
