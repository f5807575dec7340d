"""The round-4 prefill paths against the paths they replace, on ONE library build (the switches are read once per process, so every leg is its own
process running tools/ab_prefill_bits.py: a 700-token prefill + 8 greedy tokens of a DeltaNet + attention model at model_dim 1024):

  * four-channel conv apply (k_deltanet.hip::conv_apply4_kernel)                  -- BIT-IDENTICAL logits to the one-channel kernel
  * split-K reduction + Normalization in one launch (normalization_from_partials)  -- BIT-IDENTICAL logits, fewer launches
  * DeltaNet scan as two concurrent segments (k_deltanet_chunk.hip: ScanSplit)     -- another summation order: logits within the parity tolerance
  * DeltaNetPrefillPrep inside the chunk preparation (round 6: dn_chunk_prep_kernel<true>) -- BIT-IDENTICAL logits to the two launches, fewer launches
  * the head norms inside the AttentionPrepare launch, SigmoidGate inside the key-split merge (round 6)  -- BIT-IDENTICAL logits, fewer launches
  * the conv out of place without its halo launch (round 6: conv_apply4_oop_kernel)          -- BIT-IDENTICAL logits (and the same carried conv state: the decode steps
                                                                                              behind the prefill produce the same tokens)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _leg(tmp_path, name, **env):
    dump = str(tmp_path / f"{name}.npy")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_prefill_bits.py"), "--model", "tiny", "--prompt", "700", "--dump", dump],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    rec["logits"] = np.load(dump)
    return rec


def test_prefill_switches_bit_identity_and_tolerance(tmp_path):
    base = _leg(tmp_path, "default")
    conv1 = _leg(tmp_path, "conv1", UZU_HIP_TUNE="conv_apply4=0")
    two = _leg(tmp_path, "two_launches", UZU_HIP_TUNE="norm_partials=0")
    chain = _leg(tmp_path, "one_chain", UZU_HIP_TUNE="dn_split=0")
    prep = _leg(tmp_path, "prep_launch", UZU_HIP_TUNE="prep_fused=0")  # DeltaNetPrefillPrep as its own launch in front of the chunk preparation
    assert prep["logits_sha256"] == base["logits_sha256"] and prep["tokens"] == base["tokens"]
    assert prep["prefill_launches"] > base["prefill_launches"], (prep["prefill_launches"], base["prefill_launches"])
    attn = _leg(tmp_path, "attn_separate", UZU_HIP_TUNE="attn_fused=0")  # QKVNorm launches in front of AttentionPrepare, SigmoidGate behind the attention merge
    assert attn["logits_sha256"] == base["logits_sha256"] and attn["tokens"] == base["tokens"]
    assert attn["prefill_launches"] > base["prefill_launches"], (attn["prefill_launches"], base["prefill_launches"])
    inplace = _leg(tmp_path, "conv_in_place", UZU_HIP_TUNE="conv_oop=0")  # the conv in place behind its halo launch instead of out of place
    assert inplace["logits_sha256"] == base["logits_sha256"] and inplace["tokens"] == base["tokens"]
    assert conv1["logits_sha256"] == base["logits_sha256"] and conv1["tokens"] == base["tokens"]
    assert two["logits_sha256"] == base["logits_sha256"] and two["tokens"] == base["tokens"]
    assert two["prefill_launches"] > base["prefill_launches"], (two["prefill_launches"], base["prefill_launches"])  # the fused path really ran
    # the split scan: f32 sums in another order in front of a bf16 rounding, then three more layers: the logits stay within 0.05 sigma of the one-chain scan
    a, b = f32(base["logits"]).astype(np.float64), f32(chain["logits"]).astype(np.float64)
    assert np.abs(a - b).max() <= 0.05 * b.std(), (np.abs(a - b).max(), b.std())
    assert chain["first_token"] == base["first_token"]
