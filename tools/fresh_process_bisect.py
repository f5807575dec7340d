"""Production vs reference-order mode over a ladder of model variants, each the FIRST model(s) of a fresh process (this found the null-stream fill race of round 5: profiles/r5c_memset_race_before.txt / _after.txt).  gpurun: python tools/fresh_process_bisect.py"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import f32
from uzu_amd import _ffi, synthetic as S, desc as D
from uzu_amd.backend import Context
from uzu_amd.engine import HipModel

def set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact; fn.restype, fn.argtypes = None, [C.c_int32]; fn(1 if on else 0)

local = D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=256, max_sequence_length=8192, base=10000.0)
glob = D.RopeConfig(kind=D.ROPE_LINEAR, head_dim=256, max_sequence_length=8192, base=1000000.0, scaling_factor=8.0)
BASE = dict(name="g", vocab_size=4096, model_dim=2048, hidden_dim=8192, layer_kinds=[D.MIXER_ATTENTION] * 3, num_heads=8, num_groups=2, head_dim=256, rope=local,
            layer_ropes=None, rope_pattern=None, sliding_windows=[0], kv_sharing=None, ple_dim=0, group_size=128, max_context_length=1024, embedding_norm=False,
            post_norms=False, post_layer_scalars=False, normalize_values=False, qk_norm=True, seed=91)
V = {
 "plain": {},
 "post_norms": dict(post_norms=True),
 "post_norms+scalars": dict(post_norms=True, post_layer_scalars=True),
 "windows512": dict(sliding_windows=[512, 512, 0]),
 "kv_sharing_full": dict(kv_sharing={2: 0}),
 "kv_sharing_ring": dict(sliding_windows=[512, 0, 512], kv_sharing={2: 0}),
 "ple": dict(ple_dim=256),
 "embedding_norm": dict(embedding_norm=True),
 "normalize_values": dict(normalize_values=True),
 "layer_ropes": dict(layer_ropes=[local, glob], rope_pattern=[0, 1, 0]),
 "d1024": dict(model_dim=1024, hidden_dim=4096),
 "d1024+post_norms": dict(model_dim=1024, hidden_dim=4096, post_norms=True),
}
ctx = Context.new(0)
t0 = time.time()
for name, kw in V.items():
    cfg = S.tiny_gemma(**{**BASE, **kw})
    bundle = S.build_model(cfg)
    for plen in (70, 700):
        prompt = S.synthetic_prompt(plen, cfg.vocab_size)
        out = {}
        for exact in (1, 0):
            set_exact(bool(exact))
            hm = HipModel(ctx, bundle)
            tok = hm.prefill(prompt)
            out[exact] = (tok, hm.read_logits())
            hm.close()
        set_exact(False)
        w, g = f32(out[1][1]).astype(np.float64), f32(out[0][1]).astype(np.float64)
        print(f"{name:22s} prompt {plen:4d}: tok exact {out[1][0]:5d} prod {out[0][0]:5d}  err {np.abs(w - g).max() / w.std():8.3f} sigma  nan_prod {int(np.isnan(g).sum())} nan_exact {int(np.isnan(w).sum())}  [{time.time() - t0:.1f}s]", flush=True)
