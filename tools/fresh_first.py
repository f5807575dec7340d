"""first thing in a fresh process: a short prompt + chained decode right after model creation, production first, then reference-order mode"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import f32
from uzu_amd import _ffi, synthetic as S, desc as D
from uzu_amd.backend import Context
from uzu_amd.engine import HipModel

def set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact; fn.restype, fn.argtypes = None, [C.c_int32]; fn(1 if on else 0)

which = sys.argv[1]
cfg = {"qwen": lambda: S.qwen35_0p8b(max_context_length=256), "llama2": lambda: S.llama3_8b(max_context_length=256, layer_kinds=[D.MIXER_ATTENTION] * 2, vocab_size=32000),
       "q14": lambda: S.qwen3_14b_class(max_context_length=256, layer_kinds=[D.MIXER_ATTENTION] * 2, vocab_size=32000)}[which]()
plen = int(sys.argv[2])
bundle = S.build_model(cfg)
prompt = S.synthetic_prompt(plen, cfg.vocab_size)
ctx = Context.new(0)
out = {}
for exact in (0, 1):
    set_exact(bool(exact))
    hm = HipModel(ctx, bundle)
    tok = hm.prefill(prompt)
    rows = [(tok, hm.read_logits())]
    for _ in range(3):
        if exact:
            hm.set_next_token(out[0][len(rows) - 1][0])
        tok = int(hm.decode(1)[0][0])
        rows.append((tok, hm.read_logits()))
    out[exact] = rows
    hm.close()
set_exact(False)
for i, ((pt, pl), (et, el)) in enumerate(zip(out[0], out[1])):
    w, g = f32(el).astype(np.float64), f32(pl).astype(np.float64)
    print(f"{which} prompt {plen} step {i}: prod {pt} exact {et} err {np.abs(w - g).max() / w.std():.3f} sigma zeros_prod {int((g == 0).sum())}", flush=True)
