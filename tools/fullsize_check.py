"""Prefill-logit deviation of the full-size Qwen3.5-0.8B vs the CPU reference (run on a GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle.oracle as O
from uzu_amd import synthetic as S, backend as B
from uzu_amd.engine import HipModel
from helpers import f32
cfg = S.qwen35_0p8b(max_context_length=1024)
bundle = S.build_model(cfg)
prompt = S.synthetic_prompt(int(sys.argv[1]) if len(sys.argv) > 1 else 128, cfg.vocab_size)
om = O.OracleModel(bundle)
o_tok, o_logits = om.prefill(prompt, True)
hm = HipModel(B.Context.new(), bundle, 0)
h_tok = hm.prefill(prompt)
w, g = f32(o_logits).astype(np.float64), f32(hm.read_logits()).astype(np.float64)
d = np.abs(w - g)
srt = np.sort(w)
print(f"tokens {o_tok} {h_tok}  sigma {w.std():.4f}  max|d|/sigma {d.max() / w.std():.4f}  mean|d|/sigma {d.mean() / w.std():.5f}  top2 gap/sigma {(srt[-1]-srt[-2]) / w.std():.4f}")
idx = np.argsort(-d)[:12]
for i in idx:
    print(f"  vocab {i}: want {w[i]:.4f} got {g[i]:.4f} |d|/sigma {d[i]/w.std():.3f} |d|/|w| {d[i]/max(abs(w[i]),1e-9):.4f}")
print("fraction with |d| > 0.25 sigma:", float((d > 0.25 * w.std()).mean()), " > 0.25 sigma + 0.05|w|:", float((d > 0.25 * w.std() + 0.05 * np.abs(w)).mean()))
print("abs max logit / sigma:", np.abs(w).max() / w.std())
rm = S.readout_row_multipliers(cfg)
wn, gn = w / rm, g / rm
dn = np.abs(wn - gn)
print(f"row-normalised: sigma {wn.std():.4f} max|d|/sigma {dn.max() / wn.std():.4f} mean {dn.mean() / wn.std():.5f}")
