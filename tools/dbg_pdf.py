import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import cases, oracle_lib as O, rlshaders_amd as R
from gpu_util import dev, host, ggx_oracle, ggx_sampler
ctx = R.Context(0)
n = 1 << 16
c = cases.ggx_mixed(99, n); x = cases.xi(99, n, 4)
og = ggx_oracle(O, c)
wi, f_ref, pdf_ref, _ = og.sample_eval_pdf(x[0], x[1])
pdf_dec = og.pdf(wi)
print('oracle fused vs oracle decoupled pdf mismatch:', (pdf_dec != pdf_ref).mean())
s = ggx_sampler(ctx, c)
pdf = host(s.evalPdf(dev(wi)))
bad = np.nonzero(pdf != pdf_ref)[0]
print('gpu mismatch', len(bad), bad[:20], np.diff(bad)[:20])
print('gpu', pdf[bad[:8]], 'ref', pdf_ref[bad[:8]], 'dec', pdf_dec[bad[:8]])
print('mismatch vs oracle decoupled', (pdf != pdf_dec).mean())
