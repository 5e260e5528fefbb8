"""Experiment: is the run-to-run spread of the FAST reflect+refract kernel tied to the allocation
(physical placement) or to time (clocks)?  Three independent 8.3 GB layouts A, B, C of the same data,
timed round-robin.
usage: exp_placement.py [exact|fast] [pad floats] [layouts]"""
import sys
sys.path.insert(0, ".")
import torch
import rlshaders_amd as R

n = 1 << 26
ctx = R.Context(0)
ctx.set_math_mode(len(sys.argv) > 1 and sys.argv[1] == "fast")
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wo0, N0, T0 = R.gen_frame(ctx, 1234, 0, n)
u = lambda s, lo=0.0, hi=1.0: R.gen_uniform(ctx, 1234, 0, n, s, lo, hi)
src = [wo0[0], wo0[1], wo0[2], N0[0], N0[1], N0[2], T0[0], T0[1], T0[2], u(8), u(9), u(10), u(5, 0.05, 1.0),
       u(6, 1.05, 2.55), R.gen_aniso(ctx, 1234, 0, n), u(11), u(12), u(13), u(14)]

def layout():
    stride = n + pad
    big = torch.empty(31 * stride + 64, device="cuda", dtype=torch.float32)
    pl = [big[k * stride:k * stride + n] for k in range(31)]
    v3 = lambda k: big[k * stride:].as_strided((3, n), (stride, 1))
    for k, s in enumerate(src):
        pl[k].copy_(s)
    g = R.GgxSampler(ctx, v3(0), v3(3), v3(6), specColor=v3(9), ior=pl[13], roughness=pl[12], anisotropic=pl[14])
    out = (v3(19), v3(22), pl[25], pl[26], v3(27), pl[30])
    return big, (lambda: g.reflectRefract(pl[15], pl[16], pl[17], pl[18], out=out))

K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = [layout() for _ in range(K)]
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps
for rnd in range(3):
    print("round", rnd, " ".join(f"{chr(65 + k)} {t(L[k][1]):.3f}" for k in range(K)), flush=True)
