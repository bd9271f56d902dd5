"""attention_step_fwd kernel alone, captured graph of 20 launches, for the four on/off combinations of its two dropouts."""
import sys
import torch
sys.path.insert(0, ".")
import masters_thesis_amd.ops as ops
be = ops.backend()
import os
B, R, D, A, U = int(os.environ.get("ATT_B", 64)), int(os.environ.get("ATT_R", 360)), 32, 32, 512
f = lambda *s: torch.randn(*s, device="cuda")
h, F, P, W2, b2, v, bv = f(B, U), f(B, R, D), f(B, R, A), f(U, A) * 0.05, f(A), f(A), f(1)
qpre, alpha, ctx, ctx_d = f(B, A), f(B, R), f(B, D), f(B, D)
step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")

def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for ra, ri, cd in ((0.2, 0.2, ctx_d), (0.2, 0.0, ctx_d), (0.0, 0.2, ctx_d), (0.0, 0.0, ctx_d), (0.2, 0.2, None)):
    t = timeit(lambda: be.attention_step_fwd(h, F, P, W2, b2, v, bv, qpre, alpha, ctx, cd, None, B, R, D, A, U, 0.2, ra, ri,
                                             D + 512, 42, 16, 48, 0, step_dev))
    print(f"rate_attn={ra} rate_in={ri} ctx_d={'yes' if cd is not None else 'no '}: {t:6.2f} us")

keep4 = torch.zeros(1, B * R * A // 4, dtype=torch.uint8, device="cuda")
be.dropout_mask4(keep4, B * R * A, 1, 0.2, 42, 16, 0, step_dev)
t = timeit(lambda: be.attention_step_fwd(h, F, P, W2, b2, v, bv, qpre, alpha, ctx, ctx_d, None, B, R, D, A, U, 0.2, 0.2, 0.2,
                                         D + 512, 42, 16, 48, 0, step_dev, keep4=keep4[0]))
print(f"rate_attn=0.2 rate_in=0.2 stored keep bits: {t:6.2f} us")
k15 = torch.zeros(15, B * R * A // 4, dtype=torch.uint8, device="cuda")
t = timeit(lambda: be.dropout_mask4(k15, B * R * A, 15, 0.2, 42, 16, 0, step_dev))
print(f"mask generation for 15 timesteps: {t:6.2f} us")

# ---- backward kernel
dP, dF, dvb, dqpre, dh = torch.zeros(B, R, A, device="cuda"), torch.zeros(B, R, D, device="cuda"), torch.zeros(B, A + 1, device="cuda"), f(B, A), f(B, U)
dz, Wc, dctx = f(B, U, 4) * 0.01, f(D, U, 4) * 0.05, f(B, D)
alpha_n = torch.softmax(f(B, R), dim=1)
for ra, ri, fused in ((0.2, 0.2, True), (0.2, 0.0, True), (0.0, 0.0, True), (0.2, 0.2, False), (0.0, 0.0, False)):
    t = timeit(lambda: be.attention_step_bwd(None if fused else dctx, F, P, W2, v, qpre, alpha_n, dP, dF, dvb, dqpre, dh, B, R, D, A, U,
                                             0.2, ra, ri, D + 512, 42, 16, 48, 0, step_dev, dz=dz if fused else None,
                                             Wc=Wc if fused else None))
    print(f"bwd rate_attn={ra} rate_in={ri} fused_dctx={fused}: {t:6.2f} us")
t = timeit(lambda: be.attention_step_bwd(dctx, F, P, W2, v, qpre, alpha_n, dP, dF, dvb, dqpre, dh, B, R, D, A, U, 0.2, 0.2, 0.2,
                                         D + 512, 42, 16, 48, 0, step_dev, keep4=keep4[0]))
print(f"bwd rate_attn=0.2 rate_in=0.2 fused_dctx=False stored keep bits: {t:6.2f} us")
# ---- LSTM step kernels as used by the attention model (ctx operand) and by the dense model
Ur, xz, c0 = f(U, U, 4) * 0.05, f(B, U, 4), f(B, U)
h2, c2, gates = f(B, U), f(B, U), f(B, U, 4)
Wc2 = f(D, U, 4) * 0.05
for use_ctx in (True, False):
    t = timeit(lambda: be.lstm_step_fwd(xz, h, c0, Ur, ctx if use_ctx else None, Wc2 if use_ctx else None, D if use_ctx else 0,
                                        None, 0, 0, None, h2, c2, None, gates, B, U))
    print(f"lstm fwd ctx={use_ctx}: {t:6.2f} us")
