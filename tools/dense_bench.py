"""Correctness (vs float64) and back-to-back timing of hg_dense_batch_f32 (csrc/dense.hip) against the library GEMM at the
shapes of the egnn_equihnns step.  Run on the GPU box:  python tools/dense_bench.py   (profiles/r02_dense_bench.txt)"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from equihgnn_amd import ops
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return torch.randn(*s, device=dev, generator=g)
def check(M, N, K, nk, bias=False, c=False, alpha=1.0):
    a = rnd(M, K); b = rnd(N, K) if nk else rnd(K, N)
    bi = rnd(N) if bias else None; cc = rnd(M, N) if c else None
    out = ops.dense(a, b, nk, bi, cc, alpha)
    ref = alpha * (a.double() @ (b.double().t() if nk else b.double()))
    if bias: ref = ref + bi.double()
    if c: ref = ref + cc.double()
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    print(f"check M={M} N={N} K={K} nk={nk} bias={bias} c={c}: rel err {err:.2e}", flush=True)
    assert err < 2e-6 * max(1.0, (K / 256) ** 0.5) * 2, err
for args in [(4736, 256, 256, True), (4736, 256, 256, False), (100, 64, 36, True, True, True, 0.5), (33, 68, 272, False, True),
             (4736, 512, 272, True, True), (4736, 2176, 256, True, True), (4736, 256, 2176, False), (1, 4, 4, True)]:
    check(*args)
# segment prologue
M, S, K, N = 4800, 4736, 256, 256
nnz = 10000
key = torch.randint(0, M, (nnz,), device=dev, generator=g); col = torch.randint(0, S, (nnz,), device=dev, generator=g)
csr = ops.csr_build(key, col, M)
src = rnd(S, K); w = rnd(N, K); bias = rnd(N)
out, a_out = ops.dense(src, w, True, bias, seg=(csr.rowptr, csr.col, None, True, M), a_out=True)
agg = ops._segment_reduce(src, csr.col, csr.rowptr, None, M, True)
ref = agg.double() @ w.double().t() + bias.double()
print("segment: a_out err", float((a_out - agg).abs().max()), "out rel err", float((out.double() - ref).abs().max() / ref.abs().max()), flush=True)
# weighted (backward form)
csr_t = ops.csr_build(col, key, S)
dy = rnd(M, K)
out2, a2 = ops.dense(dy, w, False, seg=(csr_t.rowptr, csr_t.col, csr.rowptr, False, S), a_out=True)
agg2 = ops._segment_reduce(dy, csr_t.col, csr_t.rowptr, csr.rowptr, S, False)
ref2 = agg2.double() @ w.double()
print("segment weighted: a_out err", float((a2 - agg2).abs().max()), "out rel err", float((out2.double() - ref2).abs().max() / ref2.abs().max()), flush=True)
# relu_ln prologue
h = rnd(4736, 256); pb = rnd(256); ga = rnd(256); be = rnd(256)
out3, a3 = ops.dense(h, w, True, bias, ln=(pb, ga, be, 1e-5), a_out=True)
xn = torch.nn.functional.layer_norm(torch.relu(h.double() + pb.double()), (256,), ga.double(), be.double(), 1e-5)
ref3 = xn @ w.double().t() + bias.double()
print("relu_ln: a_out err", float((a3.double() - xn).abs().max()), "out rel err", float((out3.double() - ref3).abs().max() / ref3.abs().max()), flush=True)

busy = rnd(4096, 4096)
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(busy, busy); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M_, N_, K_, nk) in [(4736, 256, 256, True), (4736, 256, 256, False), (4864, 256, 256, True), (4736, 512, 256, True), (4736, 512, 272, True),
                      (4736, 256, 512, True), (4736, 2176, 256, True), (4736, 256, 2176, False), (9472, 256, 256, True), (31000, 256, 256, True)]:
    M, N, K = M_, N_, K_
    a = rnd(M, K); b = rnd(N, K) if nk else rnd(K, N); out = torch.empty(M, N, device=dev)
    t_lib = timeit(lambda: torch.mm(a, b.t() if nk else b, out=out))
    t_mine = timeit(lambda: ops.dense(a, b, nk, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d} nk={int(nk)}: library {t_lib:7.2f} us ({fl/t_lib/1e6:6.1f} TF)   dense {t_mine:7.2f} us ({fl/t_mine/1e6:6.1f} TF)", flush=True)
# batched: two problems in one launch
a = rnd(4736, 256); b1 = rnd(256, 256); b2 = rnd(256, 256)
t2 = timeit(lambda: ops.dense_batch([ops.DenseProblem(a, b1), ops.DenseProblem(a, b2)]))
print(f"two [4736x256]x[256x256] in one launch: {t2:.2f} us", flush=True)
t3 = timeit(lambda: ops.dense(src, w, True, bias, seg=(csr.rowptr, csr.col, None, True, 4800)))
t4 = timeit(lambda: ops.dense(h, w, True, bias, ln=(pb, ga, be, 1e-5)))
print(f"segment-prologue dense {t3:.2f} us; relu_ln-prologue dense {t4:.2f} us", flush=True)
