"""CPU emulation of split-precision GEMM schemes on a 4-layer SIREN forward and on gradient-shaped GEMMs: relative error
against fp64 of plain fp32, bf16 x {1,3,4,6,9} products and fp16 x {3,4} products with power-of-two tensor scales.
This is the experiment behind csrc/gemm_h3.inc (run: python tools/split_emul.py)."""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import inr_oracle as O
def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split(a, n):
    parts=[]; r = a.astype(np.float32)
    for _ in range(n):
        p = bf16(r); parts.append(p); r = (r - p).astype(np.float32)
    return parts
def mm_split(a, b, n, terms):
    A = split(a, n); Bp = split(b, n)
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    # small terms first
    for (i,j) in sorted(terms, key=lambda t:-(t[0]+t[1])):
        acc = acc + (A[i] @ Bp[j])   # fp32 sgemm accumulate
    return acc
rng = np.random.default_rng(0)
N=4096
x = (rng.standard_normal((N,256))).astype(np.float32); x = np.concatenate([np.sin(x[:, :128]), np.cos(x[:, :128])],1).astype(np.float32)
# siren-like weights
W0 = rng.uniform(-1/256,1/256,(512,256)).astype(np.float32)
W1 = rng.uniform(-np.sqrt(6/512)/30,np.sqrt(6/512)/30,(512,512)).astype(np.float32)
def fwd(mm):
    h = np.sin(30*mm(x, W0.T)).astype(np.float32)
    for _ in range(3):
        h = np.sin(30*mm(h, W1.T)).astype(np.float32)
    return h
ref = np.sin(30*(x.astype(np.float64)@W0.T.astype(np.float64)))
for _ in range(3): ref = np.sin(30*(ref@W1.T.astype(np.float64)))
rel = lambda a: np.linalg.norm(a-ref)/np.linalg.norm(ref)
print("fp32           ", rel(fwd(lambda a,b: a@b)))
x3 = [(0,0),(0,1),(1,0)]
x4 = x3+[(1,1)]
x6 = [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]
x9 = [(i,j) for i in range(3) for j in range(3)]
print("bf16x1         ", rel(fwd(lambda a,b: mm_split(a,b,1,[(0,0)]))))
print("bf16x3 (2 pc)  ", rel(fwd(lambda a,b: mm_split(a,b,2,x3))))
print("bf16x4 (2 pc)  ", rel(fwd(lambda a,b: mm_split(a,b,2,x4))))
print("bf16x6 (3 pc)  ", rel(fwd(lambda a,b: mm_split(a,b,3,x6))))
print("bf16x9 (3 pc)  ", rel(fwd(lambda a,b: mm_split(a,b,3,x9))))

print("---- fp16 two-piece, three products, power-of-two tensor scales, RTZ high part")
def rtz16(x):
    # round-toward-zero fp32 -> fp16 (normal + subnormal), emulate: use float16 RNE then fix if magnitude grew
    h = x.astype(np.float16).astype(np.float32)
    grew = np.abs(h) > np.abs(x)
    h16 = x.astype(np.float16)
    dn = np.nextafter(h16, np.float16(0)).astype(np.float32)
    return np.where(grew, dn, h).astype(np.float32)
def scale_for(a):
    m = np.abs(a).max()
    return np.float32(2.0 ** (14 - np.floor(np.log2(m)))) if m > 0 else np.float32(1)
def mm_f16x3(a, b, four=False):
    sa, sb = scale_for(a), scale_for(b)
    A = (a * sa).astype(np.float32); Bm = (b * sb).astype(np.float32)
    a0 = rtz16(A); a1 = (A - a0).astype(np.float16).astype(np.float32)
    b0 = rtz16(Bm); b1 = (Bm - b0).astype(np.float16).astype(np.float32)
    acc = (a0 @ b1) + (a1 @ b0)
    if four: acc = acc + a1 @ b1
    acc = acc + a0 @ b0
    return (acc / (sa * sb)).astype(np.float32)
print("fp16x3         ", rel(fwd(mm_f16x3)))
print("fp16x4         ", rel(fwd(lambda a,b: mm_f16x3(a,b,True))))
# gradient-like GEMM: dW = dz^T x with heavy-tailed dz
dz = (rng.standard_normal((N,512)) * np.exp(rng.standard_normal((N,1))*3) * 1e-7).astype(np.float32)
refg = dz.astype(np.float64).T @ x.astype(np.float64)
relg = lambda a: np.linalg.norm(a-refg)/np.linalg.norm(refg)
print("dW fp32        ", relg(dz.T @ x))
print("dW fp16x3      ", relg(mm_f16x3(np.ascontiguousarray(dz.T), x)))
print("dW bf16x6      ", relg(mm_split(np.ascontiguousarray(dz.T), x, 3, x6)))
# input-grad-like: dz @ W
refi = dz.astype(np.float64) @ W1.astype(np.float64)
reli = lambda a: np.linalg.norm(a-refi)/np.linalg.norm(refi)
print("dX fp32        ", reli(dz @ W1))
print("dX fp16x3      ", reli(mm_f16x3(dz, W1)))
print("dX bf16x6      ", reli(mm_split(dz, W1, 3, x6)))
