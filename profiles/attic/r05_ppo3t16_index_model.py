"""numpy model of the 16-column tile (ppo3T16): v_mfma_f32_16x16x32 lane images + every index map of the design, checked
against plain matrix formulas on integer data"""
import numpy as np
rng = np.random.default_rng(0)
H = 128

def mfma16(A, B, C):
    """A, B: [64 lanes][8], C: [64][4] -> D [64][4].  A lane (m = l & 15, kb = l >> 4): A[m][8 kb + i]; B lane (n, kb): B[8 kb + i][n];
    D lane (n = l & 15, kb) reg q: D[4 kb + q][n]"""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        c, kb = l & 15, l >> 4
        Am[c, 8 * kb:8 * kb + 8] = A[l]
        Bm[8 * kb:8 * kb + 8, c] = B[l]
    Dm = Am @ Bm
    D = C.copy()
    for l in range(64):
        c, kb = l & 15, l >> 4
        for q in range(4):
            D[l, q] += Dm[4 * kb + q, c]
    return D

def pi(k):  # k-position (32 s2 + 8 kb + i) -> unit j carried there by the D registers of form b
    s2, r = k >> 5, k & 31
    kb, i = r >> 3, r & 7
    return 16 * (2 * s2 + (i >> 2)) + 4 * kb + (i & 3)

assert sorted(pi(k) for k in range(H)) == list(range(H))
W2 = rng.integers(-3, 4, (H, H)).astype(np.float64)        # W2[j][u]
NW = 8
h1 = rng.integers(0, 4, (NW * 16, H)).astype(np.float64)    # h1[s][u] of a 128-sample tile
dz2_true = rng.integers(-2, 3, (NW * 16, H)).astype(np.float64)

def F16(ks, t):  # lane (c, kb): W2[16 t + c][32 ks + 8 kb + i]
    return np.array([[W2[16 * t + (l & 15), 32 * ks + 8 * (l >> 4) + i] for i in range(8)] for l in range(64)])
def G16(s2, tu):  # lane (c, kb): W2[pi(32 s2 + 8 kb + i)][16 tu + c]
    return np.array([[W2[pi(32 * s2 + 8 * (l >> 4) + i), 16 * tu + (l & 15)] for i in range(8)] for l in range(64)])

XH = np.zeros((8, 4, 64, 8)); XD = np.zeros((8, 4, 64, 8))  # [tu or consumer wave][K][lane][8]
dh1_all = {}
for v in range(NW):  # producer wave v: samples 16 v + c
    hx = [np.array([[h1[16 * v + (l & 15), 32 * ks + 8 * (l >> 4) + i] for i in range(8)] for l in range(64)]) for ks in range(4)]
    aa = [np.zeros((64, 4)) for _ in range(8)]; ab = [np.zeros((64, 4)) for _ in range(8)]
    for ks in range(4):
        for t in range(8):
            fr = F16(ks, t)
            aa[t] = mfma16(hx[ks], fr, aa[t])   # form a: [sample][unit], lanes = j
            ab[t] = mfma16(fr, hx[ks], ab[t])   # form b: [unit][sample], lanes = s
    z2 = h1[16 * v:16 * v + 16] @ W2.T         # z2[s][j]
    for l in range(64):
        c, kb = l & 15, l >> 4
        for t in range(8):
            for q in range(4):
                assert aa[t][l, q] == z2[4 * kb + q, 16 * t + c]
                assert ab[t][l, q] == z2[c, 16 * t + 4 * kb + q]
    # transposition: hy[tu] = mfma16(A = hx[tu >> 1], B = idf[tu & 1])
    hy = []
    for tu in range(8):
        hh = tu & 1
        idf = np.array([[1.0 if 8 * (l >> 4) + i - 16 * hh == (l & 15) else 0.0 for i in range(8)] for l in range(64)])
        d = mfma16(hx[tu >> 1], idf, np.zeros((64, 4)))
        for l in range(64):
            c, kb = l & 15, l >> 4
            for q in range(4):
                assert d[l, q] == h1[16 * v + 4 * kb + q, 16 * tu + c]
        hy.append(d)
        for l in range(64):  # slab write: fragment (tu, K = v >> 1), lane' = (c, 2 (v & 1) + (kb >> 1)), half kb & 1
            c, kb = l & 15, l >> 4
            lp = c + 16 * (2 * (v & 1) + (kb >> 1))
            XH[tu, v >> 1, lp, 4 * (kb & 1):4 * (kb & 1) + 4] = d[l]
    # dZ2b fragments straight from the form-b registers (here: the true dz2 in form-b layout)
    dzb = []
    for s2 in range(4):
        fr = np.zeros((64, 8))
        for l in range(64):
            c, kb = l & 15, l >> 4
            for i in range(8):
                t, q = 2 * s2 + (i >> 2), i & 3
                fr[l, i] = dz2_true[16 * v + c, 16 * t + 4 * kb + q]   # = register (t, q) of lane (c, kb)
        dzb.append(fr)
    dh1 = [np.zeros((64, 4)) for _ in range(8)]
    for s2 in range(4):
        for tu in range(8):
            dh1[tu] = mfma16(dzb[s2], G16(s2, tu), dh1[tu])
    dH1 = dz2_true[16 * v:16 * v + 16] @ W2    # dH1[s][u] = sum_j dz2[s][j] W2[j][u]
    for l in range(64):
        c, kb = l & 15, l >> 4
        for tu in range(8):
            for q in range(4):
                assert dh1[tu][l, q] == dH1[4 * kb + q, 16 * tu + c]
    # dZ2a (form-a layout: lane (c, kb) reg (t, q) = dz2[s = 16 v + 4 kb + q][j = 16 t + c]) into the consumers' slab
    for t in range(8):
        for l in range(64):
            c, kb = l & 15, l >> 4
            lp = c + 16 * (2 * (v & 1) + (kb >> 1))
            XD[t, v >> 1, lp, 4 * (kb & 1):4 * (kb & 1) + 4] = [dz2_true[16 * v + 4 * kb + q, 16 * t + c] for q in range(4)]
# consumers: wave w owns the columns j = 16 w + c
dW2_true = dz2_true.T @ h1  # dW2[j][u] = sum_s dz2[s][j] h1[s][u]
for w in range(NW):
    acc = [np.zeros((64, 4)) for _ in range(8)]
    for K in range(4):
        for tu in range(8):
            acc[tu] = mfma16(XH[tu, K], XD[w, K], acc[tu])
    for l in range(64):
        c, kb = l & 15, l >> 4
        for tu in range(8):
            for q in range(4):
                assert acc[tu][l, q] == dW2_true[16 * w + c, 16 * tu + 4 * kb + q], (w, l, tu, q)
print("all index maps of the 16-column tile check out")
