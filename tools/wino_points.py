"""Development aid (CPU, torch): Winograd F(m x m, 3x3) transform matrices from a set of interpolation points
(Toom-Cook: A^T = evaluation of the data polynomial, G = evaluation of the filter, B^T = transposed inverse Vandermonde;
rows rescaled to integers) and the float32 error of one 256 -> 256 layer against a float64 convolution for several point sets.
The set 0, 1, -1, 1/2, -2, inf of csrc/spa_wino.hip comes out of this comparison.   python tools/wino_points.py"""
import numpy as np, torch, torch.nn.functional as F
from fractions import Fraction as Fr
def matrices(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    P = [Fr(p) for p in pts]
    V = [[p**k for k in range(n)] for p in P] + [[Fr(0)]*(n-1) + [Fr(1)]]
    Ed = [[p**k for k in range(m)] for p in P] + [[Fr(0)]*(m-1) + [Fr(1)]]
    Eg = [[p**k for k in range(r)] for p in P] + [[Fr(0)]*(r-1) + [Fr(1)]]
    # inverse of V with fractions
    A = [row[:] + [Fr(int(i==j)) for j in range(n)] for i,row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c,n) if A[i][c] != 0); A[c],A[piv] = A[piv],A[c]
        pv = A[c][c]; A[c] = [v/pv for v in A[c]]
        for i in range(n):
            if i!=c and A[i][c]!=0:
                f=A[i][c]; A[i]=[a-f*b for a,b in zip(A[i],A[c])]
    Vinv = [row[n:] for row in A]
    Bt = [[Vinv[k][j] for k in range(n)] for j in range(n)]      # (V^-1)^T
    At = [[Ed[j][i] for j in range(n)] for i in range(m)]
    G = Eg
    # scale: make Bt rows integer-ish: multiply row j of Bt by s_j and divide row j of G by s_j
    for j in range(n):
        den = np.lcm.reduce([f.denominator for f in Bt[j]])
        s = Fr(int(den))
        Bt[j] = [v*s for v in Bt[j]]; G[j] = [v/s for v in G[j]]
    f = lambda M: np.array([[float(v) for v in row] for row in M])
    return f(At), f(G), f(Bt)
def check(m, r, pts, C=256, K=256, dt=torch.float32):
    At, G, Bt = matrices(m, r, pts)
    n = m + r - 1
    # exactness in float64 (1-D)
    rs = np.random.RandomState(0); d = rs.randn(n); g = rs.randn(r)
    y = At @ ((G @ g) * (Bt @ d)); ref = np.array([sum(d[i+k]*g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref), (y, ref)
    torch.manual_seed(0)
    H, W = 12*m, 12*m
    x = torch.relu(torch.randn(1, C, H, W)); w = torch.randn(K, C, 3, 3) * (2.0/(9*C))**0.5
    Att, Gt, Btt = (torch.tensor(M, dtype=dt) for M in (At, G, Bt))
    U = torch.einsum('ij,kcjl,ml->imkc', torch.tensor(G), w.double(), torch.tensor(G)).to(dt)
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, n, m).unfold(3, n, m)
    Vv = torch.einsum('ij,bcyxjl,ml->imbcyx', Btt, t, Btt)
    M = torch.einsum('imkc,imbcyx->imbkyx', U, Vv)
    Y = torch.einsum('ij,jlbkyx,ml->bkyxim', Att, M, Att)
    Y = Y.permute(0,1,2,4,3,5).reshape(1, K, H, W)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    d32 = F.conv2d(x, w, None, 1, 1)
    s = ref.abs().max().item()
    print('F(%dx%d,3x3) pts %s: err %.2e of scale (direct fp32 %.2e); max|Bt| %.1f max|G| %.2f max|At| %.1f' % (m, m, pts, (Y.double()-ref).abs().max().item()/s, (d32.double()-ref).abs().max().item()/s, np.abs(Bt).max(), np.abs(G).max(), np.abs(At).max()))
if __name__ == '__main__':
    check(2, 3, [0, 1, -1])
    for pts in ([0,1,-1,2], [0,1,-1,Fr(1,2)], [0,1,-1,-2], [0,Fr(1,2),-Fr(1,2),1], [0,1,-1,Fr(-1,2)]):
        check(3, 3, pts)
    check(4, 3, [0,1,-1,2,-2])
    check(4, 3, [0,1,-1,Fr(1,2),-Fr(1,2)])
    check(4, 3, [0,1,-1,Fr(1,2),-2])
