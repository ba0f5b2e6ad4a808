"""Debug aid: per-candidate predictions/costs of one 8x8 block from k_search2's test tap, against a
pure-python restatement of the quarter-pel interpolation.  The tap returns each candidate's 8x8 prediction
column-major (8 bytes per column).  usage: python scripts/dbg_search2.py [block]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import Oracle
from pipeline import default_segments
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence
W, H = 256, 128
s = SynthSequence(W, H, seed=7, noise=20, saturate=True)
f = [s.frame(t) for t in range(4)]
hip = api.Vp8Hip(W, H, 0.97); ora = Oracle(W, H, 0.97)
for be in (hip, ora):
    be.set_segments(default_segments()); be.upload_last(*f[2]); be.upload_current(*f[3]); be.inter_transform(1, 1, 0, 0)
n2o = ora.net(0, 2)
SIX = [[0,0,128,0,0,0],[0,-6,123,12,-1,0],[2,-11,108,36,-8,1],[0,-9,93,50,-6,0],[3,-16,77,77,-16,3],[0,-6,50,93,-9,0],[1,-8,36,108,-11,2],[0,-1,12,123,-6,0]]
ref = f[2][0].astype(np.int64); cur = f[3][0].astype(np.int64)
def pix(x, y): return ref[min(max(y, 0), H - 1), min(max(x, 0), W - 1)]
def tdiv(s): return int(s / 128) if s >= 0 else -int((-s) / 128)
def interp8(qx, qy):
    fx, fy = (qx % 4) * 2, (qy % 4) * 2; ix, iy = qx // 4, qy // 4
    Hh = np.zeros((13, 8), np.int64)
    for L in range(13):
        for c in range(8):
            sm = 64 + sum(pix(ix + c - 2 + t, iy - 2 + L) * SIX[fx][t] for t in range(6)); Hh[L, c] = min(max(tdiv(sm), 0), 255)
    out = np.zeros((8, 8), np.int64)
    for i in range(8):
        for c in range(8):
            sm = 64 + sum(Hh[i + t, c] * SIX[fy][t] for t in range(6)); out[i, c] = min(max(tdiv(sm), 0), 255)
    return out
b = int(sys.argv[1]) if len(sys.argv) > 1 else 170
bw = W // 8; cx, cy = (b % bw) * 8, (b // bw) * 8
raw = np.zeros(1024, np.uint32); buf = raw[:26 * 18].reshape(26, 18)
rc = hip.lib.vp8hip_debug_search2_block(hip.h, 0, b, C.c_void_p(raw.ctypes.data)); print("rc", rc)
v0 = n2o[b].astype(int) * 4
tot = 0
for k in range(26):
    qx = 4 * cx + v0[0] + (k % 5 - 2); qy = 4 * cy + v0[1] + (k // 5 - 2)
    if k == 25: qx, qy = 4 * cx, 4 * cy
    o = interp8(qx, qy)
    hp = buf[k, :16].copy().view(np.uint8).reshape(8, 8).T.astype(np.int64)
    nd = int((o != hp).sum()); tot += nd
    print(k, (k % 5 - 2, k // 5 - 2), "cost", buf[k, 16], "valid", buf[k, 17], "pred mismatches", nd)
    if nd:
        print(o); print(hp)
Lx, Ly, o = [int(np.int32(x)) for x in raw[730:733]]
print("Lx,Ly,o", Lx, Ly, o, "B", cx + n2o[b][0], cy + n2o[b][1], "total mismatches", tot)
win = (raw[648:648 + 70].copy().view(np.uint8) ^ 0x80).reshape(14, 20)
ax = (Lx - 3) & ~3
exp = np.array([[pix(ax + j, Ly - 3 + r) for j in range(20)] for r in range(14)])
print("window mismatches", np.argwhere(exp != win).tolist()[:20])
HT = (raw[468:468 + 180].copy().view(np.uint8) ^ 0x80).reshape(5, 144)[:, :128].reshape(5, 8, 16)
PH = [4, 6, 0, 2, 4]; XO = [-1, -1, 0, 0, 0]
for xc in range(5):
    e = np.zeros((8, 14), np.int64)
    for r in range(14):
        for c in range(8):
            sm = 64 + sum(pix(Lx + XO[xc] + c - 2 + t, Ly - 3 + r) * SIX[PH[xc]][t] for t in range(6)); e[c, r] = min(max(tdiv(sm), 0), 255)
    bad = np.argwhere(e != HT[xc][:, :14])
    print("H xc", xc, "mismatches (col,row)", len(bad), bad.tolist()[:24])
    if len(bad):
        print(" exp col0", e[0].tolist()); print(" got col0", HT[xc][0].tolist())
