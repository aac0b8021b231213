"""Statistics of the dropout hash rand_quad (egot2_amd/csrc/common.h), restated in numpy: per-lane uniformity (mean, variance,
keep rates, chi-square of 8-bit buckets), lane / adjacent-row / adjacent-column correlations, joint keep patterns of the four lanes."""
import numpy as np
M32 = np.uint64(0xffffffff)
def rand_quad(key, row, cq):
    k0 = np.uint64(key & 0xffffffff); k1 = np.uint64(key >> 32)
    row = row.astype(np.uint64); cq = cq.astype(np.uint64)
    x = ((row * np.uint64(0x9E3779B1) + k1) & M32) ^ ((cq * np.uint64(0x85EBCA77) + k0) & M32)
    x ^= x >> np.uint64(16)
    p = x * np.uint64(0x7feb352d)
    y = (p & M32) ^ (p >> np.uint64(32))
    y ^= y >> np.uint64(15)
    z = (y * np.uint64(0x846ca68b)) & M32
    z ^= z >> np.uint64(16)
    return z & M32, y & M32
def lanes(w0, w1):
    return [w0 & np.uint64(0xffff), w0 >> np.uint64(16), w1 & np.uint64(0xffff), w1 >> np.uint64(16)]
rng = np.random.default_rng(0)
for trial in range(3):
    key = int(rng.integers(0, 2**63)) | 1
    rows = np.arange(256 * 64, dtype=np.uint64)[:, None]
    cqs = np.arange(512, dtype=np.uint64)[None, :]
    w0, w1 = rand_quad(key, np.broadcast_to(rows, (rows.shape[0], 512)), np.broadcast_to(cqs, (rows.shape[0], 512)))
    L = [l.astype(np.float64) / 65536.0 for l in lanes(w0, w1)]
    print("key", hex(key))
    for i, l in enumerate(L):
        print("  lane", i, "mean %.5f var %.5f (1/12=%.5f)" % (l.mean(), l.var(), 1/12), "keep(p=.5) %.5f keep(p=.1) %.5f" % ((l >= 0.5).mean(), (l >= 0.1).mean()))
    # correlations between lanes, adjacent rows, adjacent col quads
    C = np.corrcoef(np.stack([l.ravel() for l in L]))
    print("  max |lane corr|", np.abs(C - np.eye(4)).max())
    for l in L[:2]:
        print("  adj-row corr %.5f adj-col corr %.5f" % (np.corrcoef(l[:-1].ravel(), l[1:].ravel())[0, 1], np.corrcoef(l[:, :-1].ravel(), l[:, 1:].ravel())[0, 1]))
    # bit balance of masks per row and per column at p = 0.5
    m = (L[0] >= 0.5)
    print("  per-row keep std %.5f (binomial %.5f)  per-col keep std %.5f (binomial %.5f)" % (m.mean(1).std(), 0.5 / np.sqrt(512), m.mean(0).std(), 0.5 / np.sqrt(m.shape[0])))
    # chi-square of 8-bit buckets
    for i, l in enumerate(lanes(w0, w1)):
        h = np.bincount((l >> np.uint64(8)).ravel().astype(np.int64), minlength=256)
        e = h.sum() / 256
        print("   lane", i, "chi2/255 = %.3f" % (((h - e) ** 2 / e).sum() / 255), end="")
    print()

# joint independence of mask bits at p = 0.5 across the 4 lanes (16 patterns)
key = 0x1234567887654321 | 1
rows = np.arange(256 * 64, dtype=np.uint64)[:, None]; cqs = np.arange(512, dtype=np.uint64)[None, :]
w0, w1 = rand_quad(key, np.broadcast_to(rows, (rows.shape[0], 512)), np.broadcast_to(cqs, (rows.shape[0], 512)))
L = lanes(w0, w1)
for thr in (0x8000, 6554):
    code = sum(((l >= np.uint64(thr)).astype(np.int64) << i) for i, l in enumerate(L))
    h = np.bincount(code.ravel(), minlength=16) / code.size
    pk = 1 - thr / 65536
    exp = np.array([np.prod([pk if (c >> i) & 1 else 1 - pk for i in range(4)]) for c in range(16)])
    print("thr", thr, "max rel dev of the 16 joint patterns", np.abs(h / exp - 1).max(), "n", code.size)
# small-counter structure: rows 0..47 x cq 0..511 for many clips (row = clip*64 + tok)
