"""Pair structure of the memory-only time of a tile pass, from round 2's probes (profiles/r02/geom_probe3_sets.csv: 2400 random
9-bit sets in ascending order; geom_probe4_orders.csv: 3000 random sets in random order; geom_probe5_pairs.csv: every bit pair
in every role) — host only, no GPU.  A ridge regression of the pass time on "bits a and b are both tile bits" and "... are both
walked by the lanes of a wave" (the first three high tile bits), plus one term per bit and role.  Prints the cross-validated R^2,
the heaviest pairs, the lane-pair weights by bit distance, and how well the fit (made on round 2's kernel) predicts the few-block
passes of round 4 (profiles/r04/pass_model_n30_forms.csv).  Usage: python tools/geom_pairs.py"""
import csv, itertools, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def load(fn):
    out = []
    for line in open(os.path.join(ROOT, "r02", fn)):
        p = line.strip().split(",")
        out.append(([int(x) for x in p[0].split()], float(p[1])))
    return out
rows = load("geom_probe3_sets.csv") + load("geom_probe4_orders.csv") + load("geom_probe5_pairs.csv")
y = np.array([ms for _, ms in rows])
bits = list(range(3, 30))
pairs = list(itertools.combinations(bits, 2))
pidx = {p: i for i, p in enumerate(pairs)}
NP = len(pairs)
def featurize(order):
    f = np.zeros(2 * NP + 81)
    for a, b in itertools.combinations(sorted(order), 2): f[pidx[(a, b)]] += 1          # both in the tile
    for a, b in itertools.combinations(sorted(order[0:3]), 2): f[NP + pidx[(a, b)]] += 1  # both walked by the lanes of a wave
    for r in range(3):
        for b in order[3 * r:3 * r + 3]: f[2 * NP + 27 * r + (b - 3)] += 1
    return f
X = np.array([featurize(o) for o, _ in rows])
def fit(Xt, yt, lam=10.0):
    mu, ym = Xt.mean(0), yt.mean()
    A = Xt - mu
    return mu, ym, np.linalg.solve(A.T @ A + lam * np.eye(A.shape[1]), A.T @ (yt - ym))
perm = np.random.default_rng(0).permutation(len(y))
r2 = []
for f in range(5):
    te = perm[f::5]; tr = np.setdiff1d(perm, te)
    mu, ym, c = fit(X[tr], y[tr])
    r2.append(1 - ((y[te] - ((X[te] - mu) @ c + ym)) ** 2).mean() / y[te].var())
print(f"{len(y)} passes, {y.mean():.2f} +- {y.std():.2f} ms; 5-fold cross-validated R^2 of the pair model: {np.mean(r2):.3f}")
mu, ym, c = fit(X, y)
for name, off in (("both tile bits", 0), ("both lane bits", NP)):
    top = np.argsort(-c[off:off + NP])[:8]
    print(f"heaviest pairs, {name}: " + ", ".join(f"{pairs[i]} {c[off + i]:+.2f}" for i in top))
    low = np.argsort(c[off:off + NP])[:4]
    print(f"lightest pairs, {name}: " + ", ".join(f"{pairs[i]} {c[off + i]:+.2f}" for i in low))
print("lane-pair weight by distance d (index bits a, a + d; a = 3 ...):")
for d in (1, 2, 6, 7, 8, 14):
    print(f"  d = {d:2d}: " + " ".join(f"{c[NP + pidx[(a, a + d)]]:+.2f}" for a in range(3, 30 - d)))
seen = {}
for a in csv.DictReader(open(os.path.join(ROOT, "r04", "pass_model_n30_forms.csv"))):
    if float(a["visited"]) != 1.0: continue
    nb = len(bytes.fromhex(a["forms"]))
    hm = int(a["high_mask"], 16)
    seen.setdefault((tuple(b for b in range(30) if hm >> b & 1), nb), []).append(float(a["ms"]))
G = np.array([(featurize(list(o)) - mu) @ c + ym for (o, nb) in seen])
Y = np.array([np.mean(v) for v in seen.values()])
NB = np.array([nb for (_, nb) in seen])
for lim in (3, 4, 99):
    s = NB <= lim
    A0 = np.stack([np.ones(s.sum()), np.maximum(0, NB[s] - 4)], 1)
    A1 = np.stack([np.ones(s.sum()), np.maximum(0, NB[s] - 4), G[s]], 1)
    r0 = Y[s] - A0 @ np.linalg.lstsq(A0, Y[s], rcond=None)[0]
    r1 = Y[s] - A1 @ np.linalg.lstsq(A1, Y[s], rcond=None)[0]
    print(f"round-4 passes with <= {lim} blocks ({s.sum()}): rms {np.sqrt((r0 ** 2).mean()):.3f} ms by blocks alone, {np.sqrt((r1 ** 2).mean()):.3f} with the pair model's prediction, correlation {np.corrcoef(G[s], Y[s])[0, 1]:.2f}")
