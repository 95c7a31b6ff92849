"""How many 16-vertex clusters would a wave of 256 Morton-sorted object points have to scan?  (CPU simulation)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvq_hip import synth
from dvq_hip.hand_parts import CONTACT_VERTS
from oracle import dvq_oracle as O

def bisect(ids, pts, g=16):
    if len(ids) <= g:
        return [ids]
    p = pts[ids]; ax = np.argmax(p.max(0) - p.min(0))
    o = ids[np.argsort(p[:, ax], kind="stable")]
    n = len(o); h = (n // 2 + g - 1) // g * g if n > 2 * g else (n + 1) // 2
    h = min(max(h, 1), n - 1)
    return bisect(o[:h], pts, g) + bisect(o[h:], pts, g)

def morton(p):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.int64)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)

B = 6
sd = synth.gennet_state_dict(0)
msd = synth.mano_state_dict(0)
sd.update({"rh_mano." + k: v for k, v in msd.items()})
obj = synth.object_clouds(B, 3000, 11)
g = torch.Generator().manual_seed(0)
out = O.generate_grasps(sd, obj, generator=g)
verts = out["vertices"].numpy() if isinstance(out, dict) else out[1].numpy()
vt = msd["v_template"].numpy()
cv = np.unique(np.array(CONTACT_VERTS)); rest = np.setdiff1d(np.arange(778), cv)
cl = bisect(cv, vt) + bisect(rest, vt)
print("clusters", len(cl), "sizes", sorted(set(len(c) for c in cl)))
for sort in (False, True):
    tot_scan = tot_all = 0
    for b in range(B):
        v = verts[b]; o = obj[b, :3].numpy().T
        if sort:
            o = o[np.argsort(morton(o), kind="stable")]
        lo = np.stack([v[c].min(0) for c in cl]); hi = np.stack([v[c].max(0) for c in cl])
        rep = np.stack([v[c[0]] for c in cl])
        for part in (slice(0, len(bisect(cv, vt))), slice(len(bisect(cv, vt)), len(cl))):
            d_rep = ((o[:, None, :] - rep[None, part, :]) ** 2).sum(-1)
            ub = d_rep.min(1)
            dbox = np.maximum(np.maximum(lo[None, part] - o[:, None], o[:, None] - hi[None, part]), 0)
            lb = (dbox ** 2).sum(-1)
            need = lb <= ub[:, None]                       # [N, clusters of the part]
            for w0 in range(0, 3000, 256):
                tot_scan += need[w0:w0 + 256].any(0).sum()
                tot_all += need.shape[1]
    print("sorted" if sort else "unsorted", "clusters scanned per wave: %.1f of %.1f per part pair -> fraction %.3f" % (
        tot_scan / (B * 12), tot_all / (B * 12), tot_scan / tot_all))
d = np.sqrt(((obj[0, :3].numpy().T[:, None] - verts[0][None]) ** 2).sum(-1).min(1))
print("nearest-vertex distance of points: min %.3f median %.3f max %.3f m" % (d.min(), np.median(d), d.max()))

print("--- hand moved onto the object (centroid 3 cm from a surface point) ---")
rng = np.random.default_rng(0)
nC = len(bisect(cv, vt))
for sort in (False, True):
    tot_scan = tot_all = 0
    for b in range(B):
        o = obj[b, :3].numpy().T
        p0 = o[rng.integers(0, 3000)]; c0 = o.mean(0); n = (p0 - c0) / np.linalg.norm(p0 - c0)
        v = verts[b] - verts[b].mean(0) + p0 + 0.03 * n
        if sort:
            o = o[np.argsort(morton(o), kind="stable")]
        lo = np.stack([v[c].min(0) for c in cl]); hi = np.stack([v[c].max(0) for c in cl])
        rep = np.stack([v[c[0]] for c in cl])
        for part in (slice(0, nC), slice(nC, len(cl))):
            d_rep = ((o[:, None, :] - rep[None, part, :]) ** 2).sum(-1)
            ub = d_rep.min(1)
            dbox = np.maximum(np.maximum(lo[None, part] - o[:, None], o[:, None] - hi[None, part]), 0)
            lb = (dbox ** 2).sum(-1)
            need = lb <= ub[:, None]
            for w0 in range(0, 3000, 256):
                tot_scan += need[w0:w0 + 256].any(0).sum()
                tot_all += need.shape[1]
    print("sorted" if sort else "unsorted", "fraction of clusters scanned per wave %.3f" % (tot_scan / tot_all))
    d = np.sqrt(((o[:, None] - v[None]) ** 2).sum(-1).min(1))
    print("   nearest distance: min %.4f median %.3f max %.3f; points within 2 cm: %d" % (d.min(), np.median(d), d.max(), (d < 0.02).sum()))
