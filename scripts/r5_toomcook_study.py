"""CPU-only feasibility study (round-4 review item 2): would Toom-Cook F(2,5) - 6 instead of 10 channel-GEMMs per
output pair of the k = 5 / dilation-3 convs (nnlib/v2/layers.py:1217-1280) - stay inside the 1e-4 logit gate under the
split-f16 rounding model of ``conv_f16x3_kernel``?

Model of the arithmetic (the same one for the direct form and for Toom-Cook, so that the direct form's distance from f64
can be checked against what the GPU measures - 3.2e-5 on these weights, tests/test_gpu_robustness.py):
  * every conv input is what the F16S layout stores: hi = f16(x), lo = f16(x - hi)  (22 significant bits);
  * weights (direct: W_t; Toom-Cook: the transformed G.W, computed in f64 on the host) are scaled by a power of two,
    split into hi / lo f16 planes the same way;
  * a product is hi.hi + hi.lo + lo.hi accumulated in f32 (three f32 GEMMs on exactly representable products; the lo.lo
    term is dropped), the Toom-Cook input transform B^T d runs in f32 on hi + lo and its result is split again (what the
    MFMA operand path could hold), the output transform A^T runs in f32;
  * everything around the convs (norms, GELU, residual adds, pools, heads) is evaluated in f64, so that the figure
    printed is the convs' contribution alone.
The dilation-3 conv splits into three undilated phases x[3j + phase]; F(2,5) works on each phase.

Kill criterion: logits further than 5e-5 from the f64 evaluation on the stand-in weights.

  python scripts/r5_toomcook_study.py            # prints the table DESIGN.md 3.1 quotes
"""
from __future__ import annotations

import sys
from fractions import Fraction
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from conftest import load_model_cfg  # noqa: E402
from jaeger_amd.engine import frame_length  # noqa: E402
from oracle import encoder as oenc  # noqa: E402
from oracle import forward as ofwd  # noqa: E402

INF = "inf"


def toom_cook(points, m, r):
    """(A^T (m, n), G (n, r), B^T (n, n)) as exact fractions for F(m, r) on n = m + r - 1 points (the last one infinity):
    y = A^T [(G g) * (B^T d)],  y_j = sum_t g_t d_{j + t}.  Transposition of Toom-Cook's linear convolution: A^T and G are
    the evaluation matrices, B^T the transposed inverse of the n x n evaluation matrix."""
    n = m + r - 1
    assert len(points) == n and points[-1] == INF

    def ev(cols):
        rows = []
        for p in points:
            rows.append([Fraction(int(j == cols - 1)) for j in range(cols)] if p == INF else
                        [Fraction(p) ** j for j in range(cols)])
        return rows
    v = ev(n)
    # exact inverse by Gauss-Jordan
    aug = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(v)]
    for c in range(n):
        piv = next(i for i in range(c, n) if aug[i][c] != 0)
        aug[c], aug[piv] = aug[piv], aug[c]
        d = aug[c][c]
        aug[c] = [x / d for x in aug[c]]
        for i in range(n):
            if i != c and aug[i][c] != 0:
                f = aug[i][c]
                aug[i] = [x - f * y for x, y in zip(aug[i], aug[c])]
    inv = [row[n:] for row in aug]
    bt = [[inv[j][i] for j in range(n)] for i in range(n)]          # (V^-1)^T
    at = [[ev(m)[i][j] for i in range(n)] for j in range(m)]
    g = ev(r)
    # the usual normalisation: the denominators move from B^T into G (B^T integer where the points are integers)
    for i in range(n):
        den = 1
        for x in bt[i]:
            den = den * x.denominator // np.gcd(den, x.denominator)
        num = 0
        for x in bt[i]:
            num = np.gcd(num, abs(int(x * den)))
        s = Fraction(den, max(int(num), 1))
        bt[i] = [x * s for x in bt[i]]
        g[i] = [x / s for x in g[i]]
    to = lambda mat: np.array([[float(x) for x in row] for row in mat], np.float64)  # noqa: E731
    return to(at), to(g), to(bt)


def split16(x32: torch.Tensor):
    hi = x32.to(torch.float16)
    lo = (x32 - hi.to(torch.float32)).to(torch.float16)
    return hi.to(torch.float32), lo.to(torch.float32)


def split_weights(w64: torch.Tensor):
    """f64 weights -> (hi, lo, 1 / scale) with a power-of-two scale that puts the largest entry near 2^10 (the lo plane
    stays normal, as the kernel's pre-scale does)."""
    mx = float(w64.abs().max())
    e = 0 if mx == 0 else int(np.floor(np.log2(mx)))
    scale = 2.0 ** (10 - e)
    hi, lo = split16((w64 * scale).to(torch.float32))
    return hi, lo, 1.0 / scale


def mm3(xh, xl, wh, wl):
    """hi.hi + hi.lo + lo.hi in f32 (lo.lo dropped)."""
    return xh @ wh + (xh @ wl + xl @ wh)


STATS = {"convs": 0}


def make_conv(kind: str, tc=None):
    """A replacement for oracle.forward.conv1d_nwc: f64 in, f64 out, the k = 5 / 128 -> 128 convs through the rounding
    model (``direct`` or ``toomcook``), everything else exact."""
    exact = ofwd.conv1d_nwc

    def conv(x, kernel, stride, padding, dilation):
        k, cin, cout = kernel.shape
        if not (k == 5 and cin == 128 and cout == 128 and stride == 1):
            return exact(x, kernel, stride, padding, dilation)
        io_dtype = x.dtype
        STATS["convs"] += 1
        n, length, _ = x.shape
        if padding == "SAME":
            l_out, pl, pr = ofwd.same_pad(length, k, stride, dilation)
            x = torch.nn.functional.pad(x, (0, 0, pl, pr))
        else:
            l_out = length - dilation * (k - 1)
        xh, xl = split16(x.to(torch.float32))                     # what the F16S activation layout holds
        if kind == "direct":
            y = torch.zeros((n, l_out, cout), dtype=torch.float32)
            wh, wl, inv = split_weights(kernel.to(torch.float64))
            for t in range(k):
                s = t * dilation
                y += mm3(xh[:, s:s + l_out].reshape(-1, cin), xl[:, s:s + l_out].reshape(-1, cin), wh[t], wl[t]
                         ).reshape(n, l_out, cout)
            return (y * inv).to(io_dtype)
        at, g, bt = tc
        m, nn = at.shape
        gw = torch.einsum("it,tcd->icd", torch.from_numpy(g), kernel.to(torch.float64))      # (n, cin, cout), f64 on the host
        planes = [split_weights(gw[i]) for i in range(nn)]
        x32 = xh + xl
        y = torch.zeros((n, l_out, cout), dtype=torch.float32)
        bt32 = torch.from_numpy(bt).to(torch.float32)
        at32 = torch.from_numpy(at).to(torch.float32)
        for ph in range(dilation):
            xp = x32[:, ph::dilation]                              # the phase's undilated sequence
            lp_out = len(range(ph, l_out, dilation))
            tiles = -(-lp_out // m)
            need = tiles * m + k - 1
            if xp.shape[1] < need:
                xp = torch.nn.functional.pad(xp, (0, 0, 0, need - xp.shape[1]))
            # input tiles (n, tiles, nn, cin): positions m*i .. m*i + nn - 1
            idx = (torch.arange(tiles)[:, None] * m + torch.arange(nn)[None, :])
            d = xp[:, idx]                                         # (n, tiles, nn, cin)
            dt = torch.einsum("ij,btjc->btic", bt32, d)            # B^T d in f32
            out = torch.zeros((n, tiles, m, cout), dtype=torch.float32)
            prods = []
            for i in range(nn):
                dh, dl = split16(dt[:, :, i].reshape(-1, cin))
                wh, wl, inv = planes[i]
                prods.append(mm3(dh, dl, wh, wl).reshape(n, tiles, cout) * inv)
            p = torch.stack(prods, dim=2)                          # (n, tiles, nn, cout)
            out = torch.einsum("ji,btic->btjc", at32, p)           # A^T in f32
            y[:, ph::dilation] = out.reshape(n, tiles * m, cout)[:, :lp_out]
        return y.to(io_dtype)
    return conv


def main():
    cfg = load_model_cfg("brain")
    w = ofwd.random_weights(cfg, seed=38341)
    rng = np.random.Generator(np.random.PCG64(77))
    fsize, n_win = 1500, (int(sys.argv[1]) if len(sys.argv) > 1 else 8)                                        # the inputs of test_f16x3_scale_sweep_against_f64
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=fsize * n_win).copy()
    seq[700:760] = ord("N")
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    ids = oenc.encode_windows([seq[s:s + fsize].tobytes() for s in starts], fsize, pad_to=frame_length(fsize))
    exact = ofwd.conv1d_nwc
    ref64 = ofwd.forward(cfg, w, ids, dtype=torch.float64)
    ref32 = ofwd.forward(cfg, w, ids)
    print(f"logits: max |ref| = {np.abs(ref64['prediction']).max():.3f}; torch-CPU f32 vs f64: "
          f"{np.abs(ref32['prediction'] - ref64['prediction']).max():.2e}")
    half = Fraction(1, 2)
    variants = [("direct split-f16 (today's kernel, 10 GEMMs / output pair)", "direct", None),
                ("Toom-Cook F(2,5), points {0, +-1, +-2, inf}", "toomcook", toom_cook([0, 1, -1, 2, -2, INF], 2, 5)),
                ("Toom-Cook F(2,5), points {0, +-1, +-1/2, inf}", "toomcook", toom_cook([0, 1, -1, half, -half, INF], 2, 5)),
                ("Toom-Cook F(2,5), points {0, +-1/2, +-2, inf}", "toomcook", toom_cook([0, half, -half, 2, -2, INF], 2, 5))]
    # sanity: the transforms reproduce the correlation exactly in f64
    for name, kind, tc in variants[1:]:
        at, g, bt = tc
        gg, dd = rng.standard_normal(5), rng.standard_normal(6)
        want = np.array([np.dot(gg, dd[j:j + 5]) for j in range(2)])
        got = at @ ((g @ gg) * (bt @ dd))
        assert np.abs(got - want).max() < 1e-12, name
    for rest, dtype in (("f64 (the convs' contribution alone)", torch.float64),
                        ("f32 (as on the GPU: norms, GELU, adds, pools, heads in f32)", torch.float32)):
        print(f"\nrest of the network in {rest}")
        print("| conv arithmetic of the twelve k = 5 convs | logits vs f64 (max abs) | logits vs the f32 oracle | embedding vs f64 "
              "| nmd vs f64 | max row-sum |B^T| . |G| . |A^T| |")
        print("|---|---|---|---|---|---|")
        for name, kind, tc in variants:
            STATS["convs"] = 0
            ofwd.conv1d_nwc = make_conv(kind, tc)
            try:
                got = ofwd.forward(cfg, w, ids, dtype=dtype)
            finally:
                ofwd.conv1d_nwc = exact
            assert STATS["convs"] == 12, STATS
            amp = ""
            if tc is not None:
                at, g, bt = tc
                amp = f"{np.abs(bt).sum(1).max():.3g} . {np.abs(g).sum(1).max():.3g} . {np.abs(at).sum(1).max():.3g}"
            errs = [float(np.abs(np.asarray(got[k], np.float64) - np.asarray(ref64[k], np.float64)).max())
                    for k in ("prediction", "embedding", "nmd")]
            e32 = float(np.abs(np.asarray(got["prediction"], np.float64) - np.asarray(ref32["prediction"], np.float64)).max())
            print(f"| {name} | {errs[0]:.2e} | {e32:.2e} | {errs[1]:.2e} | {errs[2]:.2e} | {amp} |")


if __name__ == "__main__":
    main()
