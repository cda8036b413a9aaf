"""Image comparison shared by the GPU parity tests.  Every pixel enters the RMSE (none is excluded); pixels outside the
pixel-wise tolerance are counted AND bounded in size."""
import numpy as np


def compare_full(img, ref, frac_bad=0.005, rmse_bar=1e-3):
    """img, ref: radiance sums (h, w, 3).  Returns the relative RMSE over all pixels.
    Pixel-wise: rel 2e-3 / abs 2e-4 (transcendentals differ by ULPs between ocml, fp32, and libm, double narrowed); at most
    frac_bad of the pixels may lie outside it (a path whose discrete decision — lobe choice, Fresnel coin, an edge shadow
    ray — flips on such a last-bit difference), none of them by more than the brightest value in the reference image."""
    assert np.isfinite(img).all()
    diff = np.abs(img - ref)
    bad = ~np.isclose(img, ref, rtol=2e-3, atol=2e-4).all(axis=2)
    rel = float(np.sqrt(np.mean((img - ref) ** 2)) / (np.sqrt(np.mean(ref ** 2)) + 1e-12))
    assert bad.mean() <= frac_bad, f"{bad.mean():.4%} of pixels differ"
    assert diff.max() <= float(ref.max()), f"an outlier differs by {diff.max():.3g} (image max {ref.max():.3g})"
    assert rel < rmse_bar, f"relative rmse over all pixels {rel:.3g}"
    return rel
