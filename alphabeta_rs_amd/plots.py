"""The reference's two plots (src/plot.rs:1-137) as functions over the files the CLIs write:
`bootstrap.png` from raw.npy (boot_model::run, src/boot_model.rs:105-109) and `metaplot.png` from results.txt
(src/cli/metaprofile.rs:113).  Host-side post-processing, matplotlib (Agg); no part of the hot path.

Same content as the reference's plotters output — 1280 x 960 pixels, the same titles, series, colours and axis ranges —
not the same pixels (a different rasteriser)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

SIZE_PX = (640 * 2, 480 * 2)   # src/plot.rs:9,91


def _figure():
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    fig, ax = plt.subplots(figsize=(SIZE_PX[0] / 100, SIZE_PX[1] / 100), dpi=100)
    return plt, fig, ax


def bootstrap(raw, output_dir) -> Path:
    """src/plot.rs:84-137: box plots of the bootstrap alphas (red) and betas (blue), y from 0 to 1.3 x the largest
    value.  raw: (n_boot, 7) rows [alpha, beta, weight, intercept, PrMM, PrUM, PrUU] (RawAnalysis) or a raw.npy path."""
    if isinstance(raw, (str, Path)):
        raw = np.load(raw)
    raw = np.asarray(raw, dtype=np.float64)
    if raw.ndim == 3:                      # metaprofile's (iterations, 7, windows): all windows pooled
        raw = np.moveaxis(raw, 1, 2).reshape(-1, 7)
    alphas, betas = raw[:, 0], raw[:, 1]
    top = max(0.0, float(np.nanmax(alphas)), float(np.nanmax(betas)))          # fold(0.0, max), :87-91
    plt, fig, ax = _figure()
    bp = ax.boxplot([alphas[np.isfinite(alphas)], betas[np.isfinite(betas)]], positions=[0, 1], widths=0.3,
                    whis=1.5, showfliers=False)
    for i, colour in enumerate(("red", "blue")):
        for part in ("boxes", "medians"):
            bp[part][i].set_color(colour)
        for part in ("whiskers", "caps"):
            for ln in bp[part][2 * i:2 * i + 2]:
                ln.set_color(colour)
    ax.set_xticks([0, 1], ["Alpha", "Beta"])
    ax.set_ylim(0.0, top * 1.3 if top > 0 else 1.0)
    ax.set_ylabel("Epimutation rate", fontsize=20)
    ax.set_title("Bootstrap Boxplot", fontsize=26)
    ax.grid(True, alpha=0.3)
    out = Path(output_dir) / "bootstrap.png"
    fig.savefig(out)
    plt.close(fig)
    return out


def read_results(path):
    """results.txt of the metaprofile driver (src/cli/metaprofile.rs:74-99): ';'-separated, one line per window"""
    lines = Path(path).read_text().splitlines()
    head = lines[0].split(";")
    rows = [dict(zip(head, ln.split(";"))) for ln in lines[1:] if ln.strip()]
    num = ("alpha", "beta", "ci_alpha_0.025", "ci_alpha_0.975", "ci_beta_0.025", "ci_beta_0.975")
    return {k: np.array([float(r[k]) for r in rows]) for k in num}


def metaplot(results, output_dir) -> Path:
    """src/plot.rs:6-82: alpha (red) and beta (blue) per window with their 95 % bootstrap intervals as bands,
    x 0..300, y 0..0.01, caption "Metaplot".  results: a results.txt path or the dict of read_results."""
    if isinstance(results, (str, Path)):
        results = read_results(results)
    x = np.arange(len(results["alpha"]), dtype=np.float32)
    plt, fig, ax = _figure()
    ax.plot(x, results["alpha"], color="red", label="Alpha")
    ax.plot(x, results["beta"], color="blue", label="Beta")
    ax.fill_between(x, results["ci_alpha_0.025"], results["ci_alpha_0.975"], color="red", alpha=0.2)
    ax.fill_between(x, results["ci_beta_0.025"], results["ci_beta_0.975"], color="blue", alpha=0.2)
    ax.set_xlim(0.0, 300.0)
    ax.set_ylim(0.0, 0.01)
    ax.set_title("Metaplot", fontsize=32)
    ax.legend(loc="upper right", framealpha=0.8, edgecolor="black")
    ax.grid(True, alpha=0.3)
    out = Path(output_dir) / "metaplot.png"
    fig.savefig(out)
    plt.close(fig)
    return out
