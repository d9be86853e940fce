"""the reference's two PNGs (src/plot.rs) from the files the CLIs write.  CPU: the inputs come
from the oracle; the GPU CLI tests call the same scripts on the CLIs' real outputs."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def png_size(path):
    import struct

    b = Path(path).read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    return struct.unpack(">II", b[16:24])


def test_bootstrap_png_from_raw_npy(oracle, abn, golden, tmp_path):
    ped, p0 = golden["generated"], golden["p0uu_generated"]
    s0 = abn.gen_start_simplices(3, 0, 4, ped[:, 3].max())
    fits = oracle.fit_batch(ped, p0, p0, 1.0, s0, 2000, lanes=8)
    k, model, pred, resid, _ = oracle.select_best(ped, p0, fits["best"])
    raw, _ = oracle.boot_model(ped, model, pred, resid, p0, p0, 1.0, 3, 0, 0, 40, lanes=8)
    np.save(tmp_path / "raw.npy", raw)
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "plot_bootstrap.py"), str(tmp_path / "raw.npy")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "bootstrap.png"
    assert out.exists() and png_size(out) == (1280, 960) and out.stat().st_size > 5000      # src/plot.rs:91


def test_metaplot_png_from_results_txt(tmp_path):
    rng = np.random.default_rng(0)
    head = ("run;window;cg_count;region;alpha;beta;1/2*(alpha+beta);pred_steady_state;obs_steady_state;sd_alpha;sd_beta;"
            "ci_alpha_0.025;ci_alpha_0.975;ci_beta_0.025;ci_beta_0.975")        # src/cli/metaprofile.rs:74-78
    lines = [head]
    for w in range(60):
        a, b = rng.uniform(1e-4, 5e-4), rng.uniform(2e-3, 6e-3)
        lines.append(";".join(map(str, ["t", w, 0, "gene", a, b, 0.5 * (a + b), 0.1, 0.1, a / 10, b / 10, 0.8 * a,
                                        1.2 * a, 0.8 * b, 1.2 * b])))
    (tmp_path / "results.txt").write_text("\n".join(lines) + "\n")
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "plot_metaplot.py"), str(tmp_path / "results.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "metaplot.png"
    assert out.exists() and png_size(out) == (1280, 960) and out.stat().st_size > 5000      # src/plot.rs:9
