#!/usr/bin/env python3
"""bootstrap.png (src/plot.rs:84-137) from the raw.npy the `alphabeta` CLI writes.
usage: scripts/plot_bootstrap.py <raw.npy> [output-dir = the file's directory]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from alphabeta_rs_amd import plots  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = Path(sys.argv[1])
    print(plots.bootstrap(src, Path(sys.argv[2]) if len(sys.argv) > 2 else src.parent))
