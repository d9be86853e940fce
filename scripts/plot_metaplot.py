#!/usr/bin/env python3
"""metaplot.png (src/plot.rs:6-82) from the results.txt the `metaprofile_alphabeta` CLI writes.
usage: scripts/plot_metaplot.py <results.txt> [output-dir = the file's directory]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from alphabeta_rs_amd import plots  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = Path(sys.argv[1])
    print(plots.metaplot(src, Path(sys.argv[2]) if len(sys.argv) > 2 else src.parent))
