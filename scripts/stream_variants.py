#!/usr/bin/env python3
"""Development aid (GPU box): build the library with different stream-loop depths
(-DABN_STREAM_BLOCKS = row blocks in flight per lane, -DABN_STREAM_WAVES = wavefronts per SIMD the stream
kernel is compiled for) and time phase B of the C5 shard with each.  Every variant must produce the same
bootstrap table (hash printed)."""
import hashlib, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"


def worker(lib, boots):
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import alphabeta_rs_amd as A
    from alphabeta_rs_amd import synthetic
    A.LIB_PATH = Path(lib)
    A._lib = None
    ctx = A.Context(0)
    ped, p0 = synthetic.c5_pedigree()
    plan = A.Plan(ctx, ped[:, :3], 1, 4, boots, options=A.default_options())
    plan.set_windows(ped[:, 3][None, :], np.array([p0]))
    ms = []
    for _ in range(3):
        plan.run()
        ms.append(plan.kernel_ms()["fit_boot"])
    d = plan.download()
    ev = int(d["info_b"]["evals"].sum())
    t = min(ms)
    print(json.dumps({"fit_boot_ms": round(t, 3), "evals": ev, "streamed_TBps": round(ev * (8 * 20100 + 40) / t / 1e9, 3),
                      "raw_sha": hashlib.sha256(d["raw"].tobytes()).hexdigest()[:16]}))
    plan.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], int(sys.argv[3]))
    OUT.mkdir(exist_ok=True)
    variants = [tuple(map(int, v.split(","))) for v in sys.argv[1:]] or [(2, 3, 2048), (4, 2, 2048), (6, 2, 2048)]
    for nb, w, boots in variants:
        lib = OUT / f"libabn_sv_{nb}_{w}.so"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                        "-fPIC", "-shared", f"-DABN_STREAM_BLOCKS={nb}", f"-DABN_STREAM_WAVES={w}", "-o", str(lib),
                        str(ROOT / "alphabeta_rs_amd/csrc/abn_api.hip")], check=True)
        r = subprocess.run([sys.executable, __file__, "--worker", str(lib), str(boots)], capture_output=True, text=True)
        print(f"blocks={nb} waves={w} boots={boots}: {r.stdout.strip()} {r.stderr.strip()[-300:]}", flush=True)


if __name__ == "__main__":
    main()
