import sys, json
sys.path.insert(0, '/root/repo')
import torch
import alphabeta_rs_amd as A
import bench
A.load_library()
ctx = A.Context(0, stream=torch.cuda.current_stream().cuda_stream)
r = bench.pairwise_bench(A, ctx, shapes=((50, 2_000_000), (50, 8_000_000), (50, 32_000_000), (15, 32_000_000), (50, 100_000_000)), reps=5)
for s in r["shapes"]:
    print(s["samples"], s["sites"], "%.1f us" % (s["kernel_ms_avg"] * 1e3), "%.0f GB/s" % s["achieved_GBps"], "frac %.3f" % s["frac"], s["sane"])
