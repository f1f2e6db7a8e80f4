#!/usr/bin/env python3
"""HBM-side bytes of the projection GEMM launches of the C4 step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
separate passes, as MI355X_MICROARCH.md prescribes) over `tools/kernel_bench.py gemm_step --B 64 --iters 0`.
Corrections (same guide): FETCH_SIZE x2 on gfx950 for 16-byte-per-lane streaming reads, WRITE_SIZE exact; counters in KiB;
FETCH_SIZE also counts Infinity-Cache hits (fabric traffic, not necessarily HBM).
Usage: python tools/gemm_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_dispatch(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("gemm_kernel" in r["Kernel_Name"] or "gemm_pers_kernel" in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


SHAPES = [(4096, 1024, "q|k|v fwd"), (1024, 2048, "o_proj fwd"), (6144, 1024, "gate|up fwd"), (1024, 3072, "down fwd"),
          (1024, 4096, "dX q|k|v"), (2048, 1024, "dX o_proj"), (1024, 6144, "dX gate|up")]
M = 131072
fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
n = len(SHAPES)
assert len(fetch) >= 2 * n and len(write) >= 2 * n, (len(fetch), len(write))
fetch, write = fetch[n:2 * n], write[n:2 * n]          # round 1 of gemm_step (round 0 is the warm-up; --iters 0 adds none)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/kernel_bench.py gemm_step --B 64 --iters 0",
       "corrections": "FETCH_SIZE x2 on gfx950; WRITE_SIZE exact; KiB; FETCH_SIZE counts Infinity-Cache hits too", "launches": {}}
for (N, K, name), f, w in zip(SHAPES, fetch, write):
    alg = 2 * (M * K + N * K + M * N)
    hbm = int(2 * f * 1024 + w * 1024)
    out["launches"][name] = {"M": M, "N": N, "K": K, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fabric_read_bytes": int(2 * f * 1024),
                             "write_bytes": int(w * 1024), "hbm_bytes": hbm, "algorithmic_bytes": alg, "ratio": round(hbm / alg, 2),
                             "read_ratio": round(2 * f * 1024 / (2 * (M * K + N * K)), 2)}
    print(f"{name:12s} N={N:5d} K={K:5d}: read {2 * f * 1024 / 1e9:6.2f} GB (x{out['launches'][name]['read_ratio']:.1f} of A+W)  write {w * 1024 / 1e9:5.2f} GB  total x{hbm / alg:.2f} of algorithmic")
json.dump(out, open(sys.argv[3], "w"), indent=1)
