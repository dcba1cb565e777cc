#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc counter CSVs of tools/pmc_bench.sh, and the HBM-side traffic of the fused
GEMM kernels per launch against the algorithmic bytes of the launches seen.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read's 128-byte requests as
64 bytes (MI355X_MICROARCH.md, HBM) - doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores.
usage: pmc_traffic.py <dir with pass*/> <workload>   -> JSON on stdout"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, wl = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
grids = collections.defaultdict(collections.Counter)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void lqer::", "").replace("lqer::", "")
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
        if row["Counter_Name"] == "FETCH_SIZE":
            grids[k][int(row["Grid_Size"]) if "Grid_Size" in row else 0] += 1
out = {"workload": wl, "kernels": {}}
for k in sorted(acc):
    if not k.startswith("k_") and "k_lqer" not in k and "i8::" not in k and "d1::" not in k:
        continue
    out["kernels"][k] = {c: round(v[0] / v[1], 1) for c, v in sorted(acc[k].items())}
    out["kernels"][k]["launches_per_pass"] = max(v[1] for v in acc[k].values())
from bench import WORKLOADS
desc, M, r, bias, qc, shapes, layers = WORKLOADS[wl]
gemm = [k for k in out["kernels"] if "k_lqer_gemm" in k or "k_decode1" in k]  # (k_decode1: the whole decode forward)
if gemm:
    fetch = sum(out["kernels"][k].get("FETCH_SIZE", 0) * out["kernels"][k]["launches_per_pass"] for k in gemm)
    write = sum(out["kernels"][k].get("WRITE_SIZE", 0) * out["kernels"][k]["launches_per_pass"] for k in gemm)
    n = sum(out["kernels"][k]["launches_per_pass"] for k in gemm)
    i8 = any("gemm_i8" in k for k in gemm)
    limbs = 1 if qc["x_quantizer"]["block_size"][-1] == 16 else 2  # fp16 A / B of the INT configurations: two bf16 limbs
    rp = -(-r // 16) * 16
    tot, cnt = 0.0, 0
    for K, N, c in shapes:
        act = M * (-(-K // 128) * 128) * 1 if i8 else M * (-(-K // 64) * 64) * 2
        w8 = qc["w_quantizer"].get("width", 4) > 4  # 8-bit weights: the int8 image of codes (1 B), or three 4-bit limb images
        w = N * K * ((1.0 if w8 else 0.5 + 1 / 128) if i8 else (3 * 0.5625 if w8 else 0.5625))
        tot += c * (act + w + M * rp * 2 + N * rp * 2 * limbs + M * N * 2 + (N * 4 if bias else 0))
        if any("k_decode1" in k for k in gemm):
            tot += c * rp * K * 2  # the one-launch forward reads A^T as well
        cnt += c
    alg = tot / cnt
    traffic = (2 * fetch + write) * 1024 / n
    out["gemm"] = {"kernels": gemm, "launches": n, "FETCH_SIZE_KiB_per_launch": round(fetch / n, 1), "WRITE_SIZE_KiB_per_launch": round(write / n, 1),
                   "traffic_bytes_per_launch": int(traffic), "algorithmic_bytes_per_launch": int(alg),
                   "ratio_to_algorithmic": round(traffic / alg, 3),
                   "correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE exact; averages over every launch of the "
                                 "workload's GEMM kernels (the shapes of one decoder layer, weighted by their count)"}
    out["traffic_bytes_per_launch"] = out["gemm"]["traffic_bytes_per_launch"]
    out["ratio_to_algorithmic"] = out["gemm"]["ratio_to_algorithmic"]
print(json.dumps(out, indent=1))
