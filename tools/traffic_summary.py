#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/collect_traffic.sh into per-launch HBM bytes of the implicit-GEMM kernels
(traffic_summary.py DIR [LAST] [wgrad]: with `wgrad` of the weight-gradient kernels, all their launches of the run).

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B per lane, LDS-DMA
included) -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv, json, sys, collections
d = sys.argv[1]
tot = {}
n = {}
WG = "wgrad" in sys.argv[2:]
LAST = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0     # 0: the launches of the last 6 forward passes (timed steps + roofline pass),
                                                         # found from bench's own launch count; everything earlier is warm-up / tuning
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f"{d}/pmc_{c}_counter_collection.csv"))
            if (("wgrad_kernel" in r["Kernel_Name"] or "wgrad_pipe_kernel" in r["Kernel_Name"] or "wgrad_wide_kernel" in r["Kernel_Name"]) if WG else
                ("igemm_kernel" in r["Kernel_Name"] or "igemm_pipe_kernel" in r["Kernel_Name"] or "igemm_stream_kernel" in r["Kernel_Name"] or "igemm_persist_kernel" in r["Kernel_Name"]))
            and r["Counter_Name"] == c]
    rows.sort()
    if not LAST and not WG:
        import glob
        per = json.loads(open(glob.glob(f"{d}/bench_*.json")[0]).read().strip().splitlines()[-1])["roofline"]["launches_per_step"]
        LAST = 6 * per
    rows = rows[-LAST:] if LAST else rows
    tot[c], n[c] = sum(v for _, v in rows), len(rows)
launches = n["FETCH_SIZE"]
fetch = 2.0 * tot["FETCH_SIZE"] * 1024
write = tot["WRITE_SIZE"] * 1024
print(json.dumps({
    "kernel": "wgrad_kernel + wgrad_pipe_kernel + wgrad_wide_kernel" if WG else "igemm_persist_kernel + igemm_pipe_kernel + igemm_kernel + igemm_stream_kernel", "launches_counted": launches,
    "hbm_read_bytes_per_launch": fetch / launches, "hbm_write_bytes_per_launch": write / launches,
    "hbm_bytes_per_launch": (fetch + write) / launches,
    "correction": "FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads), KiB -> bytes; WRITE_SIZE exact",
    "command": "rocprofv3 --pmc <C> -- python3 " + ("tools/train_steps.py 3" if WG else "bench.py --steps 3 --warmup 1 --no-train --no-nms --no-cpu --no-resnet") + " (separate passes)"}))
