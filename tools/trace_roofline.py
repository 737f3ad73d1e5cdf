#!/usr/bin/env python3
"""From `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py`: duration of the yolo_igemm launches of bench.py's
roofline pass (3 forward passes, bracketed by two yolo::sumsq_kernel marker launches), for comparison with bench.py's own
HIP-event numbers (roofline.avg_launch_ms / kernel_ms_per_step).  usage: trace_roofline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("yolo::sumsq_kernel(")]
assert len(marks) >= 2, "marker launches not found"
seg = rows[marks[0] + 1: marks[1]]
def _ig(name):
    return "igemm_kernel" in name or "igemm_pipe_kernel" in name or "igemm_stream_kernel" in name or "igemm_persist_kernel" in name
ig = [r for r in seg if _ig(r["Kernel_Name"])]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in ig]
other = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in seg if not _ig(r["Kernel_Name"]))
print(f"roofline pass: {len(ig)} igemm launches in 3 forward passes ({len(ig) // 3} per pass); mean {sum(dur) / len(dur):.4f} ms per launch; "
      f"igemm {sum(dur) / 3:.3f} ms + other kernels {other / 3:.3f} ms per forward pass")
