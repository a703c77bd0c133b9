"""timeline of ONE optimizer step from a rocprofv3 --kernel-trace csv: every dispatch's start / end relative to the step's first kernel, and the
stream (queue) it ran on — what shows whether launches on different streams overlap.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o p -- python3 bench.py --config c3 --precision bf16 --steps 10 --warmup 2 --preroll 20 --preroll-seconds 0 --no-cpu-baseline --no-psnr --no-extras
    python tools/step_timeline.py gpurun_out/tl [first-kernel-substring] [step-index-from-the-end] [min-grid-of-the-first-kernel]
(the last argument: when a step launches the first kernel twice — body and tail — a step starts at the launch whose grid is at least that many threads)"""
import csv
import glob
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k16<16, true, 1, 4"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
path = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= min_grid]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
print("step of %d dispatches (%s)" % (i1 - i0, path))
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f .. %9.1f us  (%7.1f)  queue %-4s grid %-8s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:70]))
print("next step starts at %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3) if i1 < len(rows) else "")
