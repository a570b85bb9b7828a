"""tools/frame_timeline.py TRACE_DIR OUT.txt -- the timeline of one steady-state frame from a rocprofv3 --kernel-trace CSV: for every launch of the
frame (delimited by consecutive starts of the deformation forward) its start offset, duration and the IDLE GAP in front of it (previous kernel's end
-> this kernel's start), averaged over the frames of the trace by position in the launch sequence.  Development aid: where a frame's wall time
goes that no kernel accounts for."""
import csv, glob, statistics, sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "deform_fwd16_kernel" in e[2] or "deform_fwd_kernel" in e[2]]
frames = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
frames = frames[len(frames) // 3:]                      # steady state
n = statistics.mode(len(fr) for fr in frames)
frames = [fr for fr in frames if len(fr) == n]
lines = ["%d frames of %d launches each (of %d in the trace)" % (len(frames), n, len(starts) - 1),
         "%3s %-58s %9s %9s %9s" % ("#", "kernel", "start_us", "dur_us", "gap_us")]
tot_gap = tot_dur = 0.0
for i in range(n):
    st = statistics.mean(fr[i][0] - fr[0][0] for fr in frames) / 1e3
    du = statistics.mean(fr[i][1] - fr[i][0] for fr in frames) / 1e3
    gp = statistics.mean((fr[i][0] - max(e[1] for e in fr[:i])) if i else 0 for fr in frames) / 1e3
    tot_gap += max(gp, 0.0); tot_dur += du
    lines.append("%3d %-58s %9.1f %9.1f %9.1f" % (i, frames[0][i][2][:58], st, du, gp))
period = statistics.mean(b[0][0] - a[0][0] for a, b in zip(frames[:-1], frames[1:]) if b[0][0] - a[0][0] < 5e7) / 1e3
last_gap = period - statistics.mean(max(e[1] for e in fr) - fr[0][0] for fr in frames) / 1e3
lines.append("frame period %.1f us; kernel time %.1f us; idle gaps inside the frame %.1f us; last kernel's end -> next frame's first start %.1f us"
             % (period, tot_dur, tot_gap, last_gap))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
