#!/usr/bin/env python3
"""Kernel timeline of a PRODUCT frame from a rocprofv3 --kernel-trace CSV: the median over every complete frame of the trace.

Take the trace from a run WITHOUT event stamps (`bench.py --no-stamps`, tools/trace.sh): a hipEvent record between two kernels is
a barrier packet on the queue, the kernel behind it is not dispatched under the kernel in front of it, and the frame grows by a few
microseconds per record -- round 5's timeline showed a stamped frame (eight of them trail a short bench run) and read the stamps as
"23 % of the frame idle between kernels".  Frames with stamps can still be told from the others in a trace that has both: pass
--min-gaps to see them apart (a frame counts as stamped when at least that many of its kernel boundaries are wider than 3 us).

A frame starts at its first kernel: object_cull_kernel (long scenes), else frame_cull_fused_kernel (short scenes inside
chordvis_render_frame), else group_cull_count_kernel.  Frames are grouped by their launch sequence (the names in order); the most frequent sequence is printed: per launch
the median start offset, duration and gap to the launch before, then the frame's span, the sum of its kernels' durations and the
difference (time the device ran none of the frame's kernels).  Other sequences (a frame that makes its tile schedule again: one
launch more) are listed with their counts and medians.

  python tools/timeline.py r_kernel_trace.csv [--min-gaps N]
"""
import collections
import csv
import statistics
import sys


def short(name):
    return name.replace('void ', '').replace('chord::', '').split('(')[0][:64]


def main(path, min_gaps=None):
    rows = [r for r in csv.DictReader(open(path)) if r.get('Start_Timestamp')]
    if not rows:
        print("no kernel rows in", path)
        return 1
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [short(r['Kernel_Name']) for r in rows]
    first = next(k for k in ('object_cull_kernel', 'frame_cull_fused_kernel', 'group_cull_count') if any(k in n for n in names) or k == 'group_cull_count')
    starts = [i for i, n in enumerate(names) if first in n]
    if len(starts) < 3:
        print("fewer than two complete frames in", path)
        return 1
    frames = []                                           # (sequence, [(start, end)], stamped?)
    for a, b in zip(starts[:-1], starts[1:]):             # (the last frame may be cut short by the end of the trace: dropped)
        iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows[a:b]]
        wide = sum(1 for (s0, e0), (s1, e1) in zip(iv[:-1], iv[1:]) if s1 - e0 > 3000)
        frames.append((tuple(names[a:b]), iv, wide))
    if min_gaps is not None:
        stamped = [f for f in frames if f[2] >= min_gaps]
        frames = [f for f in frames if f[2] < min_gaps]
        print("%d frames with %d or more boundaries wider than 3 us (stamped) set aside; %d frames left" % (len(stamped), min_gaps, len(frames)))
    groups = collections.defaultdict(list)
    for seq, iv, wide in frames:
        groups[seq].append(iv)
    order = sorted(groups, key=lambda s: -len(groups[s]))

    def summary(ivs):
        spans = [iv[-1][1] - iv[0][0] for iv in ivs]
        sums = [sum(e - s for s, e in iv) for iv in ivs]
        nxt = []
        return statistics.median(spans) / 1e3, statistics.median(sums) / 1e3, statistics.median([a - b for a, b in zip(spans, sums)]) / 1e3, nxt

    seq = order[0]
    ivs = groups[seq]
    print("%d complete frames in the trace, %d launch sequences; the most frequent one (%d frames, %d launches), medians over those frames:"
          % (len(frames), len(order), len(ivs), len(seq)))
    for k, name in enumerate(seq):
        off = statistics.median([iv[k][0] - iv[0][0] for iv in ivs]) / 1e3
        dur = statistics.median([iv[k][1] - iv[k][0] for iv in ivs]) / 1e3
        gap = statistics.median([iv[k][0] - iv[k - 1][1] for iv in ivs]) / 1e3 if k else 0.0
        # (the bench alternates two views: a launch whose work differs between them has TWO typical durations and a median that sits on
        # either -- the second pass's tile kernel read 22.8 us in one trace and 26.2 in the next; mean and quartiles say so)
        ds = sorted((iv[k][1] - iv[k][0]) / 1e3 for iv in ivs)
        print("%8.1f us  dur %7.1f  gap %5.1f  %-48s mean %6.1f  quartiles %6.1f .. %6.1f" % (off, dur, gap, name, statistics.fmean(ds), ds[len(ds) // 4], ds[(3 * len(ds)) // 4]))
    span, tot, idle, _ = summary(ivs)
    print("launches %d: first start to last end %.1f us, sum of kernel durations %.1f us, span - sum %.1f us (medians over %d frames; mean span %.1f us)"
          % (len(seq), span, tot, idle, len(ivs), statistics.fmean([iv[-1][1] - iv[0][0] for iv in ivs]) / 1e3))
    # frame to frame: start of a frame to the start of the next (what a timed loop divides by), where consecutive
    per = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
    per.sort()
    core = per[len(per) // 10: len(per) - len(per) // 10] or per
    print("frame start to next frame start: median %.1f us, middle 80 %% %.1f .. %.1f us (%d frames; warm-up and the bench's sync points included)"
          % (statistics.median(per), core[0], core[-1], len(per)))
    for s in order[1:6]:
        sp, tt, idl, _ = summary(groups[s])
        extra = [n for n in s if s.count(n) > seq.count(n)]
        print("other sequence: %3d frames, %d launches, span %.1f us, sum %.1f us, span - sum %.1f us%s"
              % (len(groups[s]), len(s), sp, tt, idl, ("; more of: " + ", ".join(sorted(set(extra)))) if extra else ""))
    return 0


if __name__ == "__main__":
    mg = int(sys.argv[sys.argv.index("--min-gaps") + 1]) if "--min-gaps" in sys.argv else None
    sys.exit(main(sys.argv[1], mg))
