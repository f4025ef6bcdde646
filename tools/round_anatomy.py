#!/usr/bin/env python3
"""Anatomy of one round of the multi-version erosion scheduler from a rocprofv3 kernel trace: the launches of a round by position (the trace waves start a round), each with
its average / median duration and the average idle gap to the launch before it.  usage: round_anatomy.py <rocprofv3 output dir>"""
import csv, glob, os, statistics, sys


def main(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                if "speculative_erosion" in r["Kernel_Name"] or "k_spec_round" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rounds, cur = [], None
    for s, e, n in rows:
        if "k_waves<" in n and "nolds" not in n:  # the trace waves: a round starts
            cur = []
            rounds.append(cur)
        if cur is not None:
            cur.append((s, e, n))
    modal = statistics.mode(map(len, rounds)) if rounds else 0
    full = [r for r in rounds if len(r) == modal]
    print(f"rounds {len(rounds)}, with the full launch count ({len(full[0]) if full else 0}): {len(full)}")
    if not full:
        return
    span = [r[-1][1] - r[0][0] for r in full]
    nxt = [b[0][0] - a[-1][1] for a, b in zip(rounds, rounds[1:])]
    print(f"round span avg {statistics.mean(span) / 1e3:.1f} us, gap between rounds avg {statistics.mean(nxt) / 1e3:.1f} us (median {statistics.median(nxt) / 1e3:.1f})")
    print(f"{'pos':>3s} {'avg_us':>8s} {'med_us':>8s} {'gap_before_us':>14s}  kernel")
    for i in range(len(full[0])):
        dur = [r[i][1] - r[i][0] for r in full]
        gap = [r[i][0] - r[i - 1][1] for r in full] if i else [0]
        name = full[0][i][2]
        tag = "waves" if "k_waves<" in name else ("waves_nolds" if "nolds" in name else ("generic" if "k_generic" in name else name[:40]))
        print(f"{i:3d} {statistics.mean(dur) / 1e3:8.2f} {statistics.median(dur) / 1e3:8.2f} {statistics.mean(gap) / 1e3:14.2f}  {tag}")


if __name__ == "__main__":
    main(sys.argv[1])
