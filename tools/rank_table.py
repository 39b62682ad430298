#!/usr/bin/env python3
"""Markdown table of a tools/rank_alone.py result (profiles/r06_rank_alone.json, ..._cfg5.json): one row per (world, scene, schedule).
usage: rank_table.py <json>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
single = d["single_gpu_fps"]
rows = {}
for r in d["runs"]:
    rows.setdefault((r["world"], r["scene"], r["speculate"]), {})[r["frames_in_flight"]] = r
print("| world | scene | schedule | slowest / fastest rank alone, ms (1 / 2 frames in flight) | predicted fps, 1 / 2 in flight | single GPU, 1 / 2 in flight | "
      "predicted ÷ single GPU | launches per frame, slowest rank | repair frames | wire MB per frame, busiest rank |")
print("|---|---|---|---|---|---|---|---|---|---|")
for (world, scene, spec), by in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1] != "orbit", -kv[0][2])):
    a, b = by.get(1), by.get(2)
    s1 = single.get(f"{scene} speculate={spec} frames_in_flight=1")
    s2 = single.get(f"{scene} speculate={spec} frames_in_flight=2")
    sl = lambda r: max(r["ranks"], key=lambda x: x["ms_per_frame_alone"])  # noqa: E731
    cell = lambda f, *rs: " / ".join(f(r) if r else "–" for r in rs)  # noqa: E731
    print(f"| {world} | {scene} | {'speculated' if spec else 'unspeculated'} | "
          f"{cell(lambda r: '%.3f' % r['slowest_rank_ms'], a, b)} ; {cell(lambda r: '%.3f' % r['fastest_rank_ms'], a, b)} | "
          f"{cell(lambda r: '%.0f' % r['predicted_fps'], a, b)} | {s1:.0f} / {s2:.0f} | "
          f"{cell(lambda r: '%.2f' % (r['predicted_fps'] / (s1 if r['frames_in_flight'] == 1 else s2)), a, b)} | "
          f"{cell(lambda r: str(sl(r)['launches_per_frame']), a, b)} | {cell(lambda r: '%.2f' % sl(r)['repair_frames'], a, b)} | "
          f"{cell(lambda r: '%.1f' % (max(x['wire_bytes_per_frame'] for x in r['ranks']) / 1e6), a, b)} |")
