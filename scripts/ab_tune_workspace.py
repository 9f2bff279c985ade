#!/usr/bin/env python3
"""bench.py with and without engine.tune_workspace, alternating, one process each.  usage: ab_tune_workspace.py [pairs]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for k in range(2 * (int(sys.argv[1]) if len(sys.argv) > 1 else 3)):
    extra = ["--no-tune-workspace"] if k % 2 == 0 else []
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "30", "--warmup", "5", "--cpu-sample", "0", "--no-fused"] + extra,
                         capture_output=True, text=True)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print("FAILED", out.stderr[-300:])
        continue
    d = json.loads(line[-1])
    r = d["roofline"]
    wp = d["config"]["workspace_placement"]
    print("tuned " if not extra else "plain ", "value %.3e" % d["value"], "kernel", round(r["kernel_ms"], 3), round(r["frac"], 3), "step_frac", round(r["step_frac"], 3),
          "penalty", [round(x, 3) for x in r["stream_read"]["record_write_penalty"]["penalty"]], wp["step_ms"] if isinstance(wp, dict) else "", flush=True)
