#!/usr/bin/env python3
"""Would a STRIPED record workspace help on this box?  A 16 GB counts array, K candidate workspaces walked through memory
behind 8 GiB spacers; for every candidate the write penalty (csrc/stream_probe.hip k_stream_rw: 1 152 B written per 32 KiB
read) against five regions of the counts.  If every region has SOME candidate near 0.05 while no single candidate is good
for all regions, giving each region of tiles its own workspace would gain; if the columns are flat, nothing would.
usage: python scripts/stripe_probe.py [candidates]"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ribotricer_amd._probe import write_penalty  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
counts = torch.zeros(3_966_674_436, dtype=torch.int32, device=dev)
plan_like = torch.empty(1 << 30, dtype=torch.uint8, device=dev)  # (the plan sits behind the counts in the bench)
cands, spacers = [], []
for i in range(k):
    cands.append(torch.empty(1 << 30, dtype=torch.uint8, device=dev))
    spacers.append(torch.empty((8 if i < 3 else 16) << 30, dtype=torch.uint8, device=dev))
where = (0.0, 0.25, 0.5, 0.75, 1.0)
rows = []
for i, c in enumerate(cands):
    pen = [write_penalty(counts, c, where=w) for w in where]
    rows.append([round(p[0], 3) for p in pen])
    print("candidate", i, rows[-1], flush=True)
best_single = min(rows, key=lambda r: sum(r))
best_striped = [min(r[j] for r in rows) for j in range(len(where))]
vb = open("/sys/class/drm/card0/device/vbios_version").read().strip() if os.path.exists("/sys/class/drm/card0/device/vbios_version") else None
print(json.dumps({"vbios": vb, "best_single_candidate": best_single, "best_per_region": best_striped,
                  "mean_single": sum(best_single) / len(where), "mean_striped": sum(best_striped) / len(where)}))
