#!/usr/bin/env python3
"""Would the finish pass hide under the scoring kernel of other tiles?  Two halves of one set scored on ONE stream
(score A, finish A, score B, finish B) against the same two pipelines on TWO streams (finish A can run beside score B).
usage: overlap_probe.py [n_orfs per half]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from ribotricer_amd.engine import PhaseScoreEngine, make_filter
    from ribotricer_amd.synth import synth_csr_device

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_500_000
    th = make_filter()
    sets = [synth_csr_device(n, cfg="cfg3", seed=20260213 + k, device="cuda:0") for k in range(2)]
    engines = [PhaseScoreEngine("cuda:0") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for (c, o), e, s in zip(sets, engines, streams):  # plans, workspaces, per-stream buffers
        with torch.cuda.stream(s):
            e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True)
        e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True)
    torch.cuda.synchronize()

    def run(two_streams, reps=20):
        t = []
        for _ in range(reps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k, ((c, o), e) in enumerate(zip(sets, engines)):
                if two_streams:
                    with torch.cuda.stream(streams[k]):
                        e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True)
                else:
                    e.score(c, o, thresholds=th, algo="tile", reuse_outputs=True)
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) * 1e3)
        t = sorted(t[3:])
        return t[len(t) // 2]

    one = [run(False) for _ in range(3)]
    two = [run(True) for _ in range(3)]
    one2 = [run(False) for _ in range(2)]
    print(json.dumps({"n_orfs_per_half": n, "one_stream_ms": one + one2, "two_streams_ms": two}))


if __name__ == "__main__":
    main()
