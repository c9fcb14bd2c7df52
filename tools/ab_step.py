"""Round 5: A/B of host-side options on the benched step (tf32h, ViT-B/14, 32 pairs, 518^2) in ONE process, interleaved rounds (the pool's boxes differ
by +-3 %: only same-process pairs mean anything).  Usage: python3 tools/ab_step.py [name=v0,v1,... ...]; default: the round-5 switches.
Each setting: 2 warm-up steps + 6 timed steps per round, 3 rounds, best and all rounds printed."""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gd_amd  # noqa: E402,F401
from gd_amd.options import option, set_option  # noqa: E402


def main():
    specs = sys.argv[1:] or ["wgrad_stream=0,1", "wgrad_reserve_cus=4,8,16,32", "tap_norm_fused=0,1"]
    dev = torch.device("cuda", 0)
    job = bench.Job("vit_base", "mast3r", os.environ.get("AB_DTYPE", "tf32h"), "shared", 32, 518, 300, dev, 0, 1)

    def timed(steps=6, warm=2):
        for i in range(warm):
            job.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            job.step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    timed()
    for spec in specs:
        name, vals = spec.split("=")
        vals = [int(v) for v in vals.split(",")]
        keep = option(name)
        res = {v: [] for v in vals}
        for _ in range(3):
            for v in vals:
                set_option(name, v)
                res[v].append(timed())
        set_option(name, keep)
        print(f"{name}: " + " | ".join(f"{v}: best {min(r):.2f} ms  {[round(x, 2) for x in r]}" for v, r in res.items()), flush=True)


if __name__ == "__main__":
    main()
