"""Round 5: A/B of host-side options on the benched step (tf32h, ViT-B/14, 32 pairs, 518^2) in ONE process, interleaved rounds (the pool's boxes differ
by +-3 %: only same-process pairs mean anything).  Usage: python3 tools/ab_step.py [name=v0,v1,... ...] (options.py names, or lib.<knob> for csrc/gd_knobs.h); default: the round-5 switches.
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
    specs = sys.argv[1:] or ["tap_norm_fused=0,1", "adapter_ln=0,1"]
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
    L = gd_amd._lib.lib()
    for spec in specs:
        name, vals = spec.split("=")
        vals = [int(v) for v in vals.split(",")]
        lib_knob = name.startswith("lib.")      # lib.<knob>: an option of the library (csrc/gd_knobs.h) through gd_debug_set
        get = (lambda: L.gd_debug_get(name[4:].encode())) if lib_knob else (lambda: option(name))
        put = (lambda v: L.gd_debug_set(name[4:].encode(), v)) if lib_knob else (lambda v: set_option(name, v))
        keep = get()
        res = {v: [] for v in vals}
        for _ in range(3):
            for v in vals:
                put(v)
                res[v].append(timed())
        put(keep)
        print(f"{name}: " + " | ".join(f"{v}: best {min(r):.2f} ms  {[round(x, 2) for x in r]}" for v, r in res.items()), flush=True)


if __name__ == "__main__":
    main()
