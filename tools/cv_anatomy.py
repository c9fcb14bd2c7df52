import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gd_amd
from gd_amd import ops
P, hw, C = 32, 1369, 768
f1 = torch.randn(P, hw, C, device="cuda").bfloat16(); f2 = torch.randn(P, hw, C, device="cuda").bfloat16()
t1 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1)); t2 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1))
m1 = torch.rand(P, hw, device="cuda") > 0.3; m2 = torch.rand(P, hw, device="cuda") > 0.3
ts = ops.cost_volume_teacher_stats(t1, t2)
for dbg in (0, 32, 0, 32, 1, 2, 4, 7):      # 32: the DBG instantiation with nothing switched off
    gd_amd._lib.lib().gd_debug_set(b"cv_dbg", dbg)
    t = ops.time_on_stream(lambda: ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt", tstats=ts), 3, 10)
    print(f"dbg={dbg} (1: no teacher, 2: no epilogue, 4: no mfma): fwd {t*1e6:8.1f} us total for {P} pairs", flush=True)
