"""FSQ forward at 2^24 vectors (36 bytes each): best of three runs of 20 launches, HIP events.  r4 used it to compare launch shapes
(vectors per thread x grid size, non-temporal against plain stores); the kernel keeps the winner: two vectors per thread, one block
per 512 vectors."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
ge.build()
from mqgan_amd.quantizer import FSQ
q = FSQ([8, 5, 5, 5]).cuda().eval()
n = 1 << 24
torch.manual_seed(7)
z = (torch.randn(n, 4, device="cuda") * 2.0).view(1, n, 4)
for _ in range(3): q(z)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    s.record()
    for _ in range(20): q(z)
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 20)
print(f"fsq forward {best*1e3:.1f} us  {36*n/best/1e9:.2f} TB/s")
