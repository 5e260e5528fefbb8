"""Known-byte-count kernel for calibrating rocprofv3's FETCH_SIZE on this access pattern:
rls_checksum reads n floats once, one dword per lane, coalesced (the same pattern as the closure
kernels' input planes).  Run under `rocprofv3 --pmc FETCH_SIZE`; expected bytes = 4 n per launch."""
import sys
sys.path.insert(0, ".")
import torch
import rlshaders_amd as R

n = 1 << 28
ctx = R.Context(0)
t = torch.rand(n, device="cuda")
for _ in range(4):
    R.checksum(ctx, t)
print("calibration launches done; bytes per launch =", 4 * n)
