import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dream_amd import ops
B, H, C = 128, 100, 256
x = torch.randn(B, H, H, C, device="cuda"); dy = torch.randn(B, H, H, C, device="cuda")
for _ in range(2): ops.conv3x3_wgrad(x, dy, C, C, 0)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.conv3x3_wgrad(x, dy, C, C, 0); e.record(); e.synchronize()
    best = min(best, s.elapsed_time(e))
print("wgrad 256->256@100 B=128: %.2f ms, %.1f TF" % (best, 2.0 * B * H * H * C * C * 9 / best / 1e9))
