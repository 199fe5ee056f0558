"""runs one implicit-GEMM forward shape 30 times (for tools/pmc_igemm.sh): H Cin Cout [tile_cfg]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
L = hipabi.load()
dev = torch.device('cuda:0')
H, Cin, Cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 128, 128)
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B, k = 64, 3
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
wp = torch.empty_like(w)
L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
y = torch.empty(B, H, H, Cout, device=dev)
part = torch.empty(L.straps_conv_stat_blocks(B, H, H, Cout, k * k * Cin, cfg), Cout, 2, device=dev)
for _ in range(30):
    assert L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k, 1, 1, cfg, None) == 0
torch.cuda.synchronize()
