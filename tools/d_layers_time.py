"""The PatchGAN's GEMM-shaped layers (c2 .. c5 forward and data gradient) in isolation; dev-build tile experiments are separate processes (DL_CONV_T256X128)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_budget import conv_case, Precision
P = Precision.get('bf16')
for name, cin, cout, k, s, N, H in [('D c1 4x4s2 6->64 @512', 6, 64, 4, 2, 8, 512), ('D c2 4x4s2 64->128 @256', 64, 128, 4, 2, 8, 256), ('D c3 4x4s2 128->256 @128', 128, 256, 4, 2, 8, 128),
                                    ('D c4 4x4s2 256->512 @64', 256, 512, 4, 2, 8, 64), ('D c5 4x4s1 512->512 @32', 512, 512, 4, 1, 8, 32), ('D c6 4x4s1 512->1 @31', 512, 1, 4, 1, 8, 31)]:
    r = conv_case(name, 'conv', cin, cout, k, s, 1, N, H, H, P, 1, 1, 0)
    print(name, {k_: (round(v, 1) if isinstance(v, float) else v) for k_, v in r.items() if k_.startswith(('fwd_', 'dgrad_'))}, flush=True)
