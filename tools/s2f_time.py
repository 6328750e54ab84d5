"""The stride-2 layers of the 5G + 5D step in isolation: transposed-conv forwards (with fused statistics) and stride-2 data gradients.
Env switches of the library are read at first use: run once with DL_CONV_S2F=0 (4-phase gather GEMM) and once without (fused four-phase tile)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_budget import conv_case, Precision
P = Precision.get('bf16')
rows = []
for name, kind, cin, cout, k, s, p, N, H, W, nf, nd, op in [
        ('G up2 convT3x3s2 128->64 @256->512', 'convT', 128, 64, 3, 2, 1, 8, 256, 256, 1, 0, 1),
        ('G up1 convT3x3s2 256->128 @128->256', 'convT', 256, 128, 3, 2, 1, 8, 128, 128, 1, 0, 1),
        ('G down1 3x3s2 64->128 @512->256 dgrad', 'conv', 64, 128, 3, 2, 1, 8, 512, 512, 0, 1, 0),
        ('G down2 3x3s2 128->256 @256->128 dgrad', 'conv', 128, 256, 3, 2, 1, 8, 256, 256, 0, 1, 0),
        ('D c2 4x4s2 64->128 @256->128 dgrad', 'conv', 64, 128, 4, 2, 1, 8, 256, 256, 0, 1, 0),
        ('D c3 4x4s2 128->256 @128->64 dgrad', 'conv', 128, 256, 4, 2, 1, 8, 128, 128, 0, 1, 0),
        ('D c4 4x4s2 256->512 @64->32 dgrad', 'conv', 256, 512, 4, 2, 1, 8, 64, 64, 0, 1, 0)]:
    r = conv_case(name, kind, cin, cout, k, s, p, N, H, W, P, nf, nd, 0, op=op)
    rows.append(r)
    print(name, {k_: (round(v, 1) if isinstance(v, float) else v) for k_, v in r.items() if k_.startswith(('fwd_', 'dgrad_'))}, flush=True)
print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('DL_')}, 'rows': rows}))
