"""The launches conv_s2d_kernel (ResnetGenerator down1 forward with fused statistics, up2 data gradient) and conv_s2u_kernel (up2 forward, down1 data gradient) serve, in isolation.
Env switches of the library are read at first use: run once with DL_CONV_S2D=0 (gather GEMM) and once without."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_budget import conv_case, Precision
P = Precision.get('bf16')
rows = []
for name, kind, cin, cout, k, s, p, N, H, W, nf, nd, op in [
        ('G down1 3x3s2 64->128 @512->256 fwd', 'conv', 64, 128, 3, 2, 1, 8, 512, 512, 1, 0, 0),
        ('G up2 convT3x3s2 128->64 @256->512 dgrad', 'convT', 128, 64, 3, 2, 1, 8, 256, 256, 0, 1, 1),
        ('G up2 convT3x3s2 128->64 @256->512 fwd (s2u)', 'convT', 128, 64, 3, 2, 1, 8, 256, 256, 1, 0, 1),
        ('G down1 3x3s2 64->128 @512->256 dgrad (s2u)', 'conv', 64, 128, 3, 2, 1, 8, 512, 512, 0, 1, 0)]:
    r = conv_case(name, kind, cin, cout, k, s, p, N, H, W, P, nf, nd, 0, op=op)
    rows.append(r)
    print(name, {k_: (round(v, 1) if isinstance(v, float) else v) for k_, v in r.items() if k_.startswith(('fwd_', 'dgrad_'))}, flush=True)
print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('DL_')}, 'rows': rows}))
