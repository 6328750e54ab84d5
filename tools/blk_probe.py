"""Timing of the dominant ResnetBlock conv (3x3, 256->256, 8x128x128) forward (with fused statistics) and data gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_budget import conv_case, Precision
r = conv_case('G block 3x3 256->256 @128', 'conv', 256, 256, 3, 1, 1, 8, 128, 128, Precision.get('bf16'), 1, 1, int(os.environ.get('BLK_WGRAD', '0')))
print('fwd %.1f us  dgrad %.1f us  wgrad %.1f us' % (r['fwd_us'], r['dgrad_us'], r.get('wgrad_us', 0)))
