# scratch driver for the probe of the moment (rewritten per experiment)
timeout 300 python -m pytest tests/test_gpu_post.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/post_probe.py 2>&1 | tail -3
