# scratch driver for the probe of the moment (rewritten per experiment)
timeout 600 python -m pytest tests/test_gpu_post.py -q -m gpu -x 2>&1 | tail -3
