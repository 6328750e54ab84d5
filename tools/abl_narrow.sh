# scratch driver for the probe of the moment (rewritten per experiment)
timeout 300 python tools/layer_budget.py r02j 2>&1 | grep -E "^norm"
