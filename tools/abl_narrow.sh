for rep in 1 2; do
for o in 0 1; do if [ $o = 1 ]; then export DL_OLD_EPILOGUE=1; else unset DL_OLD_EPILOGUE; fi
echo "old=$o"; timeout 100 python tools/blk_probe.py 2>&1 | tail -1; done; done
unset DL_OLD_EPILOGUE
timeout 600 python tools/layer_budget.py r02d 2>&1 | tail -22
