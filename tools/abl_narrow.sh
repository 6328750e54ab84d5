export DL_BNSTATS_MIN_BM=256
bash tools/ab_bench.sh DL_NO_BNSTATS
