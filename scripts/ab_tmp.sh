for o in "lds_entries=10" "lds_entries=8" "lds_entries=7" "lds_entries=6" "tri_threshold=2" "tri_threshold=6" "refill_threshold=4" "refill_threshold=12" "block_rays=64" "block_rays=256"; do
python scripts/kernel_time.py --work "S1M:bounce" --opt $o --tag $o 2>&1 | grep -E "median|rror" | cut -c1-90
done
