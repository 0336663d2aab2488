for o in "lds_entries=10" "lds_entries=8" "lds_entries=6" "xcd_cursors=1" "block_rays=64" "refill_threshold=12"; do
python scripts/kernel_time.py --work "S10M:bounce,S10M:primary" --reps 8 --opt $o --tag $o 2>&1 | grep -E "median|rror" | cut -c1-110
done
