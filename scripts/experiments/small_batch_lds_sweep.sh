for le in 4 6 8 10 12; do for bpc in 6 8; do
echo "lds_entries=$le blocks_per_cu=$bpc"; python scripts/kernel_time.py --work S100k:primary,S1M:primary,S1M:bounce --side 1024 --reps 40 --opt lds_entries=$le --opt blocks_per_cu=$bpc --opt ray_image_width=1024 2>&1 | grep -E "ms|median" | cut -c1-200
done; done
