for b in ploc sah retop; do
python scripts/kernel_time.py --work "S1M:bounce,S1M:primary,S100k:bounce,S100k:primary" --builder $b --tag $b 2>&1 | grep -E "median|rror"
done
