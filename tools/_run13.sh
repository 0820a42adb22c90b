for cam in K1 K2 diag -y; do
python tools/exp_variants.py 1024 0 $cam 2>&1 | grep -v "^/opt" | sed "s/^/base /"
SVR_LPT=1 python tools/exp_variants.py 1024 0 $cam 2>&1 | grep -v "^/opt" | sed "s/^/lpt  /"
done
